"""kernels.pq_loss -- fused PQ codebook training loss (no reference counterpart).

The reference composes this loss from ~25 torch ops around ``kernels.cdist``
(``naive_gpt/layers/basic/quantizer.py:80-111``); ``spt_pq_loss_*`` (pq_loss.hip) is the
same function of ``(z, weight)`` as two kernels.  Gradients flow to both, as in the
reference (the second MSE term pulls the inputs towards their centroids).
"""
import torch

from naive_gpt import ext


class PQLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, z: torch.Tensor, table: torch.Tensor, want_codes: bool = False):
        z, table = z.contiguous(), table.contiguous()
        ctx.save_for_backward(z, table)
        if want_codes:
            loss, codes = ext.pq_loss_forward(z, table, want_codes=True)
            ctx.mark_non_differentiable(codes)
            ctx.set_materialize_grads(False)     # no zero tensor (4 MB) for the codes' "gradient"
            return loss, codes
        return ext.pq_loss_forward(z, table)

    @staticmethod
    def backward(ctx, grad_loss: torch.Tensor, *unused):
        if grad_loss is None:
            return None, None, None
        z, table = ctx.saved_tensors
        # z usually has a second consumer, the attention core, whose gradient for it may exist
        # already (`pending_gradient`): the loss's gradient is then added INTO that tensor by the
        # kernel and nothing is returned for z -- autograd would otherwise add the two with an
        # elementwise pass (2 x 33 MB read, 33 MB written per q and k and layer).
        other = take_pending_gradient(z)
        if other is not None:
            _, grad_table = ext.pq_loss_backward(z, table, grad_loss.contiguous(), accumulate_into=other)
            return None, grad_table, None
        grad_z, grad_table = ext.pq_loss_backward(z, table, grad_loss.contiguous())
        return grad_z, grad_table, None


# ---- gradients of z that another backward has produced already ---------------------------------
# Filled by the attention core's backward (layers/sparse/attention.py: _MfmaAttention) for its q
# and k, emptied by PQLoss.backward above or, at the latest, by the next forward.  Autograd runs
# the node created last first, and the attention is created after the loss of the same q / k, so
# in a training step the entry is there; if it is not (another order, another graph), nothing
# happens and autograd sums as usual.  The consumer of q's total gradient (the projection's
# backward) depends on BOTH nodes, so it cannot run before the tensor has been completed.
_PENDING = {}


def pending_gradient(z: torch.Tensor, grad: torch.Tensor, tag=None) -> None:
    """`grad` is the gradient some backward has just produced for a consumer of `z` (`tag`: which
    tensor derived from z, when it is not z itself); a later backward may add its own into it."""
    if grad.dtype == torch.float32 and grad.is_contiguous() and (tag is not None or grad.shape == z.shape):
        _PENDING[(z.data_ptr(), tuple(z.shape), tag)] = grad


def take_pending_gradient(z: torch.Tensor, tag=None):
    return _PENDING.pop((z.data_ptr(), tuple(z.shape), tag), None)


def drop_pending_gradients() -> None:
    _PENDING.clear()


def pq_loss(z: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    return PQLoss.apply(z, table)


def pq_loss_and_codes(z: torch.Tensor, table: torch.Tensor):
    """(loss, codes [N * H, S, M]) for z [N, S, H, E]: the loss's argmin is the PQ code."""
    return PQLoss.apply(z, table, True)
