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
        grad_z, grad_table = ext.pq_loss_backward(z, table, grad_loss.contiguous())
        return grad_z, grad_table, None


def pq_loss(z: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    return PQLoss.apply(z, table)


def pq_loss_and_codes(z: torch.Tensor, table: torch.Tensor):
    """(loss, codes [N * H, S, M]) for z [N, S, H, E]: the loss's argmin is the PQ code."""
    return PQLoss.apply(z, table, True)
