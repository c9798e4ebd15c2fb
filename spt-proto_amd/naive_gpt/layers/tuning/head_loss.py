"""The language-model head and its cross-entropy as one function (frozen head with a LoRA adapter).

Reference: ``script/4-sparse-tuning-0.py:45-59`` -- ``loss_fn(model(src).flatten(end_dim=-2),
target.flatten())`` with ``model.lm_output`` a ``LoRALinear`` after the upgrade (``lora.py:43-80``).
Op by op that is the step's largest non-layer cost at a 30k vocabulary: the logits [tokens, V]
(1 GB at BERT-large, 8192 tokens) exist three times at the step's memory peak (logits,
log-softmax, their gradient), are read and written five times by the loss, and the backward
``dX = dlogits W`` (contraction over V = 30522, not a multiple of 4) falls to the library's fp32
GEMM: 7.4 ms of a 73 ms step.  Here:

* the logits are written ONCE, into rows padded to a multiple of 32 columns
  (``grouped_gemm_fused(..., out=)``: base product + LoRA side product in one launch);
* ``ext.cross_entropy_grad_`` turns them into the loss and, in place, into their own gradient;
* the gradients of the head's input and of its two LoRA tables are formed right away from that
  buffer -- ``dX`` on the split-bf16 matrix-core GEMM, whose contraction now runs over aligned,
  zero-padded rows -- and the buffer is released before the backward of the layers starts.
  ``backward`` only scales the three small results by the incoming scalar.

The frozen weight is read past its last row by up to 31 rows (the padded contraction): its
storage is re-allocated once with that many zero rows behind it (`_pad_rows`); the parameter keeps
its shape, its values and its place in the state_dict.
"""
import weakref

import torch

from .lora import LoRALinear, _down, _mfma_linear_usable, _one_group, tall_tn

_PADDED = {}
_SPLIT = 3            # parts of the vocabulary in the head's backward GEMM
_PAD = 32 * _SPLIT    # rows of the padded weight / columns of the padded logits: whole k-steps per part
_INDEX = {}


def _index(kind: str, T: int, device):
    key = (T, str(device))
    hit = _INDEX.get(key)
    if hit is None:
        with torch.inference_mode(False):
            t = torch.arange(T, dtype=torch.int32, device=device)
            hit = {
                'offsets': torch.arange(0, (_SPLIT + 1) * T, T, dtype=torch.int32, device=device),
                # row p = g * T + t of the parts reads buffer row t * _SPLIT + g
                'gather': torch.cat([t * _SPLIT + g for g in range(_SPLIT)]).contiguous(),
                # token t sums rows t, T + t, 2 T + t of the parts
                'pos': torch.stack([t + g * T for g in range(_SPLIT)], dim=1).contiguous(),
            }
        _INDEX.clear()                  # one token count at a time
        _INDEX[key] = hit
    return hit[kind]


def _split_offsets(T, device):
    return _index('offsets', T, device)


def _split_gather(T, device):
    return _index('gather', T, device)


def _split_pos(T, device):
    return _index('pos', T, device)


def _pad_rows(weight: torch.Tensor, rows: int) -> None:
    """Make `weight` [V, d] a view of a zero-padded [rows, d] buffer (once per parameter)."""
    hit = _PADDED.get(id(weight))
    if hit is not None and hit[0]() is weight and hit[1] == weight.data_ptr():
        return
    with torch.no_grad():
        buf = torch.zeros([rows, weight.size(1)], dtype=weight.dtype, device=weight.device)
        buf[:weight.size(0)].copy_(weight)
        weight.data = buf[:weight.size(0)]
    _PADDED[id(weight)] = (weakref.ref(weight, lambda _, k=id(weight): _PADDED.pop(k, None)),
                           weight.data_ptr())


class _FrozenLoRAHeadLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, left, right, target, ignore_index: int):
        from naive_gpt import ext
        V, d = weight.shape
        x2 = x.reshape(-1, d)
        T = x2.size(0)
        Vp = (V + _PAD - 1) // _PAD * _PAD
        _pad_rows(weight, Vp)
        one = _one_group(T, x2.device)
        u = _down(x, x2, left, d % 32 == 0)                            # [T, r]
        images = d % 32 == 0 and x2.is_contiguous()
        buf = torch.empty([T, Vp], dtype=torch.float32, device=x2.device)
        ext.grouped_gemm_fused(x2, weight, one, 1, V, d, 0, d, 1, T, a2=u.contiguous(),
                               b2=right.contiguous(),
                               a_image=ext.cached_image(x2, x) if images else None,   # (else: A32)
                               w_image=ext.weight_image(weight) if images else None, out=buf)
        target = target.reshape(-1).contiguous()
        # (a target outside [0, V) that is not `ignore_index` counts as ignored -- torch raises a
        # device-side assert there; documented in csrc/head_loss.hip)
        counted = ((target != ignore_index) & (target >= 0) & (target < V)).sum().to(torch.float32)
        scale = (1.0 / counted.clamp_min(1.0)).reshape(1)
        rows_loss = ext.cross_entropy_grad_(buf, V, target, scale, ignore_index)
        # the mean over the counted targets; no counted target at all: 0 / 0 = NaN, as torch's mean
        loss = rows_loss.sum() / counted
        # buf now holds d loss / d logits (pad columns zero).  Everything that needs it, now:
        dl = buf[:, :V]
        du = torch.matmul(dl, right)                                   # [T, r]
        # dX = dl W, a contraction over the (padded) vocabulary: [T / 128] x [d / 128] tiles with a
        # 30k-long k-loop fill a third of the GPU's workgroup slots.  Split in _SPLIT parts over
        # the vocabulary instead -- row t of part g is row t * _SPLIT + g of the SAME buffer seen as
        # [T * _SPLIT, Vp / _SPLIT], part g's weights are rows g Vp / _SPLIT .. of W: the grouped
        # GEMM's own row gather and group stride -- and the parts summed in a fixed order.
        kc = Vp // _SPLIT
        parts = ext.grouped_gemm_fused(
            buf.view(T * _SPLIT, kc), weight, _split_offsets(T, x2.device), _SPLIT, n=d, k=kc,
            w_group_stride=kc * d, w_ldn=1, w_ldk=d, n_rows=T * _SPLIT,
            gather=_split_gather(T, x2.device))
        grad_x = ext.rows_combine(parts, _split_pos(T, x2.device))
        del parts
        grad_x.addmm_(du, left.t())
        grad_right = torch.matmul(dl.t(), u)                           # [V, r]
        grad_left = tall_tn(x2, du)
        del buf, dl
        ctx.save_for_backward(grad_x.view(x.shape), grad_left, grad_right)
        return loss

    @staticmethod
    def backward(ctx, g):
        grad_x, grad_left, grad_right = ctx.saved_tensors
        return grad_x * g, None, grad_left * g, grad_right * g, None, None


def fused_usable(head, x: torch.Tensor) -> bool:
    """A frozen, bias-free LoRALinear head on the GPU whose shapes the matrix-core GEMM takes."""
    if not (isinstance(head, LoRALinear) and head.bias is None and head._frozen()):
        return False
    if not (x.is_cuda and x.dtype == torch.float32 and x.dim() >= 2 and not x.is_inference()):
        return False
    x2 = x.reshape(-1, x.size(-1))
    rank = head.lora.left.weight.size(1)
    return (_mfma_linear_usable(x2, head.weight, rank) and x2.size(1) % 4 == 0
            and head.lora.left.weight.requires_grad and head.lora.right.weight.requires_grad)


def lm_head_loss(head, x: torch.Tensor, target: torch.Tensor, ignore_index: int = -100):
    """mean cross-entropy of head(x) against target (nn.CrossEntropyLoss defaults), fused."""
    return _FrozenLoRAHeadLoss.apply(x, head.weight, head.lora.left.weight, head.lora.right.weight,
                                     target, ignore_index)
