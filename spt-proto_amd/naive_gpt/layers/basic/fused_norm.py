"""The residual addition in front of a LayerNorm, the norm, and -- in the backward -- the skip
path's gradient joining the norm's, as one function (csrc/layernorm.hip).

Reference wiring: ``naive_gpt/layers/basic/transformer.py:46-52`` (pre-norm):
``x = x + mha(norm1(x)); x = x + ffd(norm2(x))``.  Written over (stream, addend) pairs every
addition of the residual stream sits directly in front of a norm::

    x, h = add_norm(norm1, x, pending)      # x = x + pending;  h = norm1(x)
    x, h = add_norm(norm2, x, mha(h))
    pending = ffd(h)                        # added by the next block's norm1 (or the final norm)

so the forward reads the two summands once and writes the sum and the normalised row, and the
backward's ``d_sum = d_stream + dLN(d_h)`` is one kernel whose result is the gradient of BOTH
summands.  A plain affine ``nn.LayerNorm`` and LLaMA's ``LlamaRMSNorm`` take the kernels; other modules, widths
the kernels do not take and CPU tensors run the same expression op by op.
"""
import torch
from torch import nn


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps: float, rms: bool):
        from naive_gpt import ext
        s, y, mean, rstd = ext.add_layernorm_forward(x.contiguous(), r.contiguous(), gamma, beta, eps, rms)
        ctx.save_for_backward(s, gamma, mean, rstd)
        ctx.rms = rms
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        from naive_gpt import ext
        s, gamma, mean, rstd = ctx.saved_tensors
        if dy is None:                                   # (the normalised output was not used)
            return ds, ds, None, None, None, None
        dx, dgamma, dbeta = ext.layernorm_backward(s, dy.contiguous(), gamma, mean, rstd,
                                                   dskip=None if ds is None else ds.contiguous(),
                                                   rms=ctx.rms)
        return dx, dx, dgamma, None if ctx.rms else dbeta, None, None


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float, rms: bool):
        from naive_gpt import ext
        x = x.contiguous()
        _, y, mean, rstd = ext.add_layernorm_forward(x, None, gamma, beta, eps, rms)
        ctx.save_for_backward(x, gamma, mean, rstd)
        ctx.rms = rms
        return y

    @staticmethod
    def backward(ctx, dy):
        from naive_gpt import ext
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = ext.layernorm_backward(x, dy.contiguous(), gamma, mean, rstd, rms=ctx.rms)
        return dx, dgamma, None if ctx.rms else dbeta, None, None


def _kind(norm: nn.Module):
    """(gamma, beta | None, eps, rms) of a norm the kernels implement, else None."""
    from .utils import LlamaRMSNorm
    if type(norm) is nn.LayerNorm and norm.elementwise_affine and norm.bias is not None \
            and len(norm.normalized_shape) == 1:
        return norm.weight, norm.bias, norm.eps, False
    if type(norm) is LlamaRMSNorm and norm.weight.dtype == torch.float32:
        return norm.weight, None, norm.variance_epsilon, True
    return None


def usable(norm: nn.Module, x: torch.Tensor) -> bool:
    kind = _kind(norm)
    if kind is None:
        return False
    from naive_gpt import ext
    return ext.layernorm_supported(x, kind[0].numel(), kind[3]) and not x.is_inference()


def add_norm(norm: nn.Module, x: torch.Tensor, r):
    """(x + r, norm(x + r)); r None: (x, norm(x))."""
    from naive_gpt import ext
    if usable(norm, x):
        ext.note_path('norm', 'kernel')
        gamma, beta, eps, rms = _kind(norm)
        if r is None:
            return x, _LayerNorm.apply(x, gamma, beta, eps, rms)
        return _AddLayerNorm.apply(x, r, gamma, beta, eps, rms)
    ext.note_path('norm', 'library', fallback=x.is_cuda and not x.is_inference(),
                  why=lambda: '{} on x {} {}'.format(type(norm).__name__, tuple(x.shape), x.dtype))
    s = x if r is None else x + r
    return s, norm(s)
