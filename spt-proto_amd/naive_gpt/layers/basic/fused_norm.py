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
summands.  Modules other than a plain affine ``nn.LayerNorm`` (LLaMA's RMSNorm), widths the kernels
do not take and CPU tensors run the same expression op by op.
"""
import torch
from torch import nn


class _AddLayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, r, gamma, beta, eps: float):
        from naive_gpt import ext
        s, y, mean, rstd = ext.add_layernorm_forward(x.contiguous(), r.contiguous(), gamma, beta, eps)
        ctx.save_for_backward(s, gamma, mean, rstd)
        return s, y

    @staticmethod
    def backward(ctx, ds, dy):
        from naive_gpt import ext
        s, gamma, mean, rstd = ctx.saved_tensors
        if dy is None:                                   # (the normalised output was not used)
            return ds, ds, None, None, None
        dx, dgamma, dbeta = ext.layernorm_backward(s, dy.contiguous(), gamma, mean, rstd,
                                                   dskip=None if ds is None else ds.contiguous())
        return dx, dx, dgamma, dbeta, None


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps: float):
        from naive_gpt import ext
        x = x.contiguous()
        _, y, mean, rstd = ext.add_layernorm_forward(x, None, gamma, beta, eps)
        ctx.save_for_backward(x, gamma, mean, rstd)
        return y

    @staticmethod
    def backward(ctx, dy):
        from naive_gpt import ext
        x, gamma, mean, rstd = ctx.saved_tensors
        dx, dgamma, dbeta = ext.layernorm_backward(x, dy.contiguous(), gamma, mean, rstd)
        return dx, dgamma, dbeta, None


def usable(norm: nn.Module, x: torch.Tensor) -> bool:
    if not (type(norm) is nn.LayerNorm and norm.elementwise_affine and norm.bias is not None
            and len(norm.normalized_shape) == 1):
        return False
    from naive_gpt import ext
    return ext.layernorm_supported(x, norm.normalized_shape[0]) and not x.is_inference()


def add_norm(norm: nn.Module, x: torch.Tensor, r):
    """(x + r, norm(x + r)); r None: (x, norm(x))."""
    if usable(norm, x):
        if r is None:
            return x, _LayerNorm.apply(x, norm.weight, norm.bias, norm.eps)
        return _AddLayerNorm.apply(x, r, norm.weight, norm.bias, norm.eps)
    s = x if r is None else x + r
    return s, norm(s)
