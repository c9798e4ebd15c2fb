"""Norm outputs that are recomputed in the backward instead of kept.

In a pre-norm block the output of ``norm1`` is read by the q / k / v projections and the
output of ``norm2`` by the router and the routed FFN -- every one of them a LoRA layer whose
backward needs its INPUT (for the gradient of the ``left`` table), so each norm output
(``[N * S, d]`` fp32: 33.5 MB at BERT-large dimensions, micro-batch 16) stays alive from the
forward to the backward: 1.6 GB of the 24-layer model's peak.  The norm's own backward keeps
its INPUT anyway, and the output is one cheap elementwise pass away from it (17 us).

``TransformerBlock.forward`` tags a norm output with where it came from (``tag``); the LoRA
layers' autograd functions (``lora._FrozenLoRALinear``, ``grouped.RoutedLoRAFFN`` ...) save
the tagged tensor's origin instead of the tensor and call ``output`` in their backward, which
recomputes it once and hands the same tensor to the consumers that follow (q, k, v).

No reference counterpart: the reference keeps every activation (plain autograd).
"""
import torch
from torch import nn

_RECOMPUTABLE = (nn.LayerNorm,)


class Origin:
    __slots__ = ('module', 'input')

    def __init__(self, module, input):
        self.module, self.input = module, input


def register(cls):
    """Mark a norm class as a pure function of its input and parameters."""
    global _RECOMPUTABLE
    if cls not in _RECOMPUTABLE:
        _RECOMPUTABLE = _RECOMPUTABLE + (cls,)
    return cls


def tag(output: torch.Tensor, module: nn.Module, input: torch.Tensor) -> torch.Tensor:
    # A forward is under way: whatever `output()` recomputed last belongs to an earlier
    # backward.  Its key (module, address, version) could match a NEW activation -- the
    # allocator hands the same addresses out step after step -- so it is dropped here; inside
    # one backward the saved input is alive and its address cannot be reused.
    release()
    if torch.is_grad_enabled() and input.requires_grad and isinstance(module, _RECOMPUTABLE) \
            and input.is_cuda and not input.is_inference():
        output._spt_origin = Origin(module, input)
    return output


def origin_of(t: torch.Tensor):
    return getattr(t, '_spt_origin', None)


_last = [None, None]          # (key, tensor): the norm output most recently recomputed


def output(module: nn.Module, input: torch.Tensor) -> torch.Tensor:
    key = (id(module), input.data_ptr(), input._version, tuple(input.shape))
    if _last[0] == key:
        return _last[1]
    with torch.no_grad():
        # the same kernel as the forward took (basic/fused_norm.py), so the same bits
        from ..basic import fused_norm
        out = fused_norm.add_norm(module, input, None)[1]
    _last[0], _last[1] = key, out
    return out


def release():
    _last[0] = _last[1] = None


class RecomputedLinear(torch.autograd.Function):
    """y = x W^T + b for a SMALL trainable linear (the FFN router, d -> 4) whose input is a
    tagged norm output: saves the norm's input instead of x.  (Used by the routed FFN's composed
    path; the fused function holds its router inside its own node: sparse/grouped.py.)"""

    @staticmethod
    def forward(ctx, x, origin_input, weight, bias, module):
        ctx.module = module
        ctx.x_shape = x.shape
        ctx.save_for_backward(origin_input, weight)
        ctx.has_bias = bias is not None
        return nn.functional.linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dy):
        origin_input, weight = ctx.saved_tensors
        x = output(ctx.module, origin_input).reshape(-1, weight.size(1))
        dy2 = dy.reshape(-1, weight.size(0))
        grad_x = None
        if ctx.needs_input_grad[0]:
            grad_x = torch.matmul(dy2, weight).view(ctx.x_shape)
        grad_w = None
        if ctx.needs_input_grad[2]:
            # [out, tokens] x [tokens, d]: as one GEMM a handful of workgroups (39 us for 4 x 8192 x
            # 1024); split over the tokens it fills the GPU (lora.tall_tn)
            from .lora import tall_tn
            grad_w = tall_tn(dy2.contiguous(), x)
        grad_b = dy2.sum(dim=0) if ctx.has_bias and ctx.needs_input_grad[3] else None
        return grad_x, None, grad_w, grad_b, None
