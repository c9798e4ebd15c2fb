"""Routed FFN on the MI355X: device-side bucketing + grouped MFMA GEMMs.

Used by the routed FFN layers when the activations live on the GPU and the base
weights are frozen (the LoRA variants the upgrader installs: the hot path).  Nothing
here reads a bucket size on the host: the (token, block) pairs are sorted by block with
device ops, bucket boundaries stay in a device int32 ``offsets`` tensor, and every
block GEMM of a layer is ONE launch of ``spt_grouped_gemm`` (fp32 MFMA,
``spt-proto_amd/csrc/grouped_gemm.hip``) -- compare the reference's loop of boolean-mask
gathers and per-block cuBLAS calls (``layers/tuning/lora_ffn.py:87-111``).

Row space: P = T * k rows, row p = the p-th (token, block) pair in block-major order
(stable in the token id, i.e. the reference's ``x[mask]`` order inside each block).
"""
from typing import NamedTuple

import torch

from naive_gpt import ext


class Buckets(NamedTuple):
    token: torch.Tensor     # [P] int32: token id of each row
    token_long: torch.Tensor
    block: torch.Tensor     # [P] int64: block id of each row (ascending)
    offsets: torch.Tensor   # [G + 1] int32, device
    coeff: torch.Tensor     # [P] router probability of (token, block), differentiable
    n_blocks: int


def make_buckets(prob: torch.Tensor, k: int) -> Buckets:
    """prob [T, G] router probabilities -> block-sorted (token, block) rows."""
    n_tokens, n_blocks = prob.shape
    indices = torch.topk(prob, k=k, dim=-1, sorted=False).indices       # [T, k]
    block = indices.reshape(-1)
    order = torch.argsort(block, stable=True)
    token_long = torch.div(order, k, rounding_mode='floor')
    block_sorted = block.index_select(0, order)
    counts = torch.bincount(block_sorted, minlength=n_blocks)
    offsets = torch.zeros([n_blocks + 1], dtype=torch.int32, device=prob.device)
    offsets[1:] = torch.cumsum(counts, dim=0)
    coeff = prob.reshape(-1).index_select(0, token_long * n_blocks + block_sorted)
    return Buckets(token=token_long.int(), token_long=token_long, block=block_sorted,
                   offsets=offsets, coeff=coeff, n_blocks=n_blocks)


class GroupedLinear(torch.autograd.Function):
    """out[p] = rowscale[p] * (a[gather[p]] @ W_g(p)^T + bias[g(p)]) for frozen weights.

    ``view`` = (n, k, group_stride, ldn, ldk) describes W_g(n, kk) inside ``weight``'s
    storage.  Gradients flow to ``a`` (through the opposite-orientation kernel and, when
    rows were gathered, an ``index_add_``) and to ``rowscale``."""

    @staticmethod
    def forward(ctx, a, weight, offsets, view, gather, gather_long, bias, rowscale):
        n, k, gstride, ldn, ldk = view
        n_groups = offsets.numel() - 1
        out = ext.grouped_gemm(a, weight, offsets, n_groups, n, k, gstride, ldn, ldk,
                               gather=gather, bias=bias, rowscale=rowscale)
        ctx.view = view
        ctx.a_rows = a.size(0)
        ctx.save_for_backward(weight, offsets, gather_long, rowscale, out)
        return out

    @staticmethod
    def backward(ctx, grad_out):
        weight, offsets, gather_long, rowscale, out = ctx.saved_tensors
        n, k, gstride, ldn, ldk = ctx.view
        n_groups = offsets.numel() - 1
        grad_out = grad_out.contiguous()
        grad_a = grad_scale = None
        if ctx.needs_input_grad[7] and rowscale is not None:
            # out = s * u  ->  ds = <dout, u> = <dout, out> / s   (s = 2 sigmoid(.) > 0)
            grad_scale = (grad_out * out).sum(dim=-1) / rowscale
        if ctx.needs_input_grad[0]:
            # d a[src(p), kk] = s[p] * sum_n dout[p, n] W_g(n, kk): roles of n and k swap
            rows = ext.grouped_gemm(grad_out, weight, offsets, n_groups, k, n, gstride,
                                    ldk, ldn, rowscale=rowscale)
            if gather_long is None:
                grad_a = rows
            else:
                grad_a = torch.zeros([ctx.a_rows, k], dtype=rows.dtype, device=rows.device)
                grad_a.index_add_(0, gather_long, rows)
        if ctx.needs_input_grad[1]:
            raise NotImplementedError('GroupedLinear: weight gradients (frozen base only)')
        return grad_a, None, None, None, None, None, None, grad_scale


def grouped_linear(a, weight, offsets, view, gather=None, gather_long=None, bias=None,
                   rowscale=None):
    return GroupedLinear.apply(a, weight, offsets, view, gather, gather_long, bias, rowscale)


def usable(x: torch.Tensor, *frozen_weights) -> bool:
    """The grouped-GEMM path applies to fp32 CUDA activations and frozen base weights."""
    if not x.is_cuda or x.dtype != torch.float32:
        return False
    for w in frozen_weights:
        if w is None:
            continue
        if w.requires_grad or not w.is_contiguous() or w.dtype != torch.float32:
            return False
    return True
