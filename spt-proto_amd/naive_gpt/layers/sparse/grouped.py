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
import os
from typing import NamedTuple

import torch

from naive_gpt import ext
from naive_gpt.layers.tuning import recompute


class Buckets(NamedTuple):
    token: torch.Tensor     # [P] int32: token id of each row
    token_long: torch.Tensor
    block: torch.Tensor     # [P] int64: block id of each row (ascending)
    offsets: torch.Tensor   # [G + 1] int32, device
    coeff: torch.Tensor     # [P] router probability of (token, block), differentiable
    n_blocks: int
    pos: torch.Tensor       # [T, k] int32: the rows of each token (inverse of `token`)


class _RouteTopK(torch.autograd.Function):
    """prob [T, G] -> the block-sorted rows AND their coefficients scale * prob[token, block], one
    launch (routing.hip); backward: the coefficients' gradient scattered back, one launch.
    (As torch operators: two int64 copies, two index computations, a gather and a product
    forward; a product, a zero fill and an index_add backward -- x 24 layers.)"""

    @staticmethod
    def forward(ctx, prob, k, scale):
        token, block, offsets, pos, token_long, block_long, coeff = \
            ext.route_topk_coeff(prob.contiguous(), k, scale)
        ctx.save_for_backward(pos, block)
        ctx.scale, ctx.n_blocks = scale, prob.size(1)
        ctx.mark_non_differentiable(token, block, offsets, pos, token_long, block_long)
        # (or autograd hands the backward a zero tensor for each of the six index outputs: six fills)
        ctx.set_materialize_grads(False)
        return token, block, offsets, pos, token_long, block_long, coeff

    @staticmethod
    def backward(ctx, *grads):
        if grads[6] is None:
            return None, None, None
        pos, block = ctx.saved_tensors
        return ext.route_coeff_backward(grads[6].contiguous(), pos, block, ctx.scale,
                                        ctx.n_blocks), None, None


def make_buckets(prob: torch.Tensor, k: int, scale: float = 1.0) -> Buckets:
    """prob [T, G] router probabilities -> block-sorted (token, block) rows; `coeff` =
    scale * the rows' probabilities."""
    n_tokens, n_blocks = prob.shape
    if prob.is_cuda and prob.dtype == torch.float32 \
            and ext.route_topk_supported(n_tokens, n_blocks):
        # one launch (routing.hip) instead of topk + argsort + bincount + cumsum + gathers
        token, _, offsets, pos, token_long, block_long, coeff = _RouteTopK.apply(prob, k, float(scale))
        ext.note_path('route_topk', 'kernel')
        return Buckets(token=token, token_long=token_long, block=block_long, offsets=offsets,
                       coeff=coeff, n_blocks=n_blocks, pos=pos)
    ext.note_path('route_topk', 'torch', fallback=prob.is_cuda,
                  why=lambda: 'prob {} {}: spt_route_topk takes fp32, <= 8 blocks, <= 65536 tokens'.format(
                      tuple(prob.shape), prob.dtype))
    indices = torch.topk(prob, k=k, dim=-1, sorted=False).indices       # [T, k]
    block = indices.reshape(-1)
    order = torch.argsort(block, stable=True)
    token_long = torch.div(order, k, rounding_mode='floor')
    block_sorted = block.index_select(0, order)
    counts = torch.bincount(block_sorted, minlength=n_blocks)
    offsets = torch.zeros([n_blocks + 1], dtype=torch.int32, device=prob.device)
    offsets[1:] = torch.cumsum(counts, dim=0)
    coeff = prob.reshape(-1).index_select(0, token_long * n_blocks + block_sorted)
    if scale != 1.0:
        coeff = scale * coeff
    # row p holds the flat (token, choice) pair order[p]: invert the permutation
    pos = torch.empty_like(order)
    pos[order] = torch.arange(order.numel(), device=order.device)
    return Buckets(token=token_long.int(), token_long=token_long, block=block_sorted,
                   offsets=offsets, coeff=coeff, n_blocks=n_blocks,
                   pos=pos.view(n_tokens, k).int())


COEFF_FLOOR = 1e-12
# h (one reader) goes to its GEMM as fp32 rows, no image: -0.36 ms per configs[2] step, A/B on one
# box (the same for x: +0.26 ms, for dy: +-0: both keep their images)
H_A32 = os.environ.get('SPT_H_A32', '1') != '0'
ROUTER_RIDES = os.environ.get('SPT_ROUTER_RIDES', '1') != '0'


def _floor(coeff: torch.Tensor) -> torch.Tensor:
    """The coefficient gradients below are recovered as (a dot that contains c) / c, where the
    reference differentiates c * (x W^T + b) directly.  c = 2 sigmoid(logit) underflows to 0
    for logit < -104 (0 / 0 = NaN here, a finite number there) and the quotient amplifies the
    fp32 cancellation error of the numerator by 1 / c.  Both are harmless AFTER the chain
    rule -- d c / d logit = c (1 - c / 2), so the factor 1 / c cancels and what reaches the
    router is (numerator error) * (1 - c / 2) -- provided the division itself stays finite:
    hence a floor far below any coefficient that still has a gradient in fp32."""
    return coeff.clamp_min(COEFF_FLOOR)


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
            grad_scale = (grad_out * out).sum(dim=-1) / _floor(rowscale)
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


# ---------------------------------------------------------------- the fused LoRA routed FFN

def activation_code(module):
    """ext activation id of an nn activation module, or None when the fused epilogue does
    not implement it (the caller then composes the layer from grouped_linear + torch)."""
    from torch import nn
    if isinstance(module, nn.ReLU):
        return ext.ACT_RELU
    if isinstance(module, nn.GELU) and getattr(module, 'approximate', 'none') == 'none':
        return ext.ACT_GELU
    if isinstance(module, nn.SiLU):
        return ext.ACT_SILU
    return None


def _own_block(wide: torch.Tensor, block: torch.Tensor, n_blocks: int) -> torch.Tensor:
    """[P, nb * r] -> [P, r]: row p keeps the r columns of its own block."""
    rows, r = wide.size(0), wide.size(1) // n_blocks
    index = block.view(rows, 1, 1).expand(rows, 1, r)
    return wide.view(rows, n_blocks, r).gather(1, index).squeeze(1)


def _in_own_block(narrow: torch.Tensor, block: torch.Tensor, n_blocks: int) -> torch.Tensor:
    """[P, r] -> [P, nb * r], zero outside the row's own block (adjoint of _own_block)."""
    rows, r = narrow.shape
    wide = torch.zeros([rows, n_blocks, r], dtype=narrow.dtype, device=narrow.device)
    wide.scatter_(1, block.view(rows, 1, 1).expand(rows, 1, r), narrow.unsqueeze(1))
    return wide.view(rows, n_blocks * r)


def _block_major(cat: torch.Tensor, n_blocks: int) -> torch.Tensor:
    """[bs, nb * r] -> [nb * bs, r] (the layout of a LoRA table over d_feedforward)."""
    bs = cat.size(0)
    return cat.view(bs, n_blocks, -1).permute(1, 0, 2).reshape(n_blocks * bs, -1)


def _block_cat(table: torch.Tensor, n_blocks: int) -> torch.Tensor:
    """[nb * bs, r] -> [bs, nb * r]: all blocks' slices side by side."""
    bs = table.size(0) // n_blocks
    return table.view(n_blocks, bs, -1).permute(1, 0, 2).reshape(bs, -1)


def _down(a: torch.Tensor, table: torch.Tensor, want_image: bool, want_norms: bool):
    """(a @ table, split image of a | None, row norms of a | None): one pass over `a`
    (ext.lora_down) where its shapes allow, else the separate operators.  `want_norms` is asked
    for in front of a ReLU GEMM, and there the product is formed in exact fp32 (csrc/lora_side.hip:
    its error would otherwise pick the sign of pre-activations next to zero)."""
    if ext.lora_down_supported(a, table):
        ext.note_path('lora_down', 'kernel')
        if want_image or want_norms:
            return ext.lora_down(a, table, want_image=want_image, want_norms=want_norms, exact=want_norms)
        return ext.lora_down(a, table), None, None
    ext.note_path('lora_down', 'library', fallback=a.is_cuda,
                  why=lambda: 'a {} x table {}: spt_lora_down takes K % 32 == 0, 16-64 columns'.format(
                      tuple(a.shape), tuple(table.shape)))
    return (torch.matmul(a, table), ext.split_bf16(a) if want_image else None,
            ext.row_norms(a) if want_norms else None)


def _down_blocks(a: torch.Tensor, table: torch.Tensor, bk: Buckets, want_image: bool):
    """(a[p] @ table_g(p) [P, r], split image of a | None) for a per-block table [nb * bs, r] and
    rows sorted by block: one grouped pass (ext.lora_down_grouped) instead of the product with all
    blocks' tables side by side, a gather of the row's own block and a copy."""
    nb = bk.n_blocks
    tables = table.view(nb, table.size(0) // nb, table.size(1))
    if ext.lora_down_grouped_supported(a, tables):
        ext.note_path('lora_down_grouped', 'kernel')
        if want_image:
            return ext.lora_down_grouped(a, tables, bk.offsets, want_image=True)
        return ext.lora_down_grouped(a, tables, bk.offsets), None
    ext.note_path('lora_down_grouped', 'side_by_side', fallback=a.is_cuda,
                  why=lambda: 'a {} x tables {}'.format(tuple(a.shape), tuple(tables.shape)))
    wide, image, _ = _down(a, _block_cat(table, nb), want_image, False)
    return _own_block(wide, bk.block, nb).contiguous(), image


def _tn(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    from naive_gpt.layers.tuning.lora import tall_tn
    return tall_tn(a, b)


def _tn_blocks(wide: torch.Tensor, per_token: torch.Tensor, bk: Buckets) -> torch.Tensor:
    """The gradient of a per-block LoRA table [nb * bs, r]: for block g, wide[rows of g]^T .
    per_token[token of those rows] -- rows are sorted by block, so it is a grouped product
    (spt_tall_tn with the bucket offsets and the rows' token ids as the gather) instead of a
    [P, nb * r] matrix of mostly zeros times wide."""
    nb = bk.n_blocks
    if ext.tall_tn_supported(wide, per_token):
        ext.note_path('tall_tn_blocks', 'kernel')
        return ext.tall_tn(wide, per_token, gather=bk.token, offsets=bk.offsets).view(nb * wide.size(1), -1)
    ext.note_path('tall_tn_blocks', 'library', fallback=wide.is_cuda,
                  why=lambda: 'wide {} x per_token {}'.format(tuple(wide.shape), tuple(per_token.shape)))
    rows = per_token.index_select(0, bk.token_long)
    return _block_major(_tn(wide, _in_own_block(rows, bk.block, nb)), nb)


def _tn_blocks_many(pairs, bk: Buckets):
    """[_tn_blocks(wide, per_token, bk) for the pairs]: one pair of launches when they have one shape"""
    wides, narrows = [p[0] for p in pairs], [p[1] for p in pairs]
    if ext.tall_tn_batchable(wides, narrows):
        ext.note_path('tall_tn_blocks', 'kernel_batch')
        outs = ext.tall_tn_batch(wides, narrows, gather=bk.token, offsets=bk.offsets)
        return [t.view(bk.n_blocks * wides[0].size(1), -1) for t in outs]
    return [_tn_blocks(w, t, bk) for w, t in pairs]


def _router_rides(x: torch.Tensor, l1: torch.Tensor, rw: torch.Tensor, n_blocks: int) -> bool:
    """The router's logits as a second table of the x L1 pass (spt_lora_down2) and the routing from
    logits (spt_route_topk_logits): rank-16 adapter, <= 8 blocks, shapes spt_lora_down takes."""
    return (ROUTER_RIDES and x.is_cuda and l1.size(1) == 16 and n_blocks <= 8 and rw.is_contiguous()
            and rw.dtype == torch.float32 and rw.data_ptr() % 16 == 0
            and ext.lora_down_supported(x, l1) and ext.route_topk_supported(x.size(0), n_blocks))


def _route(x: torch.Tensor, rw: torch.Tensor, rb, n_blocks: int, k: int):
    """The router (sparse/feedforward.py:22-25: Linear + Sigmoid) and the bucketing of its top-k, inside
    the routed FFN's own autograd node: -> (prob [T, nb], Buckets with coeff = 2 prob
    (lora_ffn.py:96), block ids as int32 [P])."""
    prob = torch.sigmoid(torch.nn.functional.linear(x, rw, rb))
    if ext.route_topk_supported(prob.size(0), n_blocks):
        ext.note_path('route_topk', 'kernel')
        token, block, offsets, pos, token_long, block_long, coeff = ext.route_topk_coeff(prob, k, 2.0)
        return prob, Buckets(token=token, token_long=token_long, block=block_long, offsets=offsets,
                             coeff=coeff, n_blocks=n_blocks, pos=pos), block
    bk = make_buckets(prob, k, scale=2.0)
    return prob, bk, bk.block.int()


def _route_backward(ctx, grad_coeff, prob, bk: Buckets, block, x, rw, grad_x):
    """The router's share of the backward: d coeff [P] -> d logit [T, nb] (one launch: the adjoint of
    the bucketing chained through the sigmoid), its input gradient ADDED into `grad_x` (the FFN's
    own, same tensor: one sum inside one autograd node), and the gradients of its weight and bias.
    -> (grad_rw, grad_rb)."""
    from naive_gpt.layers.tuning.lora import tall_tn
    nb = bk.n_blocks
    if grad_coeff.is_cuda and nb <= 8:
        dlogit = ext.route_coeff_backward(grad_coeff.contiguous(), bk.pos, block, 2.0, nb, prob=prob)
    else:
        dprob = torch.zeros_like(prob)
        dprob.view(-1).index_put_((bk.token_long * nb + bk.block,), 2.0 * grad_coeff)
        dlogit = dprob * (1.0 - prob) * prob
    grad_x.addmm_(dlogit, rw)
    grad_rw = tall_tn(dlogit, x) if ctx.needs_input_grad[1] else None       # [nb, d]
    grad_rb = dlogit.sum(dim=0) if ctx.needs_input_grad[2] else None
    return grad_rw, grad_rb


class RoutedLoRAFFN(torch.autograd.Function):
    """y = LoRARoutedFFN(x) for frozen base weights (reference formula:
    naive_gpt/layers/tuning/lora_ffn.py:87-111), as four fused block GEMMs:

        h  = act(c (x W1_g^T + b1_g) + (x L1) R1_g^T)          spt_grouped_gemm_fused, EPI_ACT
        ys = c (h W2_g) + (h L2_g) R2^T                        spt_grouped_gemm_fused
        y  = b2 + sum over the token's rows of ys              spt_rows_combine
      backward
        ds = (c (dy W2_g^T) + (dy R2) L2_g^T) * act'(s)        EPI_DACT, + the two row dots
        dx = sum over the token's rows of c (ds W1_g) + (ds R1_g) L1^T

    Saved for backward: h (and the pre-activation only when act is not ReLU), the two
    [*, r] LoRA intermediates -- against ~8 [P, d] tensors of the op-by-op composition.
    The small [*, r] products and the LoRA table gradients are single calls of libspt_hip too
    (spt_lora_down[_grouped], spt_tall_tn).
    """

    @staticmethod
    def forward(ctx, x, rw, rb, l1, r1, l2, r2, w1, b1, w2, b2, nb: int, top_k: int, act: int,
                origin_input=None, origin_module=None):
        d_ff, d = w1.shape
        x = x.contiguous()
        bs, rank = d_ff // nb, l1.size(1)
        relu = act == ext.ACT_RELU
        imgs = _images_usable(x, w1, w2, d, bs)
        if _router_rides(x, l1, rw, nb):
            # ONE pass over x: u = x L1, the router's logits x W_r^T (a second table of the same
            # launch, exact fp32: the top-k must not depend on a split's 2^-16), x's image and row
            # norms; the sigmoid is formed inside the routing launch.  (As torch operators: a library
            # GEMM with 4 output columns and a sigmoid pass, ~20 us per layer.)
            ext.note_path('router', 'rides_lora_down')
            pack = ext.lora_down(x, l1, want_image=imgs, want_norms=relu, block_major=True, exact=True,
                                 table2=rw)
            pack, x_img, x_norm = pack if isinstance(pack, tuple) else (pack, None, None)
            u = pack[0]
            prob, token, block, offsets, pos, token_long, block_long, coeff = \
                ext.route_topk_logits(pack[1], rb, nb, top_k, 2.0)
            ext.note_path('route_topk', 'kernel')
            bk = Buckets(token=token, token_long=token_long, block=block_long, offsets=offsets,
                         coeff=coeff, n_blocks=nb, pos=pos)
        else:
            ext.note_path('router', 'library', fallback=x.is_cuda,
                          why=lambda: 'x {} L1 {} router {}'.format(tuple(x.shape), tuple(l1.shape), tuple(rw.shape)))
            prob, bk, block = _route(x, rw, rb, nb, top_k)
            # Both block GEMMs run from pre-split images: the weights' are made from the frozen
            # parameters, x's and h's cost one pass each -- the pass that also forms the LoRA down
            # product and, in front of a ReLU, the row norms the kernel wants (include/spt_hip.h:
            # a_norm): ext.lora_down, one read of the activation for all three.
            u, x_img, x_norm = _down(x, l1, imgs, relu)                      # [T, r]
        coeff = bk.coeff
        rows = bk.token.numel()
        h, s = ext.grouped_gemm_fused(
            x, w1, bk.offsets, nb, n=bs, k=d, w_group_stride=bs * d, w_ldn=d, w_ldk=1,
            n_rows=rows, gather=bk.token, bias=b1, rowscale=coeff,
            a2=u, gather2=bk.token, b2=r1, b2_group_stride=bs * rank,
            epilogue=ext.EPI_ACT, activation=act, keep_preact=not relu,
            a_image=x_img, w_image=ext.weight_image(w1) if imgs else None,
            a_norm=x_norm, w_norm=ext.weight_row_norms(w1) if relu else None)
        # (h has ONE reader: no image of it -- the GEMM splits its fp32 rows itself ("A32", +11 us)
        # and the down product runs without the image store (-20 us))
        z, h_img = _down_blocks(h, l2, bk, imgs and not H_A32)               # [P, r]
        ys = ext.grouped_gemm_fused(
            h, w2, bk.offsets, nb, n=d, k=bs, w_group_stride=bs, w_ldn=d_ff, w_ldk=1,
            n_rows=rows, rowscale=coeff, a2=z, b2=r2, b2_group_stride=0,
            a_image=h_img, w_image=ext.weight_image(w2) if imgs else None)
        y = ext.rows_combine(ys, bk.pos, bias=b2)
        ctx.bk, ctx.act = bk, act
        # x is a norm output the block can recompute: keep its origin instead (tuning/recompute.py)
        ctx.origin_module = origin_module
        ctx.save_for_backward(x if origin_module is None else origin_input, coeff, u, h, s, z,
                              l1, r1, l2, r2, w1, w2, prob, block, rw)
        return y

    @staticmethod
    def backward(ctx, dy):
        x, coeff, u, h, s, z, l1, r1, l2, r2, w1, w2, prob, block, rw = ctx.saved_tensors
        if ctx.origin_module is not None:
            x = recompute.output(ctx.origin_module, x).view(-1, w1.size(1))
        bk, act = ctx.bk, ctx.act
        nb = bk.n_blocks
        d_ff, d = w1.shape
        bs, rank, rows = d_ff // nb, l1.size(1), bk.token.numel()
        dy = dy.contiguous()
        imgs = _images_usable(dy, w1, w2, d, bs)
        dzt, dy_img, _ = _down(dy, r2, imgs, False)                          # [T, r]
        ds, dot_main, dot_act = ext.grouped_gemm_fused(
            dy, w2, bk.offsets, nb, n=bs, k=d, w_group_stride=bs, w_ldn=1, w_ldk=d_ff,
            n_rows=rows, gather=bk.token, rowscale=coeff,
            a2=dzt, gather2=bk.token, b2=l2, b2_group_stride=bs * rank,
            epilogue=ext.EPI_DACT, activation=act, h_in=h, s_in=s,
            a_image=dy_img, w_image=ext.weight_image(w2) if imgs else None, raw_dots=True)
        du, _ = _down_blocks(ds, r1, bk, False)                              # [P, r]
        # d/dc = <dy W2_g^T, h> + <ds, x W1_g^T + b1_g>.  The kernel's dot_main = <v, h> with
        # v = c (dy W2_g^T) + dzt L2_g^T (the value before act'), dot_act = <ds, s> with
        # s = c (x W1^T + b1) + u R1^T; the LoRA parts of both are [*, r] dots:
        # <dzt L2_g^T, h> = <dzt, z> and <ds, u R1_g^T> = <du, u>
        # (dot_main + dot_act - <du, u[token]> - <dzt[token], z>) / _floor(coeff), one launch
        grad_coeff = ext.ffn_coeff_grad(dot_main, dot_act, du, u, dzt, z, bk.token, coeff, COEFF_FLOOR)
        # LoRA tables (before the last GEMM: ds [P, bs] is released as soon as that has read it --
        # the step's peak memory is here, with h, ds, dxs and grad_x alive)
        du_tok = ext.rows_combine(du, bk.pos)                                # [T, r]: du[pos].sum(1)
        # (two pairs of equal shape: one pair of launches each)
        from naive_gpt.layers.tuning.lora import tall_tn_many
        grad_l1, grad_r2 = tall_tn_many([(x, du_tok), (dy, ext.rows_combine(z, bk.pos))])
        grad_r1, grad_l2 = _tn_blocks_many([(ds, u), (h, dzt)], bk)
        del dot_main, dot_act, du_tok, dzt
        dxs = ext.grouped_gemm_fused(
            ds, w1, bk.offsets, nb, n=d, k=bs, w_group_stride=bs * d, w_ldn=1, w_ldk=d,
            n_rows=rows, rowscale=coeff, a2=du, b2=l1, b2_group_stride=0)
        del ds, du
        if grad_coeff.is_cuda and nb <= 8 and d % 4 == 0:
            # the router's share d logit . W_r of grad_x inside the un-bucketing pass
            dlogit = ext.route_coeff_backward(grad_coeff.contiguous(), bk.pos, block, 2.0, nb, prob=prob)
            grad_x = ext.rows_combine(dxs, bk.pos, side=dlogit, side_weight=rw.contiguous())
            del dxs
            from naive_gpt.layers.tuning.lora import tall_tn
            grad_rw = tall_tn(dlogit, x) if ctx.needs_input_grad[1] else None   # [nb, d]
            grad_rb = dlogit.sum(dim=0) if ctx.needs_input_grad[2] else None
        else:
            grad_x = ext.rows_combine(dxs, bk.pos)
            del dxs
            grad_rw, grad_rb = _route_backward(ctx, grad_coeff, prob, bk, block, x, rw, grad_x)
        return (grad_x, grad_rw, grad_rb, grad_l1, grad_r1, grad_l2, grad_r2,
                None, None, None, None, None, None, None, None, None)


def routed_lora_ffn(x, rw, rb, l1, r1, l2, r2, w1, b1, w2, b2, n_blocks: int, top_k: int, act: int,
                    origin=None):
    """`rw`, `rb`: the router's Linear (its Sigmoid, the top-k and the bucketing run inside)."""
    if origin is None:
        return RoutedLoRAFFN.apply(x, rw, rb, l1, r1, l2, r2, w1, b1, w2, b2, n_blocks, top_k, act)
    return RoutedLoRAFFN.apply(x, rw, rb, l1, r1, l2, r2, w1, b1, w2, b2, n_blocks, top_k, act,
                               origin.input, origin.module)


def _images_usable(a: torch.Tensor, w1: torch.Tensor, w2: torch.Tensor, d: int, bs: int) -> bool:
    """The image path of the grouped GEMM: contraction lengths and block offsets in whole
    32-element blocks (ext.grouped_gemm_fused falls back to the fp32 operands otherwise)."""
    return d % 32 == 0 and bs % 32 == 0 and a.is_contiguous()


def fused_usable(rank: int) -> bool:
    return rank % 4 == 0 and 0 < rank <= 32


class RoutedLoRALLaMAFFN(torch.autograd.Function):
    """y = LoRARoutedLLaMaFFN(x) for frozen base weights (reference formula:
    naive_gpt/layers/tuning/lora_ffn.py:196-222), with c the router coefficient of a row:

        g  = c (x G_g^T) + (x Lg) Rg_g^T,   sd = c (x S_g^T) + (x Ls) Rs_g^T     two fused GEMMs
        h  = act(g) * sd                                                        (torch)
        ys = c (h D_g) + (h Ld_g) Rd^T;   y = sum over the token's rows of ys   fused GEMM + combine

    Same building blocks as RoutedLoRAFFN (LoRA side products as the GEMMs' K extension, token
    gather and rowscale inside the kernel, rows_combine instead of index_add); the gating product
    and the activation derivative stay elementwise torch ops.  The coefficient gradient uses
    <A, h> = (<c A + E, h> - <E, h>) / c with <E, h> = <dy Rd, h Ld_g> (both [*, r]), and the same
    identity for the two up projections."""

    @staticmethod
    def forward(ctx, x, rw, rb, lg, rg, ls, rs, ld, rd, wg, ws, wd, nb: int, top_k: int, activation,
                origin_input=None, origin_module=None):
        d_ff, d = wg.shape
        x = x.contiguous()
        prob, bk, block = _route(x, rw, rb, nb, top_k)
        coeff = bk.coeff
        bs, rank, rows = d_ff // nb, lg.size(1), bk.token.numel()
        imgs = _images_usable(x, wg, wd, d, bs)
        # one pass over x: both adapters' down products and the image both up projections read
        table = torch.cat([lg, ls], dim=1)
        if rank == 16 and ext.lora_down_supported(x, table):
            ext.note_path('lora_down', 'kernel')
            ugs, x_image, _ = ext.lora_down(x, table, want_image=True, block_major=True) if imgs \
                else (ext.lora_down(x, table, block_major=True), None, None)
            ug, us = ugs[0], ugs[1]                                               # [T, r]
        else:
            ext.note_path('lora_down', 'library', fallback=True,
                          why=lambda: 'gated FFN: x {} x [lg | ls] {}'.format(tuple(x.shape), tuple(table.shape)))
            ug, us = torch.matmul(x, lg), torch.matmul(x, ls)
            x_image = ext.split_bf16(x) if imgs else None

        def up(w, u, r_table):
            return ext.grouped_gemm_fused(
                x, w, bk.offsets, nb, n=bs, k=d, w_group_stride=bs * d, w_ldn=d, w_ldk=1,
                n_rows=rows, gather=bk.token, rowscale=coeff, a2=u, gather2=bk.token,
                b2=r_table, b2_group_stride=bs * rank,
                a_image=x_image, w_image=ext.weight_image(w) if imgs else None)

        g, sd = up(wg, ug, rg), up(ws, us, rs)
        silu = isinstance(activation, torch.nn.SiLU) and ext.swiglu_supported(g, sd)
        ext.note_path('swiglu', 'kernel' if silu else 'torch', fallback=not silu,
                      why=lambda: 'activation {} on {}'.format(type(activation).__name__, tuple(g.shape)))
        h = ext.swiglu_forward(g, sd) if silu else activation(g) * sd
        zw, h_img, _ = _down(h, _block_cat(ld, nb), imgs and not H_A32, False)
        z = _own_block(zw, bk.block, nb).contiguous()
        ys = ext.grouped_gemm_fused(
            h, wd, bk.offsets, nb, n=d, k=bs, w_group_stride=bs, w_ldn=d_ff, w_ldk=1,
            n_rows=rows, rowscale=coeff, a2=z, b2=rd, b2_group_stride=0,
            a_image=h_img, w_image=ext.weight_image(wd) if imgs else None)
        del x_image, h_img
        y = ext.rows_combine(ys, bk.pos)
        ctx.bk, ctx.activation = bk, activation
        ctx.origin_module = origin_module
        ctx.save_for_backward(x if origin_module is None else origin_input, coeff, ug, us, g, sd,
                              h, z, lg, rg, ls, rs, ld, rd, wg, ws, wd, prob, block, rw)
        return y

    @staticmethod
    def backward(ctx, dy):
        (x, coeff, ug, us, g, sd, h, z, lg, rg, ls, rs, ld, rd, wg, ws, wd, prob, block,
         rw) = ctx.saved_tensors
        if ctx.origin_module is not None:
            x = recompute.output(ctx.origin_module, x).view(-1, wg.size(1))
        bk, activation = ctx.bk, ctx.activation
        nb = bk.n_blocks
        d_ff, d = wg.shape
        bs, rank, rows = d_ff // nb, lg.size(1), bk.token.numel()
        dy = dy.contiguous()
        imgs = _images_usable(dy, wg, wd, d, bs)
        dzt, dy_img, _ = _down(dy, rd, imgs, False)                           # [T, r]
        dz_rows = dzt.index_select(0, bk.token_long)
        dh = ext.grouped_gemm_fused(
            dy, wd, bk.offsets, nb, n=bs, k=d, w_group_stride=bs, w_ldn=1, w_ldk=d_ff,
            n_rows=rows, gather=bk.token, rowscale=coeff, a2=dzt, gather2=bk.token, b2=ld,
            b2_group_stride=bs * rank,
            a_image=dy_img, w_image=ext.weight_image(wd) if imgs else None)
        del dy_img
        silu = isinstance(activation, torch.nn.SiLU) and ext.swiglu_supported(g, sd)
        if silu:
            # through h = silu(g) * sd, and the row dots of the coefficient gradient, in one pass
            dg, dsd, dot_h, dot_g, dot_s = ext.swiglu_backward(dh, g, sd)
            del dh
        else:
            dot_h = (dh * h).sum(dim=-1)
            dot_g = dot_s = None
            with torch.enable_grad():
                g_ = g.detach().requires_grad_(True)
                ag = activation(g_)
                dg, = torch.autograd.grad(ag, g_, dh * sd)
            dsd = dh * ag.detach()
            del dh, ag, g_                   # [P, bs] each: the step's peak memory is in here
        grad_coeff = (dot_h - (dz_rows * z).sum(dim=-1)) / _floor(coeff)

        def down(dpre, pre, w, u, l_table, r_table, dot_pre=None, into=None):
            """-> (dx rows [P, d], d coefficient [P], grad of the two LoRA tables)"""
            du, _ = _down_blocks(dpre, r_table, bk, False)
            u_rows = u.index_select(0, bk.token_long)
            if dot_pre is None:
                dot_pre = (dpre * pre).sum(dim=-1)
            dc = (dot_pre - (du * u_rows).sum(dim=-1)) / _floor(coeff)
            # (`into`: the other projection's dx rows -- this product is ADDED to them by the GEMM)
            dxs = ext.grouped_gemm_fused(
                dpre, w, bk.offsets, nb, n=d, k=bs, w_group_stride=bs * d, w_ldn=1, w_ldk=d,
                n_rows=rows, rowscale=coeff, a2=du, b2=l_table, b2_group_stride=0,
                out=into, accumulate=into is not None)
            grad_l = _tn(x, ext.rows_combine(du, bk.pos))
            grad_r = _tn_blocks(dpre, u, bk)
            return dxs, dc, grad_l, grad_r

        dxs_g, dc_g, grad_lg, grad_rg = down(dg, g, wg, ug, lg, rg, dot_g)
        del dg
        dxs_g, dc_s, grad_ls, grad_rs = down(dsd, sd, ws, us, ls, rs, dot_s, into=dxs_g)
        del dsd
        grad_x = ext.rows_combine(dxs_g, bk.pos)
        del dxs_g
        grad_coeff = grad_coeff + dc_g + dc_s
        grad_rw, grad_rb = _route_backward(ctx, grad_coeff, prob, bk, block, x, rw, grad_x)
        grad_ld = _tn_blocks(h, dzt, bk)
        grad_rd = _tn(dy, ext.rows_combine(z, bk.pos))
        return (grad_x, grad_rw, grad_rb, grad_lg, grad_rg, grad_ls, grad_rs, grad_ld, grad_rd,
                None, None, None, None, None, None, None, None)


def routed_lora_llama_ffn(x, rw, rb, lg, rg, ls, rs, ld, rd, wg, ws, wd, n_blocks: int, top_k: int,
                          activation, origin=None):
    if origin is None:
        return RoutedLoRALLaMAFFN.apply(x, rw, rb, lg, rg, ls, rs, ld, rd, wg, ws, wd, n_blocks, top_k,
                                        activation)
    return RoutedLoRALLaMAFFN.apply(x, rw, rb, lg, rg, ls, rs, ld, rd, wg, ws, wd, n_blocks, top_k,
                                    activation, origin.input, origin.module)
