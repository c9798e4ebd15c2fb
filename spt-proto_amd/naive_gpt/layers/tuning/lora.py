"""LoRA adapters (reference: ``naive_gpt/layers/tuning/lora.py``).

The frozen base layer keeps its ``weight`` / ``bias``; the trainable rank-r side
path lives in ``lora.left.weight [in, r]`` and ``lora.right.weight [out, r]`` (two
``nn.Embedding`` tables, so that the embedding variant can index ``left`` by token
id).  ``right`` starts at zero, so a fresh adapter is the identity.  No 1/r scaling
is applied (``scaling`` is stored but unused, as in the reference).
"""
import weakref

import torch
from torch import nn

from . import recompute


class LoRABase(nn.Module):
    def __init__(self, d_lora: int, in_features: int, out_features: int,
                 device: any = None, dtype: any = None):
        super().__init__()
        self.left = nn.Embedding(in_features, embedding_dim=d_lora,
                                 device=device, dtype=dtype)
        self.right = nn.Embedding(out_features, embedding_dim=d_lora,
                                  device=device, dtype=dtype)
        self.scaling = 1.0 / d_lora
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.zeros_(self.right.weight)


def _freeze(module: nn.Module):
    for param in module.parameters():
        param.requires_grad = False


def _load_base(model: nn.Module, source: nn.Module):
    # only the two LoRA tables may be absent from the source
    report = model.load_state_dict(source.state_dict(), strict=False)
    if len(report.missing_keys) != 2:
        raise RuntimeError
    return model


def tall_tn(a: torch.Tensor, b: torch.Tensor, chunk: int = 256) -> torch.Tensor:
    """a^T @ b for tall a [R, m], b [R, n] with a small m or n -- the shape of every LoRA
    table gradient (R = tokens, one side = rank).  As a single GEMM its [m, n] output is
    a handful of workgroups on a 256-CU part (34 us for 1024 x 16 x 8192, 116 us for
    1024 x 64 x 16384 measured); split over R into a batched product plus a sum it fills
    the chip.  Falls back to one matmul for short or ragged R."""
    rows = a.size(0)
    if a.is_cuda and rows >= 512:
        # one call of libspt_hip (lora_side.hip: spt_tall_tn) -- the same HBM-bound pass as the
        # library's batched product + sum, at a quarter of their host time (the step makes ~260
        # of these products and is within 10 % of being host-bound)
        from naive_gpt import ext
        if a.size(1) >= b.size(1) and ext.tall_tn_supported(a, b):
            ext.note_path('tall_tn', 'kernel')
            return ext.tall_tn(a, b)[0]
        if a.size(1) < b.size(1) and ext.tall_tn_supported(b, a):
            ext.note_path('tall_tn', 'kernel')
            return ext.tall_tn(b, a, transposed=True)[0]
        ext.note_path('tall_tn', 'library', fallback=True,
                      why=lambda: 'a {} b {}: spt_tall_tn takes a narrow side of 4, 16 or 48 columns'.format(
                          tuple(a.shape), tuple(b.shape)))
    if not a.is_cuda or rows % chunk != 0 or rows < 8 * chunk:
        return torch.matmul(a.t(), b)
    parts = rows // chunk
    out = torch.bmm(a.reshape(parts, chunk, a.size(1)).transpose(1, 2),
                    b.reshape(parts, chunk, b.size(1)))
    return out.sum(dim=0)


def tall_tn_many(pairs):
    """[tall_tn(a, b) for a, b in pairs] (None entries pass through) -- as ONE pair of launches
    (spt_tall_tn_batch) when the pairs have one shape and `a` is the wide side: a step makes these
    products in twos and threes (the two tables of a LoRA linear, the `right` tables of q / k / v),
    each alone half launch overhead."""
    live = [i for i, p in enumerate(pairs) if p is not None]
    out = [None] * len(pairs)
    if len(live) >= 2 and pairs[live[0]][0].is_cuda and pairs[live[0]][0].size(0) >= 512:
        from naive_gpt import ext
        wides, narrows = [pairs[i][0] for i in live], [pairs[i][1] for i in live]
        if len(live) <= 4 and ext.tall_tn_batchable(wides, narrows):
            ext.note_path('tall_tn', 'kernel_batch')
            for i, t in zip(live, ext.tall_tn_batch(wides, narrows)):
                out[i] = t[0]
            return out
    for i in live:
        out[i] = tall_tn(*pairs[i])
    return out


_OFFSETS = {}


def _one_group(rows: int, device) -> torch.Tensor:
    """offsets [0, rows] of a single-bucket grouped GEMM, cached per (rows, device)."""
    key = (rows, str(device))
    off = _OFFSETS.get(key)
    if off is None:
        with torch.inference_mode(False):      # never cache an inference tensor
            off = torch.tensor([0, rows], dtype=torch.int32, device=device)
        _OFFSETS[key] = off
    return off


def _mfma_linear_usable(x2: torch.Tensor, weight: torch.Tensor, rank: int, forward: bool = True) -> bool:
    """Shapes for which the frozen product goes through the library's split-bf16 matrix-core
    GEMM (grouped_gemm.hip with one bucket; ~320 TFLOP/s against ~125 for the fp32 library
    GEMM): fp32, contiguous, the contraction length a multiple of 4 (rows of A are read 16
    bytes at a time), rank inside one k-tile, enough rows to fill the GPU.  The forward
    contracts over in_features, the backward (dX = dY W) over out_features -- a vocabulary of
    30522 takes the forward only."""
    n, k = weight.shape
    return (x2.is_cuda and x2.dtype == torch.float32 and weight.dtype == torch.float32
            and x2.is_contiguous() and weight.is_contiguous() and x2.size(0) >= 2048
            and k % 4 == 0 and (forward or n % 4 == 0) and 0 < rank <= 32)


def _mfma_gemm(a, weight, n, k, ldn, ldk, bias=None, a2=None, b2=None, images=False, owner=None):
    """out[p, :n] = a[p, :k] . W(n, k)^T (+ bias) (+ a2 . b2^T), W(n, k) = weight[n * ldn + k * ldk].

    `images`: run from the pre-split bf16 images of both operands (LDS-DMA k-loop, ~20 % less
    GEMM time).  The weight's image is kept with the frozen parameter; the activation's costs
    one pass over it (about what the GEMM saves), so it pays where several GEMMs read the same
    input -- the q / k / v projections -- which the small image cache of `ext` detects."""
    from naive_gpt import ext
    rows = a.size(0)
    a_image = w_image = None
    if images and k % 32 == 0 and a.is_contiguous():
        # the activation's image only when one exists already (another consumer of the same input
        # made it: lora_down's by-product, q / k / v); a lone consumer reads the fp32 rows
        # themselves -- the kernel's A32 form, ~10 % slower than from an image, whose making costs
        # a pass of its own (include/spt_hip.h: spt_grouped_gemm_image_path)
        a_image, w_image = ext.cached_image(a, owner), ext.weight_image(weight)
    return ext.grouped_gemm_fused(a, weight, _one_group(rows, a.device), 1, n, k, 0, ldn, ldk,
                                  rows, bias=bias, a2=a2, b2=b2, a_image=a_image, w_image=w_image)


# ---- the down products u = x L of adapters that share an input -------------------------------
# The q / k / v projections read the same tensor: their three tables go side by side through ONE
# pass over it (ext.lora_down: u for each adapter, plus the split image the three GEMMs read).
# `prepare_shared` is called by the attention module in front of the three linear layers; each
# layer then finds its u here.  One entry, alive as long as the input tensor is.
_SHARED = {}


def prepare_shared(x: torch.Tensor, linears) -> None:
    from naive_gpt import ext
    _SHARED.clear()
    if not (x.is_cuda and x.dim() >= 2 and not x.is_inference()):
        return
    if not all(isinstance(m, LoRALinear) and m._frozen() for m in linears):
        return
    lefts = [m.lora.left.weight for m in linears]
    x2 = x.reshape(-1, x.size(-1))
    if any(l.size(1) != 16 for l in lefts) or len(lefts) > 4:
        return
    if not all(_mfma_linear_usable(x2, m.weight, 16) for m in linears):
        return
    with torch.no_grad():
        table = torch.cat(lefts, dim=1)
        if not ext.lora_down_supported(x2, table):
            return
        u, image, _ = ext.lora_down(x2, table, want_image=True, block_major=True)
    ext.put_image(x2, x, image)
    _SHARED[id(x)] = (weakref.ref(x, lambda _, k=id(x): _SHARED.pop(k, None)), x._version,
                      {id(l): u[i] for i, l in enumerate(lefts)})


def _shared_u(x: torch.Tensor, left: torch.Tensor):
    hit = _SHARED.get(id(x))
    if hit is None or hit[0]() is not x or hit[1] != x._version:
        return None
    return hit[2].get(id(left))


def _down(x: torch.Tensor, x2: torch.Tensor, left: torch.Tensor, images: bool):
    """u = x2 @ left: found (prepare_shared), or one pass that also leaves x2's split image when
    the GEMM behind it wants one and none is cached, or the library product."""
    from naive_gpt import ext
    u = _shared_u(x, left)
    if u is not None:
        return u
    if ext.lora_down_supported(x2, left):
        # (no image by-product: the GEMM behind a lone layer reads the fp32 rows, _mfma_gemm)
        ext.note_path('lora_down', 'kernel')
        return ext.lora_down(x2, left)
    ext.note_path('lora_down', 'library', fallback=x2.is_cuda and x2.size(0) >= 2048,
                  why=lambda: 'x {} x left {}'.format(tuple(x2.shape), tuple(left.shape)))
    return torch.matmul(x2, left)


def _down_grad(dy2: torch.Tensor, right: torch.Tensor, out: torch.Tensor = None) -> torch.Tensor:
    """dU = dY R [T, r] of a LoRA layer's backward: the same tall-times-narrow pass as the forward's
    x L (spt_lora_down; a library GEMM costs the host four times as much per call), optionally
    into a column slice `out` of a wider matrix."""
    from naive_gpt import ext
    if ext.lora_down_supported(dy2, right):
        ext.note_path('lora_down', 'kernel')
        return ext.lora_down(dy2, right, out=out)
    ext.note_path('lora_down', 'library', fallback=dy2.is_cuda,
                  why=lambda: 'dy {} x right {}'.format(tuple(dy2.shape), tuple(right.shape)))
    du = torch.matmul(dy2, right)
    if out is None:
        return du
    out.copy_(du)
    return out


class _FrozenLoRALinear(torch.autograd.Function):
    """y = x W^T + b + (x L) R^T with W, b frozen (the reference's forward,
    lora.py:70-80, differentiated by hand): the side product is accumulated into the base
    product by the GEMM epilogue (addmm, beta = 1) instead of a separate elementwise pass,
    no weight gradient is formed, and the two table gradients use tall_tn."""

    @staticmethod
    def forward(ctx, x, weight, bias, left, right):
        from naive_gpt import ext
        x2 = x.reshape(-1, x.size(-1))
        n, k = weight.shape
        ctx.mfma = _mfma_linear_usable(x2, weight, left.size(1))
        ext.note_path('lora_linear', 'mfma' if ctx.mfma else 'library', fallback=not ctx.mfma,
                      why=lambda: 'x {} {} contiguous={} W {}: the split-bf16 GEMM takes fp32, >= 2048 rows, '
                          'k % 4 == 0, rank <= 32'.format(tuple(x2.shape), x2.dtype, x2.is_contiguous(),
                                                          tuple(weight.shape)))
        u = _down(x, x2, left, ctx.mfma and x2.is_contiguous())      # [T, r]
        if ctx.mfma:
            # one launch: base product, bias and the side product (the K extension) together
            y = _mfma_gemm(x2, weight, n, k, k, 1, bias=None if bias is None else bias.view(1, n),
                           a2=u.contiguous(), b2=right.contiguous(), images=True, owner=x)
        else:
            y = nn.functional.linear(x2, weight, bias)
            y.addmm_(u, right.t())
        # a tagged norm output is not kept: its origin is (layers/tuning/recompute.py)
        origin = recompute.origin_of(x)
        ctx.origin_module = origin.module if origin is not None else None
        ctx.save_for_backward(origin.input if origin is not None else x2, u, weight, left, right)
        ctx.x_shape = x.shape
        return y.view(*x.shape[:-1], weight.size(0))

    @staticmethod
    def backward(ctx, dy):
        x2, u, weight, left, right = ctx.saved_tensors
        if ctx.origin_module is not None:
            x2 = recompute.output(ctx.origin_module, x2).reshape(-1, weight.size(1))
        dy2 = dy.reshape(-1, dy.size(-1))
        du = _down_grad(dy2, right)                                  # [T, r]
        grad_x = None
        if ctx.needs_input_grad[0]:
            n, k = weight.shape
            from naive_gpt import ext
            mfma = ctx.mfma and dy2.is_contiguous() and _mfma_linear_usable(dy2, weight, left.size(1), False)
            ext.note_path('lora_linear_backward', 'mfma' if mfma else 'library', fallback=not mfma,
                          why=lambda: 'dy {} contiguous={} W {}'.format(tuple(dy2.shape), dy2.is_contiguous(),
                                                                tuple(weight.shape)))
            if mfma:
                # dX = dY W + dU L^T: the weight read with n contiguous... W'(k', n) = W[n, k']
                grad_x = _mfma_gemm(dy2, weight, k, n, 1, k, a2=du.contiguous(),
                                    b2=left.contiguous())
            else:
                grad_x = torch.matmul(dy2, weight)
                grad_x.addmm_(du, left.t())
            grad_x = grad_x.view(ctx.x_shape)
        grad_left, grad_right = tall_tn_many([(x2, du) if ctx.needs_input_grad[3] else None,
                                              (dy2, u) if ctx.needs_input_grad[4] else None])
        return grad_x, None, None, grad_left, grad_right


# ---- the three frozen weights of a self-attention's projections in one allocation ---------------
# One grouped launch for q, k and v needs W_q, W_k, W_v at a constant stride: at first use the
# three parameters are re-homed into one [3, n, k] buffer (each `.data` a view of it: same values,
# shapes and state_dict entries; the weights are frozen, so nothing else ever writes them).
_HOMES = {}
_QKV_INDEX = {}


def _home3(tensors):
    key = id(tensors[0])
    ptrs = tuple(t.data_ptr() for t in tensors)
    hit = _HOMES.get(key)
    if hit is not None and hit[1] == ptrs and all(r() is t for r, t in zip(hit[0], tensors)):
        return hit[2]
    with torch.inference_mode(False), torch.no_grad():      # (never an inference tensor: it outlives the call)
        home = torch.empty([3, *tensors[0].shape], dtype=tensors[0].dtype, device=tensors[0].device)
        for g, t in enumerate(tensors):
            home[g].copy_(t)
            t.data = home[g]
    refs = [weakref.ref(tensors[0], lambda _, k=key: _HOMES.pop(k, None))]
    refs += [weakref.ref(t) for t in tensors[1:]]
    _HOMES[key] = (refs, tuple(t.data_ptr() for t in tensors), home)
    return home


def _equal_spacing(mats) -> int:
    """floats from each matrix to the next when the three [rows, n] fp32 matrices are contiguous,
    disjoint and equally spaced in memory (views of one buffer), else 0"""
    a, b, c = mats
    if not all(m.is_contiguous() and m.dtype == torch.float32 and m.shape == a.shape for m in mats):
        return 0
    step = b.data_ptr() - a.data_ptr()
    if step < a.numel() * 4 or c.data_ptr() - b.data_ptr() != step or step % 16 != 0:
        return 0
    if not (a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr() == c.untyped_storage().data_ptr()):
        return 0
    return step // 4


def _qkv_index(rows: int, device):
    """(gather [3 rows] = 0 .. rows - 1 three times, offsets [0, rows, 2 rows, 3 rows]) int32."""
    key = (rows, str(device))
    hit = _QKV_INDEX.get(key)
    if hit is None:
        with torch.inference_mode(False):
            gather = torch.arange(rows, dtype=torch.int32, device=device).repeat(3)
            offsets = torch.arange(4, dtype=torch.int32, device=device) * rows
        hit = _QKV_INDEX[key] = (gather, offsets)
    return hit


class _FrozenLoRAQKV(torch.autograd.Function):
    """The three projections of a self-attention input as ONE function: q, k, v = x W_g^T + b_g +
    (x L_g) R_g^T (three `_FrozenLoRALinear`s, reference lora.py:70-80 called three times from
    multihead.py:13-15).  What sharing buys: one pass over x for the three down products and the
    operand image (forward); in the backward dX = sum_g dY_g W_g accumulates in one buffer (the
    GEMM's `accumulate`: no elementwise sums) and the three `left` gradients are one product
    x^T [dU_q | dU_k | dU_v]."""

    @staticmethod
    def forward(ctx, x, *params):
        from naive_gpt import ext
        weights, biases = params[0:3], params[3:6]
        lefts, rights = params[6:9], params[9:12]
        v_apart = bool(params[12])
        x2 = x.reshape(-1, x.size(-1))
        n, k = weights[0].shape
        rows = x2.size(0)
        # (adapter tables that lie equally spaced in one allocation -- a SparseTuner's flat parameter
        # buffer -- are read where they are: no concatenated / stacked copies per step)
        if ext.spacing_of(lefts):
            ext.note_path('qkv_tables', 'in_place')
            u3, image = ext.lora_down_tables(x2, lefts, want_image=True)
        else:
            ext.note_path('qkv_tables', 'copied')
            u3, image, _ = ext.lora_down(x2, torch.cat(lefts, dim=1), want_image=True, block_major=True)
        # ONE grouped launch: group g = the rows of x once more (gathered: the same image rows)
        # against W_g, into rows g * rows .. of the output -- 1,536 tiles instead of three
        # launches of 512, which leave a third of the workgroup slots of every CU empty
        # (55 us each where the routed FFN's 2,048-tile launch does the same work in 37)
        w3 = _home3(weights)
        b3 = None if biases[0] is None else _home3(biases)
        gather, offsets = _qkv_index(rows, x2.device)
        # (`v_apart`: q and k in one buffer, v in its own -- a rotary attention replaces q and k by
        # rotated copies at once, and a shared buffer would keep their memory alive with v)
        joint = 2 if v_apart else 3
        r = rights[0].size(1)
        w_img = ext.weight_image(w3.view(3 * n, k))
        r_step = ext.spacing_of(rights[:joint])
        out = ext.grouped_gemm_fused(
            x2, w3, offsets[:joint + 1], joint, n, k, n * k, k, 1, joint * rows, gather=gather[:joint * rows],
            bias=None if b3 is None else b3[:joint],
            a2=u3[:joint].view(joint * rows, r),
            b2=rights[0].detach() if r_step else torch.stack([t.detach() for t in rights[:joint]]),
            b2_group_stride=r_step if r_step else n * r, a_image=image, w_image=w_img)
        out = out.view(joint, *x.shape[:-1], n)
        outs = [out[g] for g in range(joint)]
        if v_apart:
            v_img = ext.SplitImage(w_img.buffer[2 * n * k * 4:], n, k)       # the third weight's rows
            y = ext.grouped_gemm_fused(
                x2, weights[2], _one_group(rows, x2.device), 1, n, k, 0, k, 1, rows,
                bias=None if biases[2] is None else biases[2].view(1, n),
                a2=u3[2], b2=rights[2].detach().contiguous(), a_image=image, w_image=v_img)
            outs.append(y.view(*x.shape[:-1], n))
        origin = recompute.origin_of(x)
        ctx.origin_module = origin.module if origin is not None else None
        ctx.save_for_backward(origin.input if origin is not None else x2, u3, *weights, *lefts, *rights)
        ctx.x_shape = x.shape
        return tuple(outs)

    @staticmethod
    def backward(ctx, *dys):
        from naive_gpt import ext
        saved = ctx.saved_tensors
        x2, u3 = saved[0], saved[1]
        weights, lefts, rights = saved[2:5], saved[5:8], saved[8:11]
        n, k = weights[0].shape
        if ctx.origin_module is not None:
            x2 = recompute.output(ctx.origin_module, x2).reshape(-1, k)
        rows = x2.size(0)
        one = _one_group(rows, x2.device)
        dy2 = [None if d is None else d.reshape(-1, n).contiguous() for d in dys]
        live = [g for g in range(3) if dy2[g] is not None]
        r = lefts[0].size(1)
        # the live dU_g = dY_g R_g side by side in one matrix: the operand of the joint `left` gradient
        du_cat = torch.empty([rows, r * len(live)], dtype=x2.dtype, device=x2.device)
        dus = [None] * 3
        if len(live) == 3 and r == 16 and _equal_spacing(dy2) == rows * n and ext.spacing_of(rights) \
                and ext.lora_down_supported(dy2[0], rights[0]):
            # dU_q | dU_k | dU_v from ONE launch over the stacked gradients [dQ; dK; dV] (each alone is a
            # 512-workgroup launch for 6 us of memory time)
            ext.note_path('lora_down', 'kernel')
            ext.lora_down_stacked(dy2, rights, _qkv_index(rows, x2.device)[1], du_cat)
            dus = [du_cat[:, i * r:(i + 1) * r] for i in range(3)]
        else:
            for i, g in enumerate(live):
                dus[g] = _down_grad(dy2[g], rights[g], out=du_cat[:, i * r:(i + 1) * r])
        grad_x = None
        if ctx.needs_input_grad[0]:
            step = _equal_spacing(dy2) if len(live) == 3 and r == 16 else 0
            if step and _equal_spacing(weights) == n * k:
                # dX = [dQ | dK | dV] [Wq; Wk; Wv] + [dU_q | dU_k | dU_v] [L_q | L_k | L_v]^T: ONE
                # contraction over k = 3 n -- the three gradients lie `step` floats apart (the
                # attention's backward writes them into one buffer) and the three weights are one
                # [3 n, k] matrix (_home3): a third of the launches, each three times as long, and
                # no read-add-write of dX between them
                ext.note_path('qkv_dx', 'joint')
                l_step = ext.spacing_of(lefts)
                grad_x = ext.grouped_gemm_fused(
                    dy2[0], weights[0].as_strided((3 * n, k), (k, 1)), one, 1, k, 3 * n, 0, 1, k, rows, a2=du_cat,
                    b2=lefts[0].detach() if l_step else torch.cat([t.detach() for t in lefts], dim=1),
                    b2_segment_stride=l_step, a_segments=(n, step))
            else:
                ext.note_path('qkv_dx', 'three_launches', fallback=len(live) == 3,
                              why=lambda: 'the gradients of q, k, v are not three equally spaced matrices '
                                          '(another op between the projection and the attention)')
                for g in range(3):
                    if dy2[g] is None:
                        continue
                    # dX (+)= dY_g W_g + dU_g L_g^T
                    grad_x = ext.grouped_gemm_fused(
                        dy2[g], weights[g], one, 1, k, n, 0, 1, k, rows, a2=dus[g],
                        b2=lefts[g].contiguous(), out=grad_x, accumulate=grad_x is not None)
            if grad_x is not None:
                grad_x = grad_x.view(ctx.x_shape)
        grad_lefts, grad_rights = [None] * 3, [None] * 3
        if live:
            if r == 16 and ext.tall_tn_supported(x2, du_cat) and x2.size(0) >= 512:
                gl = ext.tall_tn(x2, du_cat, split16=True)[0]                  # [len(live), k, 16]
            else:
                gl = tall_tn(x2, du_cat).view(x2.size(1), len(live), r).permute(1, 0, 2)
            g_rights = tall_tn_many([(dy2[g], u3[g]) for g in live])
            for i, g in enumerate(live):
                grad_lefts[g] = gl[i]
                grad_rights[g] = g_rights[i]
        return (grad_x, None, None, None, None, None, None, *grad_lefts, *grad_rights, None)


def qkv_usable(x: torch.Tensor, linears) -> bool:
    from naive_gpt import ext
    if not (x.is_cuda and x.dim() >= 2 and x.dtype == torch.float32 and not x.is_inference()):
        return False
    if not all(isinstance(m, LoRALinear) and m._frozen() for m in linears):
        return False
    w = linears[0].weight
    if any(m.weight.shape != w.shape or (m.bias is None) != (linears[0].bias is None)
           or m.lora.left.weight.size(1) != 16 for m in linears):
        return False
    x2 = x.reshape(-1, x.size(-1))
    n, k = w.shape
    return (x2.is_contiguous() and _mfma_linear_usable(x2, w, 16) and _mfma_linear_usable(x2, w, 16, False)
            and k % 32 == 0 and n % 32 == 0
            and all(m.lora.left.weight.requires_grad and m.lora.right.weight.requires_grad for m in linears))


def qkv(x: torch.Tensor, linears, v_apart: bool = False):
    """(linear_q(x), linear_k(x), linear_v(x)) as one autograd function (see _FrozenLoRAQKV).
    `v_apart`: v in a buffer of its own (for callers that drop q and k at once)."""
    args = ([m.weight for m in linears] + [m.bias for m in linears]
            + [m.lora.left.weight for m in linears] + [m.lora.right.weight for m in linears])
    return _FrozenLoRAQKV.apply(x, *args, v_apart)


class LoRALinear(nn.Linear):
    def __init__(self, d_lora: int, in_features: int, out_features: int,
                 bias: bool = True, *args, **kwargs):
        super().__init__(in_features=in_features, out_features=out_features,
                         bias=bias, *args, **kwargs)
        _freeze(self)
        self.lora = LoRABase(d_lora=d_lora, in_features=in_features,
                             out_features=out_features)

    @staticmethod
    def from_pretrained(d_lora: int, source: nn.Linear):
        model = LoRALinear(d_lora=d_lora, in_features=source.in_features,
                           out_features=source.out_features,
                           bias=source.bias is not None)
        return _load_base(model, source)

    def _frozen(self) -> bool:
        return not self.weight.requires_grad and (self.bias is None or not self.bias.requires_grad)

    def forward(self, x: torch.Tensor):
        frozen = self._frozen()
        if frozen and x.is_cuda and x.dim() >= 2:
            return _FrozenLoRALinear.apply(x, self.weight, self.bias,
                                           self.lora.left.weight, self.lora.right.weight)
        y = nn.functional.linear(x, bias=self.bias, weight=self.weight)
        y += torch.matmul(torch.matmul(x, self.lora.left.weight),
                          self.lora.right.weight.T)
        return y


class _LookupRows(torch.autograd.Function):
    """table[ids] (the adapter's `left(x)`, lora.py:118-126) with a backward that a HIP graph can
    replay.  torch's `embedding_dense_backward` sorts the ids and, above 3072 of them, takes the
    number of DISTINCT ids to the host to size its next launches (thrust::unique_by_key_copy on
    ROCm): captured, that count is the capture batch's for ever -- a replay on other tokens reads
    its segment table past the end (round 3: `Memory access fault` / MEMORY_APERTURE_VIOLATION in
    rocprim's partition_kernel on the bench's captured step; round 2's captured step ran on the
    same stale count without faulting).  Here (`spt_embedding_rows_backward`) every launch has a shape
    fixed by T alone and the result does not depend on timing: stable sort by id, every run of equal
    ids summed in order in fp64 -- chunk by chunk, then chunk sums per run -- and written once.  A
    run is summed on its own: one non-finite gradient row stays in its id's row (round 3's torch
    composition took ONE running sum over all sorted rows and differences of it, which carried an inf
    into every later id: ADVICE round 3)."""

    @staticmethod
    def forward(ctx, ids, table):
        ctx.save_for_backward(ids)
        ctx.rows = table.size(0)
        return nn.functional.embedding(ids, table)

    @staticmethod
    def backward(ctx, grad):
        ids, = ctx.saved_tensors
        g = grad.reshape(ids.numel(), -1)
        if g.dtype == torch.float32 and g.size(1) % 4 == 0 and g.stride(1) == 1 and g.stride(0) % 4 == 0:
            from naive_gpt import ext
            return None, ext.embedding_rows_backward(g, ids.reshape(-1), ctx.rows)
        # (other widths / dtypes: torch's scatter-add; deterministic only without duplicates)
        out = torch.zeros([ctx.rows, g.size(1)], dtype=g.dtype, device=g.device)
        out.index_add_(0, ids.reshape(-1), g)
        return None, out


class LoRAEmbedding(nn.Embedding):
    def __init__(self, d_lora: int, num_embeddings: int, embedding_dim: int,
                 *args, **kwargs):
        super().__init__(num_embeddings=num_embeddings,
                         embedding_dim=embedding_dim, *args, **kwargs)
        _freeze(self)
        self.lora = LoRABase(d_lora=d_lora, in_features=num_embeddings,
                             out_features=embedding_dim)

    @staticmethod
    def from_pretrained(d_lora: int, source: nn.Embedding):
        model = LoRAEmbedding(d_lora=d_lora, num_embeddings=source.num_embeddings,
                              embedding_dim=source.embedding_dim)
        return _load_base(model, source)

    def forward(self, x: torch.Tensor):
        left = self.lora.left
        if x.is_cuda and left.weight.requires_grad and left.padding_idx is None \
                and left.max_norm is None and not left.scale_grad_by_freq and not left.sparse:
            u = _LookupRows.apply(x, left.weight)          # (graph-safe backward, see above)
        else:
            u = left(x)
        side = torch.matmul(u, self.lora.right.weight.T)
        return nn.functional.embedding(x, weight=self.weight) + side
