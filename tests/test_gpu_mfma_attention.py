"""spt_attention_mfma_* (mfma_attention.hip: dense score tiles on the bf16 matrix cores, fp32
operands split in two) against the oracle chain sddmm -> scale, clamp -> softmax -> spmm."""
import pytest
import torch

from test_gpu_fused_attention import CLAMP, causal_indices, oracle_chain

pytestmark = pytest.mark.gpu

SHAPES = [                             # (N, H, S, Z, d_head)
    (2, 16, 512, 64, 64), (1, 32, 256, 32, 64), (4, 8, 128, 16, 64),
    (2, 16, 80, 8, 64),                # S not a multiple of the 32-row tile
    (1, 4, 1024, 64, 64),              # four row blocks per slice
    (1, 2, 2048, 64, 64),              # the longest supported sequence
    (1, 4, 96, 64, 64),                # Z close to S: many repeated columns
    (1, 2, 2048, 256, 64),             # Z = S / 8 at the longest sequence: row 0 saturates a cell
    (1, 2, 1024, 128, 64),
    (2, 8, 512, 64, 128),              # d_head 128 (LLaMA heads): the spt::e128 kernels,
    (1, 4, 80, 8, 128),                # two 64-column backward launches each
    (1, 2, 1024, 128, 128),
    (1, 2, 2048, 256, 128),            # BASELINE configs[4] (LLaMA-7B heads at S = 2048): cell
]                                      # counts saturate in row 0, the e128 build at 2,080 tiles


@pytest.mark.parametrize('yt', [False, True])
@pytest.mark.parametrize('N,H,S,Z,E', SHAPES)
def test_mfma_forward_matches_oracle_chain(N, H, S, Z, E, yt):
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(N * S + Z)
    q, k, v = [torch.randn([N, S, H, E], generator=gen) for _ in range(3)]
    q = q * 3.0                                    # some scores beyond the clamp
    indices = causal_indices(N * H, S, Z, gen)     # random with repeats: multiplicities matter
    scale = E ** -0.5
    assert ext.attention_mfma_supported(S, E, S * Z)
    want_scores, want_attn, want_y = oracle_chain(indices, q, k, v, scale)
    y, row_sum = ext.attention_mfma_forward(indices.cuda(), q.cuda(), k.cuda(), v.cuda(),
                                            scale, CLAMP, y_transposed=yt)
    if yt:
        assert y.shape == (N * H, E, S)
        y = y.transpose(1, 2)
    rows = torch.arange(S).view(1, S, 1)
    live = (indices.view(N * H, S, Z) <= rows).float()
    want_sum = (want_scores.view(N * H, S, Z).exp() * live).sum(-1)
    # (Z = 256: row 0 is column 0 256 times -- one more than a byte-sized cell count holds; the
    # cell tiles flag such rows and the kernels add the 1 back: exact since ABI 39)
    assert torch.allclose(row_sum.cpu(), want_sum, rtol=1e-4)
    assert torch.allclose(y.cpu(), want_y, rtol=1e-3, atol=1e-4)
    # the split-bf16 products are far inside the bar: report-level check of the actual error
    assert (y.cpu() - want_y).abs().max() < 2e-4 * want_y.abs().max()


@pytest.mark.parametrize('gt', [False, True])
@pytest.mark.parametrize('N,H,S,Z,E', SHAPES)
def test_mfma_backward_matches_oracle_chain(N, H, S, Z, E, gt):
    """grad_q, grad_k, grad_v against the oracle operators chained as the reference's autograd
    does: dP = sddmm(dY, V); dS = clamp-mask(scale * softmax_backward(P, dP));
    dQ = spmm(dS, K); dK = spmm(dS^T, Q); dV = spmm(P^T, dY)."""
    import numpy as np
    from oracle import ext_stub
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(7 * N + S + Z)
    B = N * H
    q, k, v = [torch.randn([N, S, H, E], generator=gen) for _ in range(3)]
    q = q * 3.0
    gy = torch.randn([B, S, E], generator=gen)
    indices = causal_indices(B, S, Z, gen)
    scale = E ** -0.5
    scores, attn, want_y = oracle_chain(indices, q, k, v, scale)
    flat = lambda t: t.transpose(1, 2).contiguous().view(B, S, E)       # noqa: E731
    heads = lambda t: t.view(N, H, S, E).transpose(1, 2)                  # noqa: E731
    indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32)
    f, t = torch.scalar_tensor(False), torch.scalar_tensor(True)
    dp = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, gy, flat(v))
    ds = ext_stub.softmax_backward_cuda(indptr, indices, attn, dp)
    raw = torch.where(scores.abs() < CLAMP, ds * np.float32(scale), torch.zeros_like(ds))
    want_q = heads(ext_stub.spmm_forward_cuda(f, f, indptr, indices, raw, flat(k)))
    want_k = heads(ext_stub.spmm_forward_cuda(t, f, indptr, indices, raw, flat(q)))
    want_v = heads(ext_stub.spmm_forward_cuda(t, f, indptr, indices, attn, gy))

    tiles = ext.attention_mfma_prepare(indices.cuda(), S)
    y, row_sum = ext.attention_mfma_forward(tiles, q.cuda(), k.cuda(), v.cuda(), scale, CLAMP,
                                            y_transposed=gt)
    gy_in = gy.transpose(1, 2).contiguous() if gt else gy
    gq, gk, gv = ext.attention_mfma_backward(tiles, q.cuda(), k.cuda(), v.cuda(), y, gy_in.cuda(),
                                             row_sum, scale, CLAMP, transposed=gt)
    assert (scores.abs() == CLAMP).any() and (raw == 0).any()
    for got, want, name in ((gq, want_q, 'grad_q'), (gk, want_k, 'grad_k'), (gv, want_v, 'grad_v')):
        err = (got.cpu() - want).abs().max().item()
        assert torch.allclose(got.cpu(), want, rtol=1e-3, atol=2e-4 * want.abs().max().item()), \
            (name, err, want.abs().max().item())


def _scores_on_the_clamp(N, H, S, Z, E, seed, width):
    """q, k, indices with ONE entry of most rows moved onto the clamp: the row's q is rescaled so
    that its LARGEST live score becomes scale * q_i . k_j = +-CLAMP * (1 + delta), delta uniform in
    +-width (rows that would need a factor beyond 1/4 .. 12 stay as they are, so that the operands
    keep ordinary magnitudes).  With width a few 2^-17 -- the split-bf16 products' error -- a score
    formed from split operands falls on either side of the clamp whatever the fp32 score does."""
    gen = torch.Generator().manual_seed(seed)
    B = N * H
    q, k, v = [torch.randn([N, S, H, E], generator=gen) for _ in range(3)]
    indices = causal_indices(B, S, Z, gen)
    scale = E ** -0.5
    idx = indices.view(N, H, S, Z).long()
    rows = torch.arange(S)
    kk = k.transpose(1, 2).double()                                        # [N, H, S, E]
    qq = q.transpose(1, 2).double()
    kj = kk.unsqueeze(2).expand(N, H, S, S, E).gather(
        3, idx.unsqueeze(-1).expand(N, H, S, Z, E)) if S <= 128 else None
    if kj is not None:
        s = (qq.unsqueeze(3) * kj).sum(-1) * scale                         # [N, H, S, Z]
    else:                                                                  # (row blocks: memory)
        s = torch.empty([N, H, S, Z], dtype=torch.float64)
        for r0 in range(0, S, 64):
            blk = idx[:, :, r0:r0 + 64]
            kb = torch.stack([kk[n, h][blk[n, h]] for n in range(N) for h in range(H)]).view(
                N, H, blk.size(2), Z, E)
            s[:, :, r0:r0 + 64] = (qq[:, :, r0:r0 + 64].unsqueeze(3) * kb).sum(-1) * scale
    s = torch.where(idx <= rows.view(1, 1, S, 1), s, torch.zeros_like(s))
    top = s.abs().amax(-1)                                                 # [N, H, S]
    delta = (torch.rand([N, H, S], generator=gen, dtype=torch.float64) * 2 - 1) * width
    factor = CLAMP * (1 + delta) / top.clamp(min=1e-9)
    factor = torch.where((factor > 0.25) & (factor < 12.0), factor, torch.ones_like(factor))
    qq = (qq * factor.unsqueeze(-1)).float()
    return qq.transpose(1, 2).contiguous(), k, v, indices, scale


@pytest.mark.parametrize('gt', [False, True])
@pytest.mark.parametrize('N,H,S,Z,E', [(2, 16, 512, 64, 64), (1, 8, 80, 8, 64), (1, 2, 2048, 256, 64),
                                       (1, 4, 512, 64, 128), (1, 2, 96, 8, 128)])
def test_clamp_mask_is_that_of_the_exact_scores(N, H, S, Z, E, gt):
    """attention.py:125-127: the clamp passes gradient strictly inside (-10, 10) -- a discontinuous
    function of the score.  Thousands of scores are placed within 2^-15 (relative) of +-10, where the
    split-bf16 tile value and the fp32 score disagree about the side; the backward must follow the
    fp32 score (ABI 39: cells within the split's error bound of the clamp are recomputed as the oracle
    does).  The same inputs with the recomputation switched off differ visibly -- the test can see
    what it tests."""
    import numpy as np
    from oracle import ext_stub
    from naive_gpt import ext
    B = N * H
    q, k, v, indices, scale = _scores_on_the_clamp(N, H, S, Z, E, 11 * S + E, 2.0 ** -15)
    gen = torch.Generator().manual_seed(5)
    gy = torch.randn([B, S, E], generator=gen)
    scores, attn, _ = oracle_chain(indices, q, k, v, scale)
    flat = lambda t: t.transpose(1, 2).contiguous().view(B, S, E)       # noqa: E731
    heads = lambda t: t.view(N, H, S, E).transpose(1, 2)                  # noqa: E731
    indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32)
    f, t = torch.scalar_tensor(False), torch.scalar_tensor(True)
    raw_scores = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, flat(q), flat(k)) * np.float32(scale)
    near = ((raw_scores.abs() - CLAMP).abs() < CLAMP * 2.0 ** -14)
    assert near.sum() > 0.3 * B * S, 'the construction put too few scores on the clamp'
    assert 0.2 < (raw_scores[near].abs() < CLAMP).float().mean() < 0.8      # both sides populated
    dp = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, gy, flat(v))
    ds = ext_stub.softmax_backward_cuda(indptr, indices, attn, dp)
    raw = torch.where(raw_scores.abs() < CLAMP, ds * np.float32(scale), torch.zeros_like(ds))
    want_q = heads(ext_stub.spmm_forward_cuda(f, f, indptr, indices, raw, flat(k)))
    want_k = heads(ext_stub.spmm_forward_cuda(t, f, indptr, indices, raw, flat(q)))

    tiles = ext.attention_mfma_prepare(indices.cuda(), S)
    y, row_sum = ext.attention_mfma_forward(tiles, q.cuda(), k.cuda(), v.cuda(), scale, CLAMP,
                                            y_transposed=gt)
    gy_in = (gy.transpose(1, 2).contiguous() if gt else gy).cuda()

    def worst(exact):
        ext.EXACT_CLAMP = exact
        try:
            gq, gk, _ = ext.attention_mfma_backward(tiles, q.cuda(), k.cuda(), v.cuda(), y, gy_in,
                                                    row_sum, scale, CLAMP, transposed=gt)
        finally:
            ext.EXACT_CLAMP = True
        return [((got.cpu() - want).abs().max() / want.abs().max()).item()
                for got, want in ((gq, want_q), (gk, want_k))]

    exact, split = worst(True), worst(False)
    assert max(exact) < 2e-4, exact               # the bar of the other backward tests, no flips
    assert max(split) > 20 * max(exact), (split, exact)     # without it cells change sides


BF16_SHAPES = [                        # (N, H, S, Z, d_head): bf16 storage (BASELINE configs[1] "bf16")
    (2, 16, 512, 64, 64), (1, 4, 80, 8, 64), (1, 2, 2048, 256, 64),
    (2, 8, 512, 64, 128), (1, 2, 1024, 128, 128),
]
# Tolerances of the bf16-storage path.  The kernels compute on the widened inputs exactly as the
# fp32 path does (stored values are their own matrix-core operands; everything computed on the way
# is fp32 split in two), so BEFORE the store the result is within the fp32 path's 1e-3 of the
# oracle run on the bf16-rounded inputs; the store rounds once to nearest-even bf16, a relative
# 2^-9 per element.  Bar: |got - want| <= 1e-3 max|want| + 2^-8 |want|.
BF16_ULP = 2.0 ** -8


def _close_bf16(got, want, name):
    got = got.float().cpu()
    bound = 1e-3 * want.abs().max() + BF16_ULP * want.abs()
    err = (got - want).abs()
    assert (err <= bound).all(), (name, (err - bound).max().item(), want.abs().max().item())


@pytest.mark.parametrize('yt', [False, True])
@pytest.mark.parametrize('N,H,S,Z,E', BF16_SHAPES)
def test_mfma_forward_bf16_storage_matches_oracle_on_rounded_inputs(N, H, S, Z, E, yt):
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(N * S + Z + 1)
    q, k, v = [torch.randn([N, S, H, E], generator=gen) for _ in range(3)]
    q = q * 3.0                                    # some scores beyond the clamp
    q, k, v = [t.bfloat16() for t in (q, k, v)]
    indices = causal_indices(N * H, S, Z, gen)
    scale = E ** -0.5
    want_scores, _, want_y = oracle_chain(indices, q.float(), k.float(), v.float(), scale)
    y, row_sum = ext.attention_mfma_forward(indices.cuda(), q.cuda(), k.cuda(), v.cuda(),
                                            scale, CLAMP, y_transposed=yt)
    assert y.dtype == torch.bfloat16 and row_sum.dtype == torch.float32
    if yt:
        assert y.shape == (N * H, E, S)
        y = y.transpose(1, 2)
    rows = torch.arange(S).view(1, S, 1)
    live = (indices.view(N * H, S, Z) <= rows).float()
    want_sum = (want_scores.view(N * H, S, Z).exp() * live).sum(-1)
    # (a row that is column 0 Z = 256 times -- row 0; row 1 when both of its draws are 0 -- holds a
    # multiplicity a byte cannot: flagged in the cell tiles and added back, exact since ABI 39)
    if Z == 256:
        assert ((indices.view(N * H, S, Z) == 0).sum(-1) > 255).any()
    assert torch.allclose(row_sum.cpu(), want_sum, rtol=1e-4)
    _close_bf16(y, want_y, 'y')
    # and the fp32 kernels on the widened inputs give the same values before the store's rounding
    y32, rs32 = ext.attention_mfma_forward(indices.cuda(), q.float().cuda(), k.float().cuda(),
                                           v.float().cuda(), scale, CLAMP, y_transposed=yt)
    if yt:
        y32 = y32.transpose(1, 2)
    assert torch.allclose(row_sum, rs32, rtol=1e-5)
    assert ((y.float() - y32).abs() <= BF16_ULP * y32.abs() + 5e-5 * y32.abs().max()).all()


@pytest.mark.parametrize('gt', [False, True])
@pytest.mark.parametrize('N,H,S,Z,E', BF16_SHAPES)
def test_mfma_backward_bf16_storage_matches_oracle_on_rounded_inputs(N, H, S, Z, E, gt):
    import numpy as np
    from oracle import ext_stub
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(7 * N + S + Z + 1)
    B = N * H
    q, k, v = [torch.randn([N, S, H, E], generator=gen) for _ in range(3)]
    q = q * 3.0
    gy = torch.randn([B, S, E], generator=gen)
    q, k, v, gy = [t.bfloat16() for t in (q, k, v, gy)]
    indices = causal_indices(B, S, Z, gen)
    scale = E ** -0.5
    tiles = ext.attention_mfma_prepare(indices.cuda(), S)
    y, row_sum = ext.attention_mfma_forward(tiles, q.cuda(), k.cuda(), v.cuda(), scale, CLAMP,
                                            y_transposed=gt)
    gy_in = gy.transpose(1, 2).contiguous() if gt else gy
    gq, gk, gv = ext.attention_mfma_backward(tiles, q.cuda(), k.cuda(), v.cuda(), y, gy_in.cuda(),
                                             row_sum, scale, CLAMP, transposed=gt)
    assert gq.dtype == gk.dtype == gv.dtype == torch.bfloat16
    # oracle on the widened inputs.  delta = dY . Y uses the y the backward is GIVEN (the stored,
    # bf16-rounded one), as autograd would hand it over: the oracle's softmax backward forms the
    # same sum from P and dP, i.e. from the unrounded y -- the difference is one more 2^-9-relative
    # perturbation of delta, inside the bar below
    qf, kf, vf, gyf = q.float(), k.float(), v.float(), gy.float()
    scores, attn, _ = oracle_chain(indices, qf, kf, vf, scale)
    flat = lambda t: t.transpose(1, 2).contiguous().view(B, S, E)       # noqa: E731
    heads = lambda t: t.view(N, H, S, E).transpose(1, 2)                  # noqa: E731
    indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32)
    f, t = torch.scalar_tensor(False), torch.scalar_tensor(True)
    dp = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, gyf, flat(vf))
    ds = ext_stub.softmax_backward_cuda(indptr, indices, attn, dp)
    raw = torch.where(scores.abs() < CLAMP, ds * np.float32(scale), torch.zeros_like(ds))
    want_q = heads(ext_stub.spmm_forward_cuda(f, f, indptr, indices, raw, flat(kf)))
    want_k = heads(ext_stub.spmm_forward_cuda(t, f, indptr, indices, raw, flat(qf)))
    want_v = heads(ext_stub.spmm_forward_cuda(t, f, indptr, indices, attn, gyf))
    for got, want, name in ((gq, want_q, 'grad_q'), (gk, want_k, 'grad_k'), (gv, want_v, 'grad_v')):
        got = got.float().cpu()
        bound = 4e-3 * want.abs().max() + BF16_ULP * want.abs()
        err = (got - want).abs()
        assert (err <= bound).all(), (name, (err - bound).max().item(), want.abs().max().item())
    # sharper: against the fp32 kernels on the widened inputs AND the widened stored y
    y32 = y.float()
    g32 = ext.attention_mfma_backward(tiles, qf.cuda(), kf.cuda(), vf.cuda(), y32, gy_in.float().cuda(),
                                      row_sum, scale, CLAMP, transposed=gt)
    for got, ref, name in zip((gq, gk, gv), g32, ('grad_q', 'grad_k', 'grad_v')):
        assert ((got.float() - ref).abs() <= BF16_ULP * ref.abs() + 5e-5 * ref.abs().max()).all(), name


def test_bf16_storage_codes_equal_the_codes_of_the_widened_values():
    """PQ codes from bf16 storage are those of the fp32 operator on the widened values, bit for
    bit (and those of the oracle on them)."""
    import numpy as np
    from oracle import oracle as O
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(5)
    N, S, H, E, M, C = 2, 128, 4, 64, 8, 16
    z = torch.randn([N, S, H, E], generator=gen).bfloat16()
    table = torch.randn([M, C, E // M], generator=gen)
    got = ext.pq_encode_heads(z.cuda(), table.cuda())
    ref = ext.pq_encode_heads(z.float().cuda(), table.cuda())
    assert torch.equal(got, ref)
    zq = z.float().permute(0, 2, 1, 3).reshape(N * H * S, M, E // M).permute(1, 0, 2).contiguous().numpy()
    _, codes = O.cdist_forward(zq, table.numpy())
    assert np.array_equal(got.cpu().numpy().reshape(N * H * S, M), codes.T)


def test_mfma_forward_masks_columns_beyond_the_row():
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(4)
    N, H, S, Z = 2, 16, 128, 16
    q, k, v = [torch.randn([N, S, H, 64], generator=gen) for _ in range(3)]
    indices = torch.randint(0, S, [N * H, S * Z], generator=gen, dtype=torch.int32)
    indices.view(N * H, S, Z)[:, :, 0] = 0         # every row keeps one live entry
    _, _, want_y = oracle_chain(indices, q, k, v, 0.125)
    y, _ = ext.attention_mfma_forward(indices.cuda(), q.cuda(), k.cuda(), v.cuda(), 0.125, CLAMP)
    assert torch.allclose(y.cpu(), want_y, rtol=1e-3, atol=1e-4)


def test_mfma_unsupported_shapes_are_refused():
    from naive_gpt import ext
    assert not ext.attention_mfma_supported(128, 32, 128 * 16)        # d_head 32 (64 and 128 only)
    assert not ext.attention_mfma_supported(4096, 64, 4096 * 64)      # S > 2048
    assert not ext.attention_mfma_supported(4096, 64, 4096 * 512)     # Z > 256
    q = torch.randn([1, 4096, 2, 64], device='cuda')
    idx = torch.zeros([2, 4096 * 64], dtype=torch.int32, device='cuda')
    with pytest.raises(RuntimeError):
        ext.attention_mfma_forward(idx, q, q, q, 1.0, CLAMP)


def test_cell_tiles_hold_the_multiplicity_of_every_live_cell():
    """spt_attention_mfma_prepare: per (32-row, 32-key) tile the count of each (row, col) cell in
    the accumulator permutation (byte form when some count is >= 2, else 32-bit row masks), as
    documented in mfma_attention.hip."""
    import numpy as np
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(9)
    B, S, Z = 6, 160, 16
    indices = torch.randint(0, S, [B, S * Z], generator=gen, dtype=torch.int32)
    indices.view(B, S, Z)[:, :64, :3] = 0                    # repeated column 0, as lookup pads
    # some rows with distinct columns only, so that tiles in mask form exist as well
    for r in range(96, S):
        for b in range(B):
            indices.view(B, S, Z)[b, r] = torch.randperm(r + 1, generator=gen)[:Z].to(torch.int32)
    tiles = ext.attention_mfma_prepare(indices.cuda(), S)
    RT = S // 32
    raw = tiles.buffer.cpu().numpy()[256:]                   # behind the 256-byte header
    mask_bytes = (B * RT * 20 + 255) // 256 * 256            # masks [B, RT, 2] u64 | sat [B, RT] u32
    masks = raw[:B * RT * 16].view('uint64').reshape(B, RT, 2)
    assert not raw[B * RT * 16:B * RT * 20].any()            # no row holds a column 256 times
    ntile = RT * (RT + 1) // 2
    cells, cells_t = raw[mask_bytes:mask_bytes + 2 * B * ntile * 1024].reshape(2, B, ntile, 1024)
    idx = indices.view(B, S, Z).numpy()
    seen = set()
    for b in range(B):
        for rt in range(RT):
            want = np.zeros([RT, 32, 32], dtype=np.int64)       # [key tile][row][key]
            for r in range(32):
                for c in idx[b, rt * 32 + r]:
                    if c <= rt * 32 + r:
                        want[c // 32, r, c % 32] += 1
            for t in range(rt + 1):
                live = bool((int(masks[b, rt, 0]) >> t) & 1)
                multi = bool((int(masks[b, rt, 1]) >> t) & 1)
                assert live == bool(want[t].any())
                if not live:
                    continue
                assert multi == bool((want[t] >= 2).any())
                seen.add(multi)
                tile, tile_t = cells[b, rt * (rt + 1) // 2 + t], cells_t[b, rt * (rt + 1) // 2 + t]
                if multi:
                    tile, tile_t = tile.reshape(64, 16), tile_t.reshape(64, 16)
                    for lane in range(64):
                        c, h = lane % 32, lane // 32
                        for g in range(4):
                            for u in range(4):
                                assert tile[lane, 4 * g + u] == want[t, c, 8 * g + 4 * h + u]
                                assert tile_t[lane, 4 * g + u] == want[t, 8 * g + 4 * h + u, c]
                else:
                    rows, cols = tile[:128].view('uint32'), tile_t[:128].view('uint32')
                    for c in range(32):
                        assert int(rows[c]) == sum(int(want[t, c, k]) << k for k in range(32))
                        assert int(cols[c]) == sum(int(want[t, r, c]) << r for r in range(32))
    assert seen == {True, False}


@pytest.mark.parametrize('kind', ['vanilla', 'rotary'])
def test_layer_through_mfma_kernels_equals_separate_operators(kind, monkeypatch):
    from naive_gpt import ext, layers
    torch.manual_seed(0)
    cls = layers.SparseVanillaAttentionV2 if kind == 'vanilla' else layers.SparseRotaryAttentionV2
    attn = cls(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0).cuda()
    N, S, H = 2, 256, 16
    q, k, v, w = [torch.randn([N, S, H, 64], device='cuda') for _ in range(4)]

    def run(mfma):
        if not mfma:
            monkeypatch.setattr(ext, 'attention_mfma_supported', lambda *a: False)
            monkeypatch.setattr(ext, 'fused_attention_supported', lambda *a: False)
        qi, ki, vi = [t.clone().requires_grad_(True) for t in (q, k, v)]
        y = attn(qi, ki, vi, attn_mask=None)
        (y * w).sum().backward()
        monkeypatch.undo()
        return y.detach(), qi.grad, ki.grad, vi.grad

    fast, separate = run(True), run(False)
    for a, b, name in zip(fast, separate, ['y', 'grad_q', 'grad_k', 'grad_v']):
        assert torch.allclose(a, b, rtol=1e-3, atol=2e-4 * b.abs().max().item()), name


@pytest.mark.parametrize('kind', ['vanilla', 'rotary'])
def test_layer_with_bf16_storage_equals_the_fp32_layer_on_the_widened_values(kind):
    """BASELINE configs[1] "bf16": q, k, v stored bf16.  PQ codes and the CSR structure are those
    of the fp32 layer on the widened values bit for bit; y and the gradients agree to the
    rounding of the stored results (the rotary variant rotates q, k in bf16 first: compared
    after that step, on what the sparse core is given)."""
    from naive_gpt import layers
    torch.manual_seed(0)
    cls = layers.SparseVanillaAttentionV2 if kind == 'vanilla' else layers.SparseRotaryAttentionV2
    attn = cls(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0).cuda()
    N, S, H = 2, 256, 16
    q, k, v, w = [torch.randn([N, S, H, 64], device='cuda').bfloat16() for _ in range(4)]
    got_attn = attn._sparse_attn(q, k)
    want_attn = attn._sparse_attn(q.float(), k.float())
    assert got_attn[0] == want_attn[0] == 'mfma'
    assert torch.equal(got_attn[2], want_attn[2])                 # CSR column indices, bit-exact
    if kind == 'rotary':
        return

    def run(cast):
        qi, ki, vi = [cast(t).clone().requires_grad_(True) for t in (q, k, v)]
        y = attn(qi, ki, vi, attn_mask=None)
        (y * cast(w)).sum().backward()
        return y.detach(), qi.grad, ki.grad, vi.grad

    low, full = run(lambda t: t), run(lambda t: t.float())
    assert low[0].dtype == torch.bfloat16 and low[1].dtype == torch.bfloat16
    for a, b, name in zip(low, full, ['y', 'grad_q', 'grad_k', 'grad_v']):
        # (the bf16 backward is handed the ROUNDED y and a bf16 grad_y = w: 4e-3 as in the
        # operator test above)
        bound = 4e-3 * b.abs().max() + BF16_ULP * b.abs()
        assert ((a.float() - b).abs() <= bound).all(), name


def test_bf16_storage_outside_the_matrix_core_path_is_refused():
    from naive_gpt import layers
    attn = layers.SparseVanillaAttentionV2(d_head=32, d_codeword=8, n_codewords=16, p_dropout=0.0).cuda()
    q = torch.randn([2, 128, 16, 32], device='cuda').bfloat16()
    with pytest.raises(RuntimeError):
        attn(q, q, q, attn_mask=None)


def test_pq_loss_is_an_output_of_the_attention_node(monkeypatch):
    """An armed layer: the PQ loss of q / k is a second output of the attention's autograd node and
    its input gradient is added into the attention's grad_q / grad_k by the loss's backward kernel,
    inside that node -- against two separate autograd functions summed by autograd.  q also has a
    THIRD consumer here (a regulariser): whatever order autograd visits the consumers in, no
    contribution may be lost (round-2 ADVICE: the earlier process-global hand-over could drop one)."""
    from naive_gpt import layers
    from naive_gpt.layers.sparse import attention as A
    torch.manual_seed(0)
    attn = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0).cuda()
    q, k, v, w = [torch.randn([2, 256, 16, 64], device='cuda') for _ in range(4)]

    def run(third_first):
        qi, ki, vi = [t.clone().requires_grad_(True) for t in (q, k, v)]
        attn.zero_grad()
        attn.arm()
        extra = 0.25 * (qi * qi).sum() + (ki * w).sum() if third_first else None
        y = attn(qi, ki, vi, attn_mask=None)
        if extra is None:
            extra = 0.25 * (qi * qi).sum() + (ki * w).sum()
        ((y * w).sum() + 3.0 * attn.loss + extra).backward()
        return qi.grad, ki.grad, vi.grad, attn.quantizer.weight.grad.clone()

    assert A.FOLD_PQ_LOSS
    folded = [run(False), run(True)]
    assert attn.loss.grad_fn is not None and 'MfmaAttention' in type(attn.loss.grad_fn).__name__
    monkeypatch.setattr(A, 'FOLD_PQ_LOSS', False)
    plain = run(False)
    assert 'PQLoss' in str(type(attn.loss.grad_fn.next_functions[0][0]))
    for got in folded:
        for a, b in zip(got, plain):
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * b.abs().max().item())
    # q and k as two halves of ONE buffer (what the joint projection hands over): one pass of the
    # loss over both, forward and backward (spt_pq_loss_*_parts)
    from naive_gpt import ext
    monkeypatch.setattr(A, 'FOLD_PQ_LOSS', True)
    qk = torch.stack([q, k]).requires_grad_(True)
    vi = v.clone().requires_grad_(True)
    attn.zero_grad()
    attn.arm()
    ext.reset_paths()
    y = attn(qk[0], qk[1], vi, attn_mask=None)
    ((y * w).sum() + 3.0 * attn.loss + 0.25 * (qk[0] * qk[0]).sum() + (qk[1] * w).sum()).backward()
    assert ext.paths_taken('pq_loss_forward') == {('pq_loss_forward', 'pair'): 1}
    assert ext.paths_taken('pq_loss_backward') == {('pq_loss_backward', 'pair'): 1}
    for a, b in zip((qk.grad[0], qk.grad[1], vi.grad, attn.quantizer.weight.grad), plain):
        assert torch.allclose(a, b, rtol=1e-5, atol=1e-6 * b.abs().max().item())
    # only the loss is used: the attention's share of the gradients is zero, the loss's is there
    qi = q.clone().requires_grad_(True)
    attn.zero_grad()
    attn.arm()
    attn(qi, k, v, attn_mask=None)
    attn.loss.backward()
    assert qi.grad is not None and float(qi.grad.abs().max()) > 0
    qj = q.clone().requires_grad_(True)
    attn.quantizer.train_loss(qj).backward()
    assert torch.allclose(qi.grad, qj.grad, rtol=1e-5, atol=1e-7)


def test_host_side_arming_equals_the_reference_protocol():
    """module.arm() (no device read-back in the next forward) against the reference's
    `module.trigger.fill_(True)`: same loss buffer, trigger disarmed afterwards, one-shot."""
    from naive_gpt import layers
    torch.manual_seed(1)
    attn = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16,
                                           p_dropout=0.0).cuda()
    q, k, v = [torch.randn([2, 128, 16, 64], device='cuda') for _ in range(3)]
    attn.trigger.fill_(True)
    attn(q, k, v, attn_mask=None)
    want = attn.loss.clone()
    assert not bool(attn.trigger)
    attn.loss.zero_()
    attn.arm()
    attn(q, k, v, attn_mask=None)
    assert torch.equal(attn.loss, want) and not bool(attn.trigger)
    attn.loss.zero_()
    attn(q, k, v, attn_mask=None)                     # not armed any more: no loss computed
    assert float(attn.loss.detach()) == 0.0


# ------------------------------------------------------------------ compact tile layout

def _lookup_like(B, S, Z, gen):
    """Rows as lookup emits them: distinct columns <= row, then zero padding."""
    idx = torch.zeros([B, S, Z], dtype=torch.int32)
    for r in range(S):
        n = min(r + 1, Z)
        for b in range(B):
            idx[b, r, :n] = torch.randperm(r + 1, generator=gen)[:n].int()
    return idx.view(B, S * Z)


@pytest.mark.parametrize('builder', ['bits', 'counts'])
@pytest.mark.parametrize('N,H,S,Z,E', [(1, 4, 512, 64, 64), (1, 2, 2048, 256, 64), (1, 2, 256, 32, 128),
                                       (1, 3, 80, 8, 64)])
def test_compact_tiles_equal_full_tiles_on_lookup_patterns(N, H, S, Z, E, builder, monkeypatch):
    """SPT_TILES_COMPACT (128-byte mask slots + one byte-form slot per row tile for key tile 0)
    against SPT_TILES_FULL: bit-identical forward and backward on patterns whose only repeated
    column is the padding column 0, at a quarter of the workspace.  Both builders of the compact
    workspace (the bit-mask kernel, the default above S = 512, and the byte-count kernel)."""
    from naive_gpt import ext
    monkeypatch.setenv('SPT_CELL_TILES_BITS' if builder == 'bits' else 'SPT_CELL_TILES_COUNTS', '1')
    gen = torch.Generator().manual_seed(S + Z + E)
    B = N * H
    q, k, v = [torch.randn([N, S, H, E], generator=gen).cuda() for _ in range(3)]
    gy = torch.randn([B, S, E], generator=gen).cuda()
    idx = _lookup_like(B, S, Z, gen).cuda()
    full = ext.attention_mfma_prepare(idx, S)
    compact = ext.attention_mfma_prepare(idx, S, lookup_pattern=True)
    assert compact.layout == ext.TILES_COMPACT and not compact.broke_promise()
    if S >= 512:
        assert compact.buffer.numel() < 0.3 * full.buffer.numel()
    outs = []
    for tiles in (full, compact):
        y, row_sum = ext.attention_mfma_forward(tiles, q, k, v, E ** -0.5, CLAMP)
        grads = ext.attention_mfma_backward(tiles, q, k, v, y, gy, row_sum, E ** -0.5, CLAMP)
        outs.append((y, row_sum) + tuple(grads))
    for a, b, name in zip(outs[0], outs[1], ['y', 'row_sum', 'grad_q', 'grad_k', 'grad_v']):
        if name in ('grad_k', 'grad_v'):
            # the order in which a key's duplicate entries are summed is not fixed
            assert torch.allclose(a, b, rtol=1e-5, atol=1e-5 * float(a.abs().max())), name
        else:
            assert torch.equal(a, b), name
    # and it is the right answer
    want = oracle_chain(idx.cpu(), q.cpu(), k.cpu(), v.cpu(), E ** -0.5)[2]
    assert torch.allclose(outs[1][0].cpu(), want, rtol=1e-3, atol=1e-4)


@pytest.mark.parametrize('builder', ['bits', 'counts'])
def test_compact_tiles_flag_a_pattern_that_repeats_other_columns(builder, monkeypatch):
    from naive_gpt import ext
    monkeypatch.setenv('SPT_CELL_TILES_BITS' if builder == 'bits' else 'SPT_CELL_TILES_COUNTS', '1')
    gen = torch.Generator().manual_seed(8)
    idx = causal_indices(4, 256, 32, gen).cuda()          # random repeats anywhere
    tiles = ext.attention_mfma_prepare(idx, 256, lookup_pattern=True)
    assert tiles.broke_promise()
    assert not ext.attention_mfma_prepare(idx, 256).broke_promise()


def test_rotary_kernel_equals_the_layer_and_its_autograd():
    """spt_rotary: cos * x + sin * rotate_half(x) (position.py:24-34) of q and k in one launch, and the
    adjoint of both plus a copy of dv in one launch -- against layers.RotaryEmbedding and autograd."""
    from naive_gpt import ext, layers
    from naive_gpt.layers.sparse.attention import _Rotary3
    torch.manual_seed(0)
    N, S, H, E = 2, 300, 4, 128
    emb = layers.RotaryEmbedding(n_embeddings=512, d_model=E).cuda()
    ids = torch.arange(S, device='cuda')
    q, k, v = [torch.randn([N, S, H, E], device='cuda', requires_grad=True) for _ in range(3)]
    wq, wk, wv = [torch.randn([N, S, H, E], device='cuda') for _ in range(3)]
    rq, rk, rv = _Rotary3.apply(q, k, v, emb.cos_cached, emb.sin_cached)
    assert ext.back_to_back(rq, rk)
    ((rq * wq).sum() + (rk * wk).sum() + (rv * wv).sum()).backward()
    got = [rq.detach(), rk.detach(), q.grad.clone(), k.grad.clone(), v.grad.clone()]
    assert ext.spacing_of([q.grad, k.grad, v.grad]) == q.numel()       # one buffer, equally spaced
    for t in (q, k, v):
        t.grad = None
    tq, tk = emb(q, ids), emb(k, ids)
    ((tq * wq).sum() + (tk * wk).sum() + (v * wv).sum()).backward()
    want = [tq.detach(), tk.detach(), q.grad, k.grad, v.grad]
    for a, b in zip(got, want):
        assert torch.allclose(a, b, rtol=1e-6, atol=1e-6)
