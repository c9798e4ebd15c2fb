"""GPU tier (`-m gpu`): every HIP operator, called through the C ABI
(naive_gpt.ext -> libspt_hip.so), against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): PQ codes / top-k indices / CSR structure
bit-exact; fp32 values within 1e-3 relative (tolerances are written per test).
Shapes follow the reference's tests (test/kernel/*.py) plus the configs of
BASELINE.json, at sizes the oracle finishes in seconds.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import ragged_csr, uniform_csr

pytestmark = pytest.mark.gpu

RTOL = 1e-3   # north_star: fp32 values within 1e-3 rel
ATOL = 1e-4   # absolute floor for values near zero (sums of O(10) terms of O(1))


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(got: torch.Tensor, want: np.ndarray, rtol=RTOL, atol=ATOL):
    got = got.detach().cpu().numpy()
    assert got.shape == want.shape
    err = np.abs(got - want)
    tol = atol + rtol * np.abs(want)
    bad = err > tol
    assert not bad.any(), 'max err {} at {} (want {})'.format(
        err.max(), np.unravel_index(err.argmax(), err.shape), want.flat[err.argmax()])


def test_library_is_the_native_one():
    from naive_gpt import ext
    lib = ext.load_library()
    assert lib.spt_abi_version() == ext.ABI_VERSION
    assert torch.cuda.is_available()
    assert 'gfx950' in torch.cuda.get_device_properties(0).gcnArchName


# ------------------------------------------------------------------ cdist

@pytest.mark.parametrize('M,NQ,C,D', [
    (8, 4096, 16, 8),      # BERT-large head: E=64 -> M=8, D=8, C=16
    (16, 2048, 16, 8),     # LLaMA head: E=128
    (3, 64 * 5, 16 * 3, 4),  # reference test distribution (test_cdist.py:8-12)
    (1, 64, 256, 8),
    (5, 1000, 20, 16),     # NQ, C not multiples of 16: fine here
    (2, 192, 16, 32), (2, 128, 32, 24),
])
def test_cdist_forward(M, NQ, C, D):
    from naive_gpt import ext
    rng = np.random.default_rng(M * 7 + NQ)
    q = rng.standard_normal([M, NQ, D]).astype(np.float32)
    t = rng.standard_normal([M, C, D]).astype(np.float32)
    dist_o, idx_o = O.cdist_forward(q, t)
    dist, idx = ext.cdist_forward_cuda(dev(q), dev(t))
    # same fp32 summation order -> bit-exact distances and codes
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert np.array_equal(dist.cpu().numpy(), dist_o)
    assert np.array_equal(ext.cdist_encode(dev(q), dev(t)).cpu().numpy(), idx_o)


def test_cdist_forward_ties_and_sentinel():
    from naive_gpt import ext
    q = np.zeros([2, 64, 8], np.float32)
    t = np.ones([2, 16, 8], np.float32)
    t[1, 9] = 0.25
    t[1, 12] = 0.25                     # tie between 9 and 12 -> 9
    q[0, 5] = 1e14                      # nothing beats the 1e13 start value -> 0
    _, idx_o = O.cdist_forward(q, t)
    _, idx = ext.cdist_forward_cuda(dev(q), dev(t))
    assert np.array_equal(idx.cpu().numpy(), idx_o)
    assert (idx_o[0] == 0).all() and (idx_o[1] == 9).all()


@pytest.mark.parametrize('M,NQ,C,D', [(8, 4096, 16, 8), (2, 320, 48, 4), (16, 1024, 16, 8),
                                      (1, 64, 16, 16), (3, 8192 + 64, 32, 8)])
def test_cdist_backward(M, NQ, C, D):
    from naive_gpt import ext
    rng = np.random.default_rng(NQ + C)
    q = rng.standard_normal([M, NQ, D]).astype(np.float32)
    t = rng.standard_normal([M, C, D]).astype(np.float32)
    go = rng.standard_normal([M, NQ, C]).astype(np.float32)
    gq_o, gt_o = O.cdist_backward(q, t, go)
    gq, gt = ext.cdist_backward_cuda(dev(q), dev(t), dev(go))
    close(gq, gq_o)
    # grad_table sums NQ signed terms: tolerance relative to the sum of magnitudes
    scale = np.abs(go).sum(axis=1)[:, :, None]          # [M, C, 1]
    err = np.abs(gt.cpu().numpy() - gt_o)
    assert (err <= 1e-5 * scale + 1e-4).all(), err.max()


# ------------------------------------------------------------------ lookup

@pytest.mark.parametrize('B,S,M,hi,coeff', [
    (4, 512, 8, 16, 8),     # config 2/3: PQ codes in [0,16), Z = 64
    (16, 512, 8, 8, 8),     # reference test distribution (test_lookup.py:36-43)
    (2, 1024, 8, 8, 8),
    (2, 256, 8, 2, 8),      # dense slots: cursor saturation + overwrite quirks
    (2, 512, 8, 1, 8),      # every column matches fully
    (2, 256, 10, 4, 8), (2, 256, 16, 3, 8), (1, 128, 5, 2, 4), (3, 64, 8, 4, 4),
    (1, 2048, 8, 16, 8),    # config 4: Z = 256 (beyond the reference's templates)
    (1, 2048, 16, 16, 8),   # config 5: M = 16
    # the long-row kernel (S > 512): saturated lists + the overwrite quirk, every column a full
    # match, M = 10, codes beyond 4 bits (exact uint16 form), a last row block of 32 rows
    (2, 1024, 8, 2, 8), (1, 1024, 8, 1, 8), (1, 2048, 8, 2, 8), (1, 1024, 10, 4, 8),
    (1, 1024, 16, 40, 8), (1, 768, 8, 300, 4), (2, 544, 8, 3, 2),
])
def test_lookup_bit_exact(B, S, M, hi, coeff):
    from naive_gpt import ext
    rng = np.random.default_rng(S * 3 + M + hi)
    q = rng.integers(0, hi, [B, S, M], dtype=np.int32)
    k = rng.integers(0, hi, [B, S, M], dtype=np.int32)
    want = O.lookup_forward(q, k, coeff)
    got = ext.lookup_forward_cuda(torch.empty([coeff]), dev(q), dev(k))
    assert got.dtype == torch.int32 and tuple(got.shape) == want.shape
    assert np.array_equal(got.cpu().numpy(), want)


def test_lookup_uint16_truncation_and_errors():
    from naive_gpt import ext
    rng = np.random.default_rng(1)
    q = rng.integers(0, 4, [1, 64, 8], dtype=np.int32)
    k = rng.integers(0, 4, [1, 64, 8], dtype=np.int32)
    want = O.lookup_forward(q, k, 4)
    got = ext.lookup_forward_cuda(torch.empty([4]), dev(q + 65536), dev(k - 131072))
    assert np.array_equal(got.cpu().numpy(), want)
    with pytest.raises(RuntimeError):
        ext.lookup_forward_cuda(torch.empty([8]), dev(q), dev(k))       # Z = 8
    with pytest.raises(RuntimeError):
        ext.lookup_forward_cuda(torch.empty([4]), dev(q), dev(k[:, :32]))
    with pytest.raises(RuntimeError):
        ext.lookup_forward_cuda(torch.empty([4]), dev(q).float(), dev(k))


def test_lookup_from_real_pq_codes_end_to_end():
    """cdist -> lookup on float inputs: the structure the layer actually builds."""
    from naive_gpt import ext
    rng = np.random.default_rng(2)
    B, S, E = 4, 512, 64
    table = rng.standard_normal([8, 16, 8]).astype(np.float32)
    q = rng.standard_normal([B * S, E]).astype(np.float32)
    k = rng.standard_normal([B * S, E]).astype(np.float32)

    def codes_cpu(z):
        zf = np.ascontiguousarray(z.reshape(B * S, 8, 8).transpose(1, 0, 2))
        return np.ascontiguousarray(O.cdist_forward(zf, table, False)[1].T).reshape(B, S, 8)

    def codes_gpu(z):
        zf = dev(z).view(B * S, 8, 8).transpose(0, 1).contiguous()
        return ext.cdist_encode(zf, dev(table)).t().contiguous().view(B, S, 8)

    want = O.lookup_forward(codes_cpu(q), codes_cpu(k), 8)
    got = ext.lookup_forward_cuda(torch.empty([8]), codes_gpu(q), codes_gpu(k))
    assert np.array_equal(got.cpu().numpy(), want)


# ------------------------------------------------------------------ sddmm / spmm

SPARSE_SHAPES = [
    # B, S, E, Z, causal
    (32, 512, 64, 64, True),     # config 2 slice -> LDS-resident kernels
    (48, 256, 64, 32, True),     # LDS-resident, rows split over blocks
    (3, 512, 64, 64, True),      # few batches -> gather-from-global kernels
    (2, 2048, 64, 256, True),    # config 4: tile does not fit LDS -> chunked transposed form
    (3, 1024, 64, 128, True),    # two 512-row chunks
    (2, 1536, 64, 48, False),    # three chunks, non-causal pattern (every chunk reaches every column)
    (2, 512, 128, 64, True),     # config 5 head dim
    (5, 16 * 7, 48, 14, False),  # reference test distribution (test_sddmm.py:49-53)
    (4, 16, 16, 2, False), (2, 256, 32, 32, False), (40, 128, 16, 16, False),
    (33, 64, 4, 8, False), (1, 64, 256, 8, False),
]


@pytest.mark.parametrize('B,S,E,Z,causal', SPARSE_SHAPES)
def test_sddmm_forward(B, S, E, Z, causal):
    from naive_gpt import ext
    rng = np.random.default_rng(B + S + E)
    indptr, idx = uniform_csr(rng, min(B, 4), S, Z, causal)
    idx = np.ascontiguousarray(np.tile(idx, [(B + 3) // 4, 1])[:B])
    q = rng.standard_normal([B, S, E]).astype(np.float32)
    k = rng.standard_normal([B, S, E]).astype(np.float32)
    want = O.sddmm_forward(indptr, idx, q, k)
    got = ext.sddmm_forward_cuda(torch.scalar_tensor(False), torch.scalar_tensor(True),
                                 dev(indptr), dev(idx), dev(q), dev(k))
    close(got, want, atol=1e-4 * np.sqrt(E))
    # fused epilogue of attention.py:125-127
    got2 = ext.sddmm_forward_cuda(False, True, dev(indptr), dev(idx), dev(q), dev(k),
                                  scale=E ** -0.5, clamp=1.0)
    close(got2, np.clip(want * np.float32(E ** -0.5), -1.0, 1.0))


@pytest.mark.parametrize('B,S,E,Z,causal', SPARSE_SHAPES)
@pytest.mark.parametrize('trans', [False, True])
def test_spmm_forward(B, S, E, Z, causal, trans):
    from naive_gpt import ext
    rng = np.random.default_rng(B + S + E + 1)
    indptr, idx = uniform_csr(rng, min(B, 4), S, Z, causal)
    idx = np.ascontiguousarray(np.tile(idx, [(B + 3) // 4, 1])[:B])
    vals = rng.standard_normal([B, S * Z]).astype(np.float32)
    x = rng.standard_normal([B, S, E]).astype(np.float32)
    want = O.spmm_forward(trans, indptr, idx, vals, x)
    got = ext.spmm_forward_cuda(torch.scalar_tensor(trans), torch.scalar_tensor(False),
                                dev(indptr), dev(idx), dev(vals), dev(x))
    # a row of y sums up to S*Z/S..S terms of O(1): absolute floor scales with sqrt(terms)
    close(got, want, atol=2e-4 * np.sqrt(max(Z, 16)) * (4 if trans and causal else 1))


@pytest.mark.parametrize('B,S,E', [(2, 48, 16), (40, 64, 64), (3, 80, 32)])
def test_sparse_ops_on_ragged_csr_with_duplicates_and_empty_rows(B, S, E):
    from naive_gpt import ext
    rng = np.random.default_rng(S)
    indptr, idx = ragged_csr(rng, B, S, 11)
    vals = rng.standard_normal(idx.shape).astype(np.float32)
    x = rng.standard_normal([B, S, E]).astype(np.float32)
    q = rng.standard_normal([B, S, E]).astype(np.float32)
    args = (dev(indptr), dev(idx))
    close(ext.sddmm_forward_cuda(False, True, *args, dev(q), dev(x)),
          O.sddmm_forward(indptr, idx, q, x), atol=1e-3)
    close(ext.spmm_forward_cuda(False, False, *args, dev(vals), dev(x)),
          O.spmm_forward(False, indptr, idx, vals, x), atol=1e-3)
    close(ext.spmm_forward_cuda(True, False, *args, dev(vals), dev(x)),
          O.spmm_forward(True, indptr, idx, vals, x), atol=1e-3)
    y = ext.softmax_forward_cuda(*args, dev(vals))
    y_o = O.softmax_forward(indptr, idx, vals)
    close(y, y_o)
    dy = rng.standard_normal(idx.shape).astype(np.float32)
    close(ext.softmax_backward_cuda(*args, dev(y_o), dev(dy)),
          O.softmax_backward(indptr, idx, y_o, dy))


def test_transposed_structure_is_shared_and_reproducible():
    """csr_transpose + spmm_transposed == spmm(trans_lhs=True); one build serves both
    A^T products of a backward.  Columns without duplicate (row, column) entries are
    summed in ascending row order, bitwise reproducibly; column 0 here carries the
    padding duplicates of short causal rows, whose order may vary."""
    from naive_gpt import ext
    rng = np.random.default_rng(21)
    B, S, E, Z = 32, 512, 64, 64
    indptr, idx = uniform_csr(rng, 2, S, Z, True)
    idx = np.ascontiguousarray(np.tile(idx, [B // 2, 1]))
    vals = rng.standard_normal([B, S * Z]).astype(np.float32)
    x = rng.standard_normal([B, S, E]).astype(np.float32)
    d_indptr, d_idx, d_vals, d_x = dev(indptr), dev(idx), dev(vals), dev(x)
    t1 = ext.transposed_for(d_indptr, d_idx)
    assert ext.transposed_for(d_indptr, d_idx) is t1          # memoised on the tensor
    y1 = ext.spmm_transposed(t1, d_indptr, d_idx, d_vals, d_x)
    y2 = ext.spmm_forward_cuda(True, False, d_indptr, d_idx, d_vals, d_x)
    y3 = ext.spmm_transposed(ext.csr_transpose(d_indptr, d_idx), d_indptr, d_idx, d_vals, d_x)
    assert torch.equal(y1[:, 1:], y2[:, 1:]) and torch.equal(y1[:, 1:], y3[:, 1:])
    assert torch.allclose(y1, y2, rtol=1e-4, atol=1e-3)
    close(y1, O.spmm_forward(True, indptr, idx, vals, x), atol=2e-3)
    d_idx[0, 5] = 0                                            # in-place edit invalidates
    assert ext.transposed_for(d_indptr, d_idx) is not t1


# ------------------------------------------------------------------ softmax

@pytest.mark.parametrize('B,S,Z', [(32, 512, 64), (4, 1024, 128), (2, 2048, 256), (7, 64, 8),
                                   (3, 128, 16), (2, 320, 40), (1, 4096 // 8, 4)])
def test_softmax_forward_backward(B, S, Z):
    from naive_gpt import ext
    rng = np.random.default_rng(S + Z)
    indptr, idx = uniform_csr(rng, min(B, 2), S, Z, True)
    idx = np.ascontiguousarray(np.tile(idx, [(B + 1) // 2, 1])[:B])
    # values over the clamp range of attention.py:125-127
    vals = rng.uniform(-10, 10, [B, S * Z]).astype(np.float32)
    y_o = O.softmax_forward(indptr, idx, vals)
    y = ext.softmax_forward_cuda(dev(indptr), dev(idx), dev(vals))
    close(y, y_o, atol=1e-7)
    dy = rng.standard_normal([B, S * Z]).astype(np.float32)
    dv_o = O.softmax_backward(indptr, idx, y_o, dy)
    dv = ext.softmax_backward_cuda(dev(indptr), dev(idx), dev(y_o), dev(dy))
    close(dv, dv_o, atol=1e-6)


def test_softmax_quirks_on_gpu():
    from naive_gpt import ext
    indptr = np.array([0, 4, 8], np.int32)
    idx = np.array([[0, 1, 1, 0, 0, 1, 1, 5], [3, 3, 3, 3, 0, 1, 1, 5]], np.int32)
    vals = np.zeros([2, 8], np.float32)
    y_o = O.softmax_forward(indptr, idx, vals)
    y = ext.softmax_forward_cuda(dev(indptr), dev(idx), dev(vals))
    close(y, y_o, atol=1e-7)
    dy = -np.ones([2, 8], np.float32)     # sum(y*dy) < 0 -> clamped at 1e-9 (softmax.cu:69)
    close(ext.softmax_backward_cuda(dev(indptr), dev(idx), dev(y_o), dev(dy)),
          O.softmax_backward(indptr, idx, y_o, dy), atol=1e-7)


# ------------------------------------------------------------------ autograd wiring

def test_kernels_autograd_matches_dense_reference_formulas():
    """kernels.* with autograd vs dense torch, as test/kernel/test_{sddmm,spmm}.py do."""
    from naive_gpt import kernels
    rng = np.random.default_rng(9)
    B, S, E, Z = 36, 128, 32, 16
    indptr, idx = uniform_csr(rng, B, S, Z, False)
    q = torch.randn([B, S, E], device='cuda', requires_grad=True)
    k = torch.randn([B, S, E], device='cuda', requires_grad=True)
    rows = torch.arange(S, device='cuda').repeat_interleave(Z)
    cols = dev(idx).long()
    mask = torch.zeros([B, S, S], device='cuda')
    mask[torch.arange(B, device='cuda')[:, None], rows[None, :], cols] = 1.0

    y1 = mask * torch.matmul(q, k.transpose(-1, -2))
    y1.sum().backward()
    gq1, gk1 = q.grad.clone(), k.grad.clone()
    q.grad = k.grad = None
    y2 = kernels.sddmm(dev(indptr), dev(idx), query=q, key=k)
    y2.sum().backward()
    assert torch.allclose(y1[torch.arange(B, device='cuda')[:, None], rows[None, :], cols],
                          y2, atol=1e-3)
    assert torch.allclose(gq1, q.grad, atol=1e-3)
    assert torch.allclose(gk1, k.grad, atol=1e-3)

    vals = torch.rand([B, S * Z], device='cuda', requires_grad=True)
    x = torch.randn([B, S, E], device='cuda', requires_grad=True)
    dense = torch.zeros([B, S, S], device='cuda').index_put(
        (torch.arange(B, device='cuda')[:, None].expand(B, S * Z), rows[None, :].expand(B, S * Z),
         cols), vals)
    w = torch.randn([B, S, E], device='cuda')
    (torch.matmul(dense, x) * w).sum().backward()
    gv1, gx1 = vals.grad.clone(), x.grad.clone()
    vals.grad = x.grad = None
    y = kernels.spmm(dev(indptr), dev(idx), vals, x=x)
    (y * w).sum().backward()
    assert torch.allclose(gv1, vals.grad, atol=1e-3)
    assert torch.allclose(gx1, x.grad, atol=1e-3)
    with pytest.raises(NotImplementedError):
        kernels.lookup.__globals__['Lookup'].backward(None, None)


# ------------------------------------------------------------------ head layout ([N,S,H,E])

def test_pq_encode_heads_equals_reference_layout_chain():
    from naive_gpt import ext
    rng = np.random.default_rng(31)
    N, S, H, E, M, C, D = 3, 128, 4, 64, 8, 16, 8
    z = rng.standard_normal([N, S, H, E]).astype(np.float32)
    table = rng.standard_normal([M, C, D]).astype(np.float32)
    # reference chain: [N,S,H,E] -> [N*H,S,E] -> [M,NQ,D] -> argmin -> [N*H,S,M]
    zb = np.ascontiguousarray(z.transpose(0, 2, 1, 3)).reshape(N * H * S, M, D)
    want = O.cdist_forward(np.ascontiguousarray(zb.transpose(1, 0, 2)), table, False)[1]
    want = np.ascontiguousarray(want.T).reshape(N * H, S, M)
    got = ext.pq_encode_heads(dev(z), dev(table))
    assert got.dtype == torch.int32 and np.array_equal(got.cpu().numpy(), want)


@pytest.mark.parametrize('N,S,H,E', [(8, 256, 4, 64), (16, 512, 16, 64), (4, 128, 8, 128)])
def test_head_layout_operands_equal_contiguous_ones(N, S, H, E):
    from naive_gpt import ext
    rng = np.random.default_rng(N + S + H)
    B, Z = N * H, S // 8
    indptr, idx = uniform_csr(rng, 2, S, Z, True)
    idx = np.ascontiguousarray(np.tile(idx, [B // 2, 1]))
    d_indptr, d_idx = dev(indptr), dev(idx)
    q4 = torch.randn([N, S, H, E], device='cuda')
    k4 = torch.randn([N, S, H, E], device='cuda')
    vals = torch.randn([B, S * Z], device='cuda')

    def flat(t):
        return t.transpose(1, 2).contiguous().view(B, S, E)

    def heads(t):                        # [B,S,E] -> [N,S,H,E]
        return t.view(N, H, S, E).transpose(1, 2).contiguous()

    a = ext.sddmm_forward_cuda(False, True, d_indptr, d_idx, q4, k4, query_heads=H, key_heads=H)
    b = ext.sddmm_forward_cuda(False, True, d_indptr, d_idx, flat(q4), flat(k4))
    assert torch.equal(a, b)
    c = ext.sddmm_forward_cuda(False, True, d_indptr, d_idx, flat(q4), k4, key_heads=H)
    assert torch.equal(c, b)
    y1 = ext.spmm_forward_cuda(False, False, d_indptr, d_idx, vals, k4, x_heads=H, y_heads=H)
    y0 = ext.spmm_forward_cuda(False, False, d_indptr, d_idx, vals, flat(k4))
    assert y1.shape == (N, S, H, E) and torch.equal(y1, heads(y0))
    y2 = ext.spmm_forward_cuda(False, False, d_indptr, d_idx, vals, k4, x_heads=H, y_heads=0)
    assert torch.equal(y2, y0)
    t = ext.transposed_for(d_indptr, d_idx)
    t1 = ext.spmm_transposed(t, d_indptr, d_idx, vals, k4, x_heads=H, y_heads=H)
    t0 = ext.spmm_transposed(t, d_indptr, d_idx, vals, flat(k4))
    assert torch.allclose(t1, heads(t0), rtol=1e-5, atol=1e-4)
    t2 = ext.spmm_transposed(t, d_indptr, d_idx, vals, flat(k4), x_heads=0, y_heads=H)
    assert torch.allclose(t2, heads(t0), rtol=1e-5, atol=1e-4)


def test_softmax_backward_clamped_is_the_chained_gradient():
    from naive_gpt import ext
    rng = np.random.default_rng(77)
    B, S, Z = 8, 256, 32
    indptr, idx = uniform_csr(rng, 2, S, Z, True)
    idx = np.ascontiguousarray(np.tile(idx, [B // 2, 1]))
    raw = rng.uniform(-14, 14, [B, S * Z]).astype(np.float32)
    scale, clamp = 0.7, 10.0
    scores = np.clip(raw * np.float32(scale), -clamp, clamp).astype(np.float32)
    y = O.softmax_forward(indptr, idx, scores)
    dy = rng.standard_normal([B, S * Z]).astype(np.float32)
    want = O.softmax_backward(indptr, idx, y, dy)
    want = np.where(np.abs(scores) < clamp, want * np.float32(scale), 0).astype(np.float32)
    got = ext.softmax_backward_clamped(dev(indptr), dev(idx), dev(y), dev(dy), dev(scores),
                                       scale, clamp)
    close(got, want, atol=1e-6)


def test_layer_head_layout_path_equals_copying_path():
    """SparseVanillaAttentionV2 on a shape that takes the [N,S,H,E] fast path, against the
    same layer forced through the reference-style transposing path."""
    from naive_gpt import ext, layers
    torch.manual_seed(4)
    N, S, H, E = 8, 256, 4, 64
    attn = layers.SparseVanillaAttentionV2(d_head=E, d_codeword=8, n_codewords=16,
                                           p_dropout=0.0).cuda()
    q, k, v = [torch.randn([N, S, H, E], device='cuda', requires_grad=True) for _ in range(3)]
    w = torch.randn([N, S, H, E], device='cuda')

    def run(trigger):
        for t in (q, k, v):
            t.grad = None
        attn.zero_grad()
        if trigger:
            attn.trigger.fill_(True)
        y = attn(q, k, v, attn_mask=None)
        loss = (y * w).sum()
        if trigger:
            loss = loss + 1e-2 * attn.loss
        loss.backward()
        return [y.detach().clone(), q.grad.clone(), k.grad.clone(), v.grad.clone(),
                attn.quantizer.weight.grad.clone() if trigger else None]

    assert ext.head_layout_supported(S, E, N * H)
    for trigger in (False, True):
        fast = run(trigger)
        orig = ext.head_layout_supported
        ext.head_layout_supported = lambda *a: False
        try:
            slow = run(trigger)
        finally:
            ext.head_layout_supported = orig
        for a, b in zip(fast, slow):
            if a is not None:
                assert torch.allclose(a, b, rtol=1e-3, atol=2e-4)


def test_chunked_transposed_product_head_layout_and_reuse():
    """S = 1024 at E = 64: one chunked build serves both products of a backward, with the
    head-layout operands of the attention layer; equal to the plain-layout result."""
    from naive_gpt import ext
    rng = np.random.default_rng(5)
    N, H, S, E, Z = 2, 4, 1024, 64, 128
    B = N * H
    indptr, idx = uniform_csr(rng, B, S, Z, True)
    vals = rng.standard_normal([B, S * Z]).astype(np.float32)
    x4 = torch.from_numpy(rng.standard_normal([N, S, H, E]).astype(np.float32)).cuda()
    flat = lambda t: t.transpose(1, 2).contiguous().view(B, S, E)       # noqa: E731
    d_indptr, d_idx, d_vals = dev(indptr), dev(idx), dev(vals)
    t = ext.transposed_for(d_indptr, d_idx, E)
    assert ext.transposed_for(d_indptr, d_idx, E) is t
    y_flat = ext.spmm_transposed(t, d_indptr, d_idx, d_vals, flat(x4))
    close(y_flat, O.spmm_forward(True, indptr, idx, vals, flat(x4).cpu().numpy()), atol=5e-3)
    y_heads = ext.spmm_transposed(t, d_indptr, d_idx, d_vals, x4, x_heads=H, y_heads=H)
    assert y_heads.shape == (N, S, H, E)
    assert torch.allclose(flat(y_heads), y_flat, rtol=1e-5, atol=1e-4)
    # a structure built for another head size is refused, not misread
    with pytest.raises(RuntimeError):
        ext.spmm_transposed(ext.csr_transpose(d_indptr, d_idx, 128), d_indptr, d_idx, d_vals, flat(x4))
