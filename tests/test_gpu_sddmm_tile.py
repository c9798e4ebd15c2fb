"""GPU tier: `kernels.sddmm`'s matrix-core form (csrc/sddmm_tile.hip) against the CPU oracle and
against the fp32 gather form, at shapes that take it -- asserted through `ext.paths_taken`.

Reference: extension/sddmm.cpp:3-73 (cusparseSDDMM); the epilogue folds attention.py:125-127.
Bar (BASELINE.json north_star): fp32 values within 1e-3 relative; the form's own error is the
split's 2^-16 per product, so the tests hold it to 1e-4 * sqrt(E) absolute + 1e-4 relative.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import ragged_csr, uniform_csr

pytestmark = pytest.mark.gpu


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(got, want, rtol=1e-4, atol=8e-4):
    got = got.detach().cpu().numpy()
    assert got.shape == want.shape
    err = np.abs(got - want)
    bad = err > atol + rtol * np.abs(want)
    assert not bad.any(), 'max err {} at {} (want {})'.format(
        err.max(), np.unravel_index(err.argmax(), err.shape), want.flat[err.argmax()])


def took(ext, path):
    return ext.paths_taken('sddmm').get(('sddmm', path), 0)


def tiled(ext, *args, **kw):
    """One call that must take the matrix-core form."""
    before = took(ext, 'matrix_cores')
    out = ext.sddmm_forward_cuda(False, True, *args, **kw)
    assert took(ext, 'matrix_cores') == before + 1, 'the call took the gather form'
    return out


TILE_SHAPES = [
    # B, S, E, Z, causal
    (160, 512, 64, 64, True),     # configs[2] heads (5/8 of the batch): lookup's density, causal
    (160, 512, 64, 64, False),
    (160, 256, 64, 32, True),     # four live waves, four that only keep the barriers
    (160, 128, 64, 16, False),   # the smallest S: two live waves
    (80, 1024, 64, 128, True),    # two key blocks per slice; rows of 128 entries: the long-row loop
    (160, 496, 64, 40, False),    # S not a multiple of 32: a ragged last stripe and a ragged last key tile
    (160, 200, 64, 20, False),    # ... nor of 8
]


@pytest.mark.parametrize('B,S,E,Z,causal', TILE_SHAPES)
def test_matrix_core_form_against_the_oracle(B, S, E, Z, causal):
    from naive_gpt import ext
    rng = np.random.default_rng(B + S + Z)
    indptr, idx = uniform_csr(rng, 4, S, Z, causal)
    idx = np.ascontiguousarray(np.tile(idx, [B // 4, 1]))
    q = rng.standard_normal([B, S, E]).astype(np.float32)
    k = rng.standard_normal([B, S, E]).astype(np.float32)
    want = O.sddmm_forward(indptr, idx, q, k)
    close(tiled(ext, dev(indptr), dev(idx), dev(q), dev(k)), want)
    # fused epilogue of attention.py:125-127
    got = tiled(ext, dev(indptr), dev(idx), dev(q), dev(k), scale=E ** -0.5, clamp=1.0)
    close(got, np.clip(want * np.float32(E ** -0.5), -1.0, 1.0), atol=1e-4)


def test_matrix_core_form_against_the_gather_form(monkeypatch):
    from naive_gpt import ext
    B, S, E, Z = 160, 512, 64, 64
    rng = np.random.default_rng(7)
    indptr, idx = uniform_csr(rng, 2, S, Z, True)
    idx = np.ascontiguousarray(np.tile(idx, [B // 2, 1]))
    q, k = (dev(rng.standard_normal([B, S, E]).astype(np.float32)) for _ in range(2))
    a = tiled(ext, dev(indptr), dev(idx), q, k, scale=0.125, clamp=10.0)
    monkeypatch.setenv('SPT_SDDMM_GATHER', '1')          # (read by the library at every call)
    before = took(ext, 'gather')
    b = ext.sddmm_forward_cuda(False, True, dev(indptr), dev(idx), q, k, scale=0.125, clamp=10.0)
    assert took(ext, 'gather') == before + 1
    # three bf16 MFMAs per fp32 product against an fp32 FMA chain: 2^-16 of |q| |k| per product
    assert float((a - b).abs().max()) < 2e-4
    assert float((a - b).abs().max()) > 0.0              # (two different kernels did run)


def test_matrix_core_form_on_ragged_rows():
    """Empty rows, duplicate columns, rows longer than a wave (the LONG instantiation) and
    shorter ones in one pattern; columns anywhere in [0, S)."""
    from naive_gpt import ext
    B, S, E = 160, 256, 64
    rng = np.random.default_rng(11)
    indptr, idx = ragged_csr(rng, B, S, 150)              # mean 75 entries per row: density 0.29
    assert int(np.diff(indptr).max()) > 64 and int(np.diff(indptr).min()) == 0
    q = rng.standard_normal([B, S, E]).astype(np.float32)
    k = rng.standard_normal([B, S, E]).astype(np.float32)
    close(tiled(ext, dev(indptr), dev(idx), dev(q), dev(k)), O.sddmm_forward(indptr, idx, q, k))


def test_matrix_core_form_on_short_ragged_rows():
    """No row longer than a wave: the branch-free instantiation, with empty rows and duplicates."""
    from naive_gpt import ext
    B, S, E = 160, 256, 64
    rng = np.random.default_rng(12)
    indptr, idx = ragged_csr(rng, B, S, 64)
    assert int(np.diff(indptr).max()) <= 64 and indptr[-1] * 16 >= S * S
    q = rng.standard_normal([B, S, E]).astype(np.float32)
    k = rng.standard_normal([B, S, E]).astype(np.float32)
    close(tiled(ext, dev(indptr), dev(idx), dev(q), dev(k)), O.sddmm_forward(indptr, idx, q, k))


def test_matrix_core_form_in_the_head_layout():
    """q / k as slices of [N, S, H, E] (the attention layers' tensors), either or both."""
    from naive_gpt import ext
    N, S, H, E, Z = 10, 256, 16, 64, 32
    rng = np.random.default_rng(13)
    indptr, idx = uniform_csr(rng, 4, S, Z, True)
    idx = dev(np.ascontiguousarray(np.tile(idx, [N * H // 4, 1])))
    q4 = torch.randn([N, S, H, E], device='cuda')
    k4 = torch.randn([N, S, H, E], device='cuda')

    def flat(t):
        return t.transpose(1, 2).reshape(N * H, S, E).contiguous()

    a = tiled(ext, dev(indptr), idx, q4, k4, query_heads=H, key_heads=H)
    b = tiled(ext, dev(indptr), idx, flat(q4), flat(k4))
    c = tiled(ext, dev(indptr), idx, flat(q4), k4, key_heads=H)
    assert torch.equal(a, b) and torch.equal(a, c)
    want = O.sddmm_forward(indptr.copy(), idx.cpu().numpy(), flat(q4).cpu().numpy(), flat(k4).cpu().numpy())
    close(a, want)


def test_sparse_or_small_calls_keep_the_gather_form():
    from naive_gpt import ext
    lib = ext.load_library()
    assert lib.spt_sddmm_form(256, 512, 64, 512 * 64) == 1
    assert lib.spt_sddmm_form(256, 512, 64, 512 * 16) == 0       # density 1 / 32
    assert lib.spt_sddmm_form(256, 512, 128, 512 * 64) == 0      # d_head 128
    assert lib.spt_sddmm_form(128, 512, 64, 512 * 64) == 0       # too few workgroups: the gather form's finer split wins
    assert lib.spt_sddmm_form(256, 64, 64, 64 * 8) == 0          # S < 128
    assert lib.spt_sddmm_form(128, 1024, 64, 1024 * 128) == 1    # two key blocks per slice
    assert lib.spt_sddmm_form(64, 2048, 64, 2048 * 256) == 0     # four: every block would walk all column ids
