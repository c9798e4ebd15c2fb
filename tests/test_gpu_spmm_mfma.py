"""GPU tier: `kernels.spmm`'s matrix-core form (csrc/mfma_attention.hip: spmm_mfma_kernel; opt-in through
SPT_SPMM_MFMA=1, which the library reads at every call) against the CPU oracle and against the default
fp32 gather form, at shapes that take it -- asserted through `ext.paths_taken`.

Reference: extension/spmm.cpp:3-72 (cusparseSpMM, non-transposed).
Bar (BASELINE.json north_star): fp32 values within 1e-3 relative; the form's own error is the split's
2^-16 per product, so the tests hold it to 1e-4 relative + 2e-4 * sqrt(row length) absolute.
"""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import uniform_csr

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def opted_in(monkeypatch):
    monkeypatch.setenv('SPT_SPMM_MFMA', '1')


def dev(a):
    return torch.from_numpy(np.ascontiguousarray(a)).cuda()


def close(got, want, rtol=1e-4, atol=2e-3):
    got = got.detach().cpu().numpy()
    assert got.shape == want.shape
    err = np.abs(got - want)
    bad = err > atol + rtol * np.abs(want)
    assert not bad.any(), 'max err {} at {} (want {})'.format(
        err.max(), np.unravel_index(err.argmax(), err.shape), want.flat[err.argmax()])


def took(ext, path):
    return ext.paths_taken('spmm').get(('spmm', path), 0)


def tiled(ext, *args, **kw):
    """One call that must take the matrix-core form."""
    before = took(ext, 'matrix_cores')
    out = ext.spmm_forward_cuda(False, False, *args, **kw)
    assert took(ext, 'matrix_cores') == before + 1, 'the call took the gather form'
    return out


TILE_SHAPES = [
    # B, S, E, Z, causal
    (32, 512, 64, 64, True),      # configs[1] heads: lookup's density, causal (column-0 duplicates)
    (32, 512, 64, 64, False),
    (8, 256, 64, 32, True),       # one workgroup per slice
    (8, 128, 64, 16, False),      # half a workgroup's waves have rows
    (6, 1024, 64, 64, False),     # density 1 / 16: the dispatcher's edge; 32 key tiles
    (8, 496, 64, 40, False),      # S not a multiple of 32: a ragged last row tile and key tile
    (8, 200, 64, 20, False),      # ... nor of 8
    (8, 512, 128, 64, True),      # d_head 128 (spt::e128)
    (4, 72, 128, 8, False),
]


@pytest.mark.parametrize('B,S,E,Z,causal', TILE_SHAPES)
def test_matrix_core_form_against_the_oracle(B, S, E, Z, causal):
    from naive_gpt import ext
    rng = np.random.default_rng(B + S + Z)
    indptr, idx = uniform_csr(rng, 2, S, Z, causal)
    idx = np.ascontiguousarray(np.tile(idx, [B // 2, 1]))
    vals = rng.standard_normal([B, S * Z]).astype(np.float32)
    x = rng.standard_normal([B, S, E]).astype(np.float32)
    want = O.spmm_forward(False, indptr, idx, vals, x)
    close(tiled(ext, dev(indptr), dev(idx), dev(vals), dev(x)), want, atol=2e-4 * np.sqrt(Z))


def test_matrix_core_form_against_the_gather_form(monkeypatch):
    from naive_gpt import ext
    B, S, E, Z = 32, 512, 64, 64
    rng = np.random.default_rng(7)
    indptr, idx = uniform_csr(rng, 2, S, Z, True)
    idx = np.ascontiguousarray(np.tile(idx, [B // 2, 1]))
    vals, x = dev(rng.standard_normal([B, S * Z]).astype(np.float32)), dev(rng.standard_normal([B, S, E]).astype(np.float32))
    a = tiled(ext, dev(indptr), dev(idx), vals, x)
    monkeypatch.delenv('SPT_SPMM_MFMA')                 # (read by the library at every call)
    before = took(ext, 'gather')
    b = ext.spmm_forward_cuda(False, False, dev(indptr), dev(idx), vals, x)
    assert took(ext, 'gather') == before + 1
    assert torch.allclose(a, b, rtol=1e-4, atol=1e-3)
    monkeypatch.setenv('SPT_SPMM_MFMA', '1')
    # the same call twice: bitwise the same (ds_add_f32 in program and lane order)
    assert torch.equal(a, tiled(ext, dev(indptr), dev(idx), vals, x))


def test_ragged_rows_duplicates_empty_and_long_rows():
    """Row lengths 0 .. 150 with mean <= 64 (rows past 64 entries take the slow loop), random columns
    with duplicates, rows without entries (their y rows are zero), every row of one slice empty."""
    from naive_gpt import ext
    rng = np.random.default_rng(11)
    B, S, E = 6, 320, 64
    lens = rng.integers(0, 100, size=S)
    lens[[3, 77, 200]] = [150, 129, 65]
    lens[[0, 50, 319]] = 0
    assert lens.mean() <= 64 and lens.sum() * 16 >= S * S
    indptr = np.concatenate([[0], np.cumsum(lens)]).astype(np.int32)
    nnz = int(indptr[-1])
    idx = rng.integers(0, S, size=[B, nnz]).astype(np.int32)
    idx[:, indptr[3]:indptr[3] + 40] = 7                  # forty copies of one entry in a long row
    vals = rng.standard_normal([B, nnz]).astype(np.float32)
    vals[5] = 0.0
    x = rng.standard_normal([B, S, E]).astype(np.float32)
    want = O.spmm_forward(False, indptr, idx, vals, x)
    got = tiled(ext, dev(indptr), dev(idx), dev(vals), dev(x))
    close(got, want, atol=2e-4 * np.sqrt(150))
    assert not got[:, [0, 50, 319]].any() and not got[5].any()


@pytest.mark.parametrize('E', [64, 128])
def test_head_layouts(E):
    """x as [N, S, H, E] (the projections' layout) and y written the same way."""
    from naive_gpt import ext
    rng = np.random.default_rng(E)
    N, H, S, Z = 3, 4, 256, 32
    B = N * H
    indptr, idx = uniform_csr(rng, 2, S, Z, True)
    idx = np.ascontiguousarray(np.tile(idx, [B // 2, 1]))
    vals = rng.standard_normal([B, S * Z]).astype(np.float32)
    xh = rng.standard_normal([N, S, H, E]).astype(np.float32)
    x = np.ascontiguousarray(xh.transpose(0, 2, 1, 3)).reshape(B, S, E)
    want = O.spmm_forward(False, indptr, idx, vals, x)
    got = tiled(ext, dev(indptr), dev(idx), dev(vals), dev(xh), x_heads=H, y_heads=H)
    assert got.shape == (N, S, H, E)
    close(got.permute(0, 2, 1, 3).reshape(B, S, E), want, atol=2e-4 * np.sqrt(Z))


def test_shapes_the_form_leaves_to_the_gather_kernels(monkeypatch):
    from naive_gpt import ext
    lib = ext.load_library()
    assert lib.spt_spmm_form(0, 256, 512, 64, 512 * 64) == 1
    assert lib.spt_spmm_form(1, 256, 512, 64, 512 * 64) == 0      # transposed
    assert lib.spt_spmm_form(0, 256, 512, 32, 512 * 64) == 0      # d_head
    assert lib.spt_spmm_form(0, 256, 2048, 64, 2048 * 256) == 0   # rows of 256 entries
    assert lib.spt_spmm_form(0, 256, 1536, 64, 1536 * 48) == 0    # thinner than 1 / 16
    assert lib.spt_spmm_form(0, 256, 32, 64, 32 * 8) == 0
    monkeypatch.delenv('SPT_SPMM_MFMA')
    assert lib.spt_spmm_form(0, 256, 512, 64, 512 * 64) == 0      # not asked for: the default
