"""CPU tier: the oracle against independent statements of the same maths --
the reference's own test formulas (test/kernel/*.py) restated with dense torch /
numpy, and an independently written closed form of the lookup selector."""
import numpy as np
import pytest
import torch

from oracle import oracle as O
from tests.helpers import (dense_from_csr, lookup_closed_form, ragged_csr,
                           uniform_csr)


@pytest.mark.parametrize('M,NQ,C,D', [(8, 256, 16, 8), (3, 64, 48, 4), (1, 128, 256, 8),
                                      (16, 64, 16, 16)])
def test_cdist_matches_torch_cdist(M, NQ, C, D):
    # reference formula: test/kernel/test_cdist.py:24-52
    g = torch.Generator().manual_seed(M * 1000 + NQ + C + D)
    query = torch.randn([M, NQ, D], generator=g, requires_grad=True)
    table = torch.randn([M, C, D], generator=g, requires_grad=True)
    dist_t = torch.cdist(query, table, p=1.0)
    idx_t = torch.argmin(dist_t, dim=-1)
    dist_o, idx_o = O.cdist_forward(query.detach().numpy(), table.detach().numpy())
    assert np.allclose(dist_t.detach().numpy(), dist_o, atol=1e-5)
    assert np.array_equal(idx_t.numpy(), idx_o)
    # gradient of the gathered min, as the reference test does
    torch.gather(dist_t, -1, idx_t.unsqueeze(-1)).sum().backward()
    go = np.zeros([M, NQ, C], np.float32)
    np.put_along_axis(go, idx_o[..., None].astype(np.int64), 1.0, axis=-1)
    gq, gt = O.cdist_backward(query.detach().numpy(), table.detach().numpy(), go)
    assert np.allclose(query.grad.numpy(), gq, atol=1e-4)
    assert np.allclose(table.grad.numpy(), gt, atol=1e-3)


def test_cdist_tie_breaks_to_lowest_index_and_nan_rule():
    query = np.zeros([1, 16, 4], np.float32)
    table = np.ones([1, 16, 4], np.float32)          # every codeword equidistant
    _, idx = O.cdist_forward(query, table)
    assert (idx == 0).all()                           # strict '<', cdist.cu:52-54
    table[0, 5] = 0.5
    _, idx = O.cdist_forward(query, table)
    assert (idx == 5).all()
    query[0, 3] = 1e14                                # all distances >= 1e13 start value
    _, idx = O.cdist_forward(query, table)
    assert idx[0, 3] == 0                             # cdist.cu:28-29


@pytest.mark.parametrize('S,M,hi,coeff', [(256, 8, 8, 8), (128, 10, 4, 8), (256, 16, 2, 8),
                                          (512, 8, 2, 8), (128, 8, 3, 4), (64, 16, 16, 4),
                                          (64, 5, 2, 4)])
def test_lookup_literal_emulation_equals_closed_form(S, M, hi, coeff):
    rng = np.random.default_rng(S + M + hi)
    q = rng.integers(0, hi, [2, S, M], dtype=np.int32)
    k = rng.integers(0, hi, [2, S, M], dtype=np.int32)
    assert np.array_equal(O.lookup_forward(q, k, coeff), lookup_closed_form(q, k, coeff))


def test_lookup_compares_uint16_truncations():
    rng = np.random.default_rng(7)
    q = rng.integers(0, 4, [1, 64, 8], dtype=np.int32)
    k = rng.integers(0, 4, [1, 64, 8], dtype=np.int32)
    base = O.lookup_forward(q, k, 4)
    assert np.array_equal(O.lookup_forward(q + 65536, k, 4), base)   # lookup.cu:22,43
    assert np.array_equal(O.lookup_forward(q, k - 65536 * 3, 4), base)


def test_lookup_recall_property_of_the_reference_test():
    # test/kernel/test_lookup.py:36-75: codes uniform in [0,8), M=8, recall > 0.8
    rng = np.random.default_rng(0)
    B, S, M = 2, 512, 8
    q = rng.integers(0, 8, [B, S, M], dtype=np.int32)
    k = rng.integers(0, 8, [B, S, M], dtype=np.int32)
    out = O.lookup_forward(q, k, 8)
    recall = []
    for b in range(B):
        score = (q[b][:, None, :] == k[b][None, :, :]).sum(-1)
        for row in range(0, S, 7):
            kk = min(row + 1, S // 8)
            s = score[row, :row + 1]
            gt = set(np.argsort(-s, kind='stable')[:kk].tolist())
            thresh = np.sort(s)[::-1][kk - 1]
            pred = out[b, row, :kk].tolist()
            # ties at the threshold are interchangeable members of the top-k
            hit = sum(1 for c in set(pred) if s[c] > thresh or c in gt or s[c] == thresh)
            recall.append(min(hit, kk) / kk)
    assert np.mean(recall) > 0.8


def test_lookup_rejects_reference_shape_violations():
    q = np.zeros([1, 24, 8], np.int32)
    with pytest.raises(RuntimeError):
        O.lookup_forward(q, q, 8)          # S % 16 != 0, lookup.cu:103
    q = np.zeros([1, 64, 8], np.int32)
    with pytest.raises(RuntimeError):
        O.lookup_forward(q, q, 8)          # Z = 8, Z % 16 != 0, lookup.cu:106


@pytest.mark.parametrize('causal', [False, True])
def test_sddmm_spmm_match_dense(causal):
    # test/kernel/test_sddmm.py:57-85, test_spmm.py:55-82
    rng = np.random.default_rng(3)
    B, S, E, Z = 3, 64, 32, 8
    indptr, idx = uniform_csr(rng, B, S, Z, causal)
    q = rng.standard_normal([B, S, E]).astype(np.float32)
    k = rng.standard_normal([B, S, E]).astype(np.float32)
    out = O.sddmm_forward(indptr, idx, q, k)
    full = np.einsum('bie,bje->bij', q.astype(np.float64), k.astype(np.float64))
    rows = np.repeat(np.arange(S), Z)
    for b in range(B):
        assert np.allclose(out[b], full[b, rows, idx[b]], atol=1e-5)
    vals = rng.standard_normal([B, S * Z]).astype(np.float32)
    dense = dense_from_csr(indptr, idx, vals, S)
    y = O.spmm_forward(False, indptr, idx, vals, k)
    assert np.allclose(y, dense @ k, atol=1e-4)
    yt = O.spmm_forward(True, indptr, idx, vals, k)
    assert np.allclose(yt, dense.transpose(0, 2, 1) @ k, atol=1e-4)


def test_spmm_ragged_and_duplicates():
    rng = np.random.default_rng(5)
    B, S, E = 2, 48, 16
    indptr, idx = ragged_csr(rng, B, S, 9)
    vals = rng.standard_normal(idx.shape).astype(np.float32)
    x = rng.standard_normal([B, S, E]).astype(np.float32)
    dense = dense_from_csr(indptr, idx, vals, S)       # duplicates accumulate
    assert np.allclose(O.spmm_forward(False, indptr, idx, vals, x), dense @ x, atol=1e-4)
    assert np.allclose(O.spmm_forward(True, indptr, idx, vals, x),
                       dense.transpose(0, 2, 1) @ x, atol=1e-4)


def test_softmax_matches_dense_masked_softmax():
    # test/kernel/test_softmax.py:62-95 (distinct causal columns per row)
    rng = np.random.default_rng(11)
    B, S, Z = 2, 64, 8
    indptr, idx = uniform_csr(rng, B, S, Z, causal=True)
    vals = rng.random([B, S * Z]).astype(np.float32)
    y = O.softmax_forward(indptr, idx, vals)
    rows = np.repeat(np.arange(S), Z)
    dy = rng.standard_normal([B, S * Z]).astype(np.float32)
    dv = O.softmax_backward(indptr, idx, y, dy)
    for b in range(B):
        v = torch.full([S, S], float('-inf'), dtype=torch.float64)
        seen = torch.zeros([S, S], dtype=torch.bool)
        leaf = torch.tensor(vals[b].astype(np.float64), requires_grad=True)
        # rows shorter than Z carry duplicate column-0 padding: keep the first only
        keep = []
        for p, (r, c) in enumerate(zip(rows, idx[b])):
            if c <= r and not seen[r, c]:
                seen[r, c] = True
                keep.append(p)
        keep = np.array(keep)
        dense = v.index_put((torch.tensor(rows[keep]), torch.tensor(idx[b][keep])),
                            leaf[torch.tensor(keep)])
        sm = torch.softmax(dense, dim=-1)
        full_rows = np.array([r for r in range(S) if r + 1 >= Z])
        sel = np.isin(rows, full_rows)
        got = y[b][sel]
        want = sm[torch.tensor(rows[sel]), torch.tensor(idx[b][sel])].detach().numpy()
        assert np.allclose(got, want, atol=1e-5)
        # backward on rows whose sum y*dy is positive (the reference clamps it, softmax.cu:69)
        ysum = (y[b] * dy[b]).reshape(S, Z).sum(-1)
        okrow = np.array([r for r in full_rows if ysum[r] > 1e-6])
        selb = np.isin(rows, okrow)
        (sm[torch.tensor(rows[selb]), torch.tensor(idx[b][selb])]
         * torch.tensor(dy[b][selb].astype(np.float64))).sum().backward()
        assert np.allclose(dv[b][selb], leaf.grad.numpy()[selb], atol=1e-5)


def test_softmax_quirks_are_kept():
    indptr = np.array([0, 4, 8], np.int32)
    idx = np.array([[0, 1, 1, 0, 0, 1, 1, 5]], np.int32)      # col 1 > row 0 -> masked; 5 > 1
    vals = np.zeros([1, 8], np.float32)
    y = O.softmax_forward(indptr, idx, vals)
    assert np.allclose(y[0, :4], [0.5, 0, 0, 0.5])            # duplicates each get a share
    assert np.allclose(y[0, 4:], [1 / 3, 1 / 3, 1 / 3, 0])
    dy = -np.ones([1, 8], np.float32)                          # sum y*dy < 0 -> clamped to 1e-9
    dv = O.softmax_backward(indptr, idx, y, dy)
    assert np.allclose(dv[0, 0], 0.5 * (-1 - 1e-9))
    # fully masked row: denominator clamps to 1e-9, outputs are 0 (softmax.cu:30)
    idx2 = np.array([[3, 3, 3, 3, 0, 1, 1, 5]], np.int32)
    assert (O.softmax_forward(indptr, idx2, vals)[0, :4] == 0).all()
