"""spt_sparse_attention_forward (fused_attention.hip) against the oracle chain
sddmm -> scale, clamp -> softmax -> spmm, and the layer through it against the layer through
the separate operators (values and every gradient)."""
import numpy as np
import pytest
import torch

from oracle import ext_stub

pytestmark = pytest.mark.gpu
CLAMP = 10.0


def oracle_chain(indices, q, k, v, scale):
    """q, k, v [N, S, H, E] on the CPU -> scores, attn [B, nnz], y [B, S, E]."""
    N, S, H, E = q.shape
    B, Z = N * H, indices.size(-1) // S
    flat = lambda t: t.transpose(1, 2).contiguous().view(B, S, E)       # noqa: E731
    indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32)
    f, t = torch.scalar_tensor(False), torch.scalar_tensor(True)
    raw = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, flat(q), flat(k))
    scores = (raw * np.float32(scale)).clamp(-CLAMP, CLAMP)
    attn = ext_stub.softmax_forward_cuda(indptr, indices, scores)
    y = ext_stub.spmm_forward_cuda(f, f, indptr, indices, attn, flat(v))
    return scores, attn, y


def causal_indices(B, S, Z, gen):
    """Uniform rows with the structure lookup produces: columns <= row, zero padding."""
    idx = torch.zeros([B, S, Z], dtype=torch.int32)
    for r in range(S):
        n = min(r + 1, Z)
        idx[:, r, :n] = torch.randint(0, r + 1, [B, n], generator=gen, dtype=torch.int32)
    return idx.view(B, S * Z)


@pytest.mark.parametrize('causal', [False, True])
@pytest.mark.parametrize('N,H,S,Z,yt', [
    (2, 16, 512, 64, False), (2, 16, 512, 64, True),
    (1, 32, 256, 32, True),           # Z = 32: half of the entry lanes idle
    (4, 8, 128, 16, True),            # one 16-entry group per row
    (2, 16, 80, 8, False),            # S not a multiple of 64 (ragged last write-out phase)
    (2, 16, 80, 8, True),
])
def test_fused_forward_matches_oracle_chain(N, H, S, Z, yt, causal):
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(N * S + Z)
    q, k, v = [torch.randn([N, S, H, 64], generator=gen) for _ in range(3)]
    q = q * 3.0                                    # some scores beyond the clamp
    indices = causal_indices(N * H, S, Z, gen)
    scale = 64 ** -0.5
    want_scores, want_attn, want_y = oracle_chain(indices, q, k, v, scale)
    scores, attn, y = ext.sparse_attention_forward(indices.cuda(), q.cuda(), k.cuda(), v.cuda(),
                                                   scale, CLAMP, y_transposed=yt, causal=causal)
    assert (want_scores.abs() == CLAMP).any()
    assert torch.allclose(scores.cpu(), want_scores, rtol=1e-4, atol=1e-4)
    assert torch.allclose(attn.cpu(), want_attn, rtol=1e-3, atol=1e-6)
    if yt:
        assert y.shape == (N * H, 64, S)
        y = y.transpose(1, 2)
    assert torch.allclose(y.cpu(), want_y, rtol=1e-3, atol=1e-4)
    # rows are probability vectors over the unmasked entries
    assert torch.allclose(attn.view(N * H, S, Z).sum(-1).cpu(), torch.ones(N * H, S), atol=1e-5)


def test_masked_entries_get_zero_probability():
    """Non-causal columns (col > row) take no part, as in extension/softmax.cu:19-31."""
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(4)
    N, H, S, Z = 2, 16, 128, 16
    q, k, v = [torch.randn([N, S, H, 64], generator=gen) for _ in range(3)]
    indices = torch.randint(0, S, [N * H, S * Z], generator=gen, dtype=torch.int32)
    indices.view(N * H, S, Z)[:, :, 0] = 0         # every row keeps one live entry
    want_scores, want_attn, want_y = oracle_chain(indices, q, k, v, 0.125)
    # causal=False: K and V are staged whole, so even masked entries get their true score
    scores, attn, y = ext.sparse_attention_forward(indices.cuda(), q.cuda(), k.cuda(), v.cuda(),
                                                   0.125, CLAMP)
    rows = torch.arange(S).view(1, S, 1)
    dead = indices.view(N * H, S, Z) > rows
    assert dead.any() and (attn.cpu().view(N * H, S, Z)[dead] == 0).all()
    assert torch.allclose(attn.cpu(), want_attn, rtol=1e-3, atol=1e-6)
    assert torch.allclose(y.cpu(), want_y, rtol=1e-3, atol=1e-4)


def test_unsupported_shapes_are_refused():
    from naive_gpt import ext
    q = torch.randn([2, 128, 16, 32], device='cuda')          # E = 32
    idx = torch.zeros([32, 128 * 16], dtype=torch.int32, device='cuda')
    assert not ext.fused_attention_supported(128, 32, 32, 128 * 16)
    with pytest.raises(RuntimeError):
        ext.sparse_attention_forward(idx, q, q, q, 1.0, CLAMP)


@pytest.mark.parametrize('kind', ['vanilla', 'rotary'])
def test_layer_through_fused_kernel_equals_separate_operators(kind, monkeypatch):
    from naive_gpt import ext, layers
    torch.manual_seed(0)
    if kind == 'vanilla':
        attn = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0)
    else:
        attn = layers.SparseRotaryAttentionV2(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0)
    attn = attn.cuda()
    N, S, H = 2, 256, 16
    q, k, v, w = [torch.randn([N, S, H, 64], device='cuda') for _ in range(4)]

    def run(fused):
        if not fused:
            monkeypatch.setattr(ext, 'fused_attention_supported', lambda *a: False)
        qi, ki, vi = [t.clone().requires_grad_(True) for t in (q, k, v)]
        y = attn(qi, ki, vi, attn_mask=None)
        (y * w).sum().backward()
        monkeypatch.undo()
        return y.detach(), qi.grad, ki.grad, vi.grad

    fused, separate = run(True), run(False)
    for a, b, name in zip(fused, separate, ['y', 'grad_q', 'grad_k', 'grad_v']):
        assert torch.allclose(a, b, rtol=1e-3, atol=1e-4), name


@pytest.mark.parametrize('causal', [False, True])
@pytest.mark.parametrize('N,H,S,Z,gt', [
    (2, 16, 512, 64, True), (2, 16, 512, 64, False),
    (1, 32, 256, 32, True), (2, 16, 80, 8, True), (2, 16, 80, 8, False),
])
def test_backward_rows_matches_oracle_chain(N, H, S, Z, gt, causal):
    """dP = sddmm(dY, V); dS = clamp-mask(scale * softmax_backward(P, dP)); dQ = spmm(dS, K)."""
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(7 * N + S + Z)
    B, E = N * H, 64
    q, k, v = [torch.randn([N, S, H, E], generator=gen) for _ in range(3)]
    q = q * 3.0
    gy = torch.randn([B, S, E], generator=gen)
    indices = causal_indices(B, S, Z, gen)
    scale = E ** -0.5
    scores, attn, _ = oracle_chain(indices, q, k, v, scale)
    flat = lambda t: t.transpose(1, 2).contiguous().view(B, S, E)       # noqa: E731
    indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32)
    f, t = torch.scalar_tensor(False), torch.scalar_tensor(True)
    dp = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, gy, flat(v))
    ds = ext_stub.softmax_backward_cuda(indptr, indices, attn, dp)
    want_raw = torch.where(scores.abs() < CLAMP, ds * np.float32(scale), torch.zeros_like(ds))
    want_q = ext_stub.spmm_forward_cuda(f, f, indptr, indices, want_raw, flat(k))
    want_q = want_q.view(N, H, S, E).transpose(1, 2)                      # -> [N, S, H, E]

    gy_in = gy.transpose(1, 2).contiguous() if gt else gy
    raw, gq, rows = ext.sparse_attention_backward_rows(
        indices.cuda(), gy_in.cuda(), v.cuda(), k.cuda(), scores.cuda(), attn.cuda(),
        scale, CLAMP, grad_y_transposed=gt, causal=causal)
    assert (scores.abs() == CLAMP).any() and (want_raw == 0).any()
    assert torch.allclose(raw.cpu(), want_raw, rtol=1e-3, atol=1e-5)
    assert torch.allclose(gq.cpu(), want_q, rtol=1e-3, atol=1e-4)
    assert torch.equal(rows.cpu(), gy)                                    # a pure re-layout
