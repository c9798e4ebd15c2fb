"""Edge shapes of the kernels added in round 3 (rotary, SwiGLU, the second table of spt_lora_down,
routing from logits, the un-bucketing pass with a side product): odd sequence lengths and head counts,
a single row, K = 2752 (ragged k-chunks), row counts that are not multiples of the tile."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


def test_edge_shapes_of_the_round_3_kernels():
    from naive_gpt import ext, layers
    torch.manual_seed(0)
    # rotary: E = 64, odd S and heads, 1 .. 3 parts, n_rot 0 .. 2
    for (N, S, H, E) in [(1, 7, 3, 64), (2, 130, 5, 128), (1, 1, 1, 8)]:
        emb = layers.RotaryEmbedding(n_embeddings=256, d_model=E).cuda()
        xs = [torch.randn([N, S, H, E], device='cuda') for _ in range(3)]
        ids = torch.arange(S, device='cuda')
        for n_parts in (1, 2, 3):
            for n_rot in range(0, min(2, n_parts) + 1):
                out = ext.rotary(xs[:n_parts], n_rot, emb.cos_cached, emb.sin_cached)
                for p in range(n_parts):
                    want = emb(xs[p], ids) if p < n_rot else xs[p]
                    assert torch.allclose(out[p], want, rtol=1e-6, atol=1e-6), (N, S, H, E, n_parts, n_rot, p)
    # swiglu: tiny rows, n = 4
    g, s, dh = [torch.randn([3, 4], device='cuda') for _ in range(3)]
    h = ext.swiglu_forward(g, s)
    assert torch.allclose(h, nn.functional.silu(g) * s, rtol=1e-6, atol=1e-6)
    dg, ds, d1, d2, d3 = ext.swiglu_backward(dh, g, s)
    g64 = g.double().requires_grad_(True); s64 = s.double().requires_grad_(True)
    a, b = torch.autograd.grad(nn.functional.silu(g64) * s64, (g64, s64), dh.double())
    assert torch.allclose(dg.double(), a, atol=1e-6) and torch.allclose(ds.double(), b, atol=1e-6)
    # lora_down2 at K = 2752 (ragged chunks) and K = 32, rows not a multiple of 16
    for rows, K in [(1000, 2752), (17, 32), (4099, 1024)]:
        x = torch.randn([rows, K], device='cuda'); t = torch.randn([K, 16], device='cuda'); rw = torch.randn([5, K], device='cuda')
        pack = ext.lora_down(x, t, block_major=True, exact=True, table2=rw)
        assert torch.allclose(pack[0].double(), x.double() @ t.double(), rtol=0, atol=1e-4 * float((x.double() @ t.double()).abs().max()))
        ref = x.double() @ rw.double().T
        assert torch.allclose(pack[1][:, :5].double(), ref, rtol=0, atol=1e-5 * float(ref.abs().max())), (rows, K)
        assert float(pack[1][:, 5:].abs().max()) == 0
    # route_topk_logits: T not a multiple of 256, G = 8, k = 4; ld = 16
    T, G, k = 1000, 8, 4
    logits = torch.randn([T, 16], device='cuda'); bias = torch.randn([G], device='cuda')
    prob, token, block, offsets, pos, t64, b64, coeff = ext.route_topk_logits(logits, bias, G, k, 2.0)
    pr = torch.sigmoid(logits[:, :G] + bias)
    want = ext.route_topk_coeff(pr.contiguous(), k, 2.0)
    assert torch.allclose(prob, pr, rtol=2e-6, atol=1e-7)
    assert all(torch.equal(a, b) for a, b in zip((token, block, offsets, pos, t64, b64), want[:6]))
    # rows_combine side with k = 1 and bias
    rows = torch.randn([50, 8], device='cuda'); pos = torch.randperm(50, device='cuda').view(50, 1).int()
    side = torch.randn([50, 3], device='cuda'); w = torch.randn([3, 8], device='cuda'); bias = torch.randn([8], device='cuda')
    got = ext.rows_combine(rows, pos, bias=bias, side=side, side_weight=w)
    assert torch.allclose(got, rows[pos.long().view(-1)] + bias + side @ w, rtol=1e-5, atol=1e-5)
