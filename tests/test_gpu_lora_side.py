"""spt_lora_down (lora_side.hip): the rank-r down product of the LoRA adapters
(reference: naive_gpt/layers/tuning/lora.py:70-80) against fp64 products of the same fp32 inputs."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(got, want, name):
    # split-bf16 products: <= 2^-16 relative per product; the bar is the library's 1e-3
    scale = want.abs().max().item()
    err = (got.double().cpu() - want).abs().max().item()
    assert err <= 2e-5 * scale, (name, err, scale)


@pytest.mark.parametrize('rows,K,n', [(8192, 1024, 16), (1000, 1024, 16), (16384, 1024, 64),
                                      (4096, 2048, 32), (520, 4096, 48), (16, 1024, 16), (5, 1024, 16),
                                      # K % 1024 != 0: the LLaMA-7B FFN block (2752 = 86 k-steps: 22, 22,
                                      # 21, 21 per wave), fewer steps than waves, one ragged chunk only
                                      (4096, 2752, 16), (1000, 2752, 32), (77, 32, 16), (300, 96, 16),
                                      (640, 160, 64), (2048, 11008, 16), (130, 1056, 48)])
def test_lora_down_is_x_times_table(rows, K, n):
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(rows + n)
    x = torch.randn([rows, K], generator=gen)
    table = torch.randn([K, n], generator=gen)
    want = x.double() @ table.double()
    u = ext.lora_down(x.cuda(), table.cuda())
    assert u.shape == (rows, n)
    _close(u, want, 'u')
    # exact = True: an fp32 FMA chain per element (the fp32-input MFMA), for the u in front of a ReLU
    # GEMM -- an order of magnitude inside the split products' error
    ue = ext.lora_down(x.cuda(), table.cuda(), exact=True)
    err = (ue.double().cpu() - want).abs().max().item()
    assert err <= 2e-6 * want.abs().max().item(), (err, want.abs().max().item())
    # by-products of the same pass: the split image (bit-equal to spt_split_bf16's) and row norms
    u2, image, norms = ext.lora_down(x.cuda(), table.cuda(), want_image=True, want_norms=True)
    assert torch.equal(u2, u)
    u3, image3, _ = ext.lora_down(x.cuda(), table.cuda(), want_image=True, want_norms=True, exact=True)
    assert torch.equal(u3, ue) and torch.equal(image3.buffer, image.buffer)
    assert torch.equal(image.buffer, ext.split_bf16(x.cuda()).buffer)
    assert torch.allclose(norms.cpu(), x.norm(dim=1), rtol=1e-5)


def test_lora_down_reads_a_strided_matrix():
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(3)
    wide = torch.randn([300, 2048 + 64], generator=gen).cuda()
    x = wide[:, 64:64 + 2048]                                   # row stride 2112, 256-byte offset
    table = torch.randn([2048, 16], generator=gen).cuda()
    _close(ext.lora_down(x, table), x.double().cpu() @ table.double().cpu(), 'u')


def test_unsupported_shapes_are_refused():
    from naive_gpt import ext
    x = torch.randn([64, 1000], device='cuda')                  # K % 32 != 0
    assert not ext.lora_down_supported(x, torch.randn([1000, 16], device='cuda'))
    assert ext.lora_down_supported(torch.randn([64, 2752], device='cuda'), torch.randn([2752, 16], device='cuda'))
    assert not ext.lora_down_supported(torch.randn([64, 1024], device='cuda'), torch.randn([1024, 8], device='cuda'))
    with pytest.raises(RuntimeError):
        ext.lora_down(x, torch.randn([1000, 16], device='cuda'))


@pytest.mark.parametrize('K', [1024, 2752])
@pytest.mark.parametrize('sizes', [[4100, 4000, 4200, 4084], [0, 1000, 0, 17], [16, 1, 15, 33, 0, 64]])
def test_grouped_down_product_per_block_tables(sizes, K):
    """spt_lora_down_grouped: rows sorted by block, each block its own [K, 16] table -- against the
    composition it replaces (all tables side by side, then the row's own block picked out) and an
    fp64 reference; the operand image is the one spt_split_bf16 writes."""
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(len(sizes))
    P, G, r = sum(sizes), len(sizes), 16
    x = torch.randn([P, K], generator=gen).cuda()
    tables = (torch.randn([G, K, r], generator=gen) * 0.05).cuda()
    offsets = torch.tensor([0] + sizes).cumsum(0).int().cuda()
    block = torch.repeat_interleave(torch.arange(G), torch.tensor(sizes)).cuda()
    u, image = ext.lora_down_grouped(x, tables, offsets, want_image=True)
    ref = torch.einsum('pk,pkr->pr', x.double(), tables.double()[block]).float()
    assert float((u - ref).abs().max()) <= 2e-5 * float(ref.abs().max())
    assert torch.equal(image.buffer, ext.split_bf16(x).buffer)
    assert torch.equal(ext.lora_down_grouped(x, tables, offsets), u)


def test_embedding_rows_backward_is_the_embedding_gradient_and_bit_reproducible():
    """lora._LookupRows (the adapter's left(x) of LoRAEmbedding, lora.py:118-126): same gradient as
    torch's embedding backward, identical bits over repeated runs (no atomics on distinct
    addends), with many repeated ids."""
    from naive_gpt.layers.tuning.lora import _LookupRows
    gen = torch.Generator().manual_seed(4)
    ids = torch.randint(0, 3000, [16, 512], generator=gen).cuda()
    ids[0, :64] = 7                                              # a long run
    table = torch.randn([30522, 16], generator=gen).cuda().requires_grad_(True)
    w = torch.randn([16, 512, 16], generator=gen).cuda()
    (torch.nn.functional.embedding(ids, table) * w).sum().backward()
    want, table.grad = table.grad.clone(), None
    runs = []
    for _ in range(3):
        (_LookupRows.apply(ids, table) * w).sum().backward()
        runs.append(table.grad.clone())
        table.grad = None
    assert torch.allclose(runs[0], want, rtol=1e-5, atol=1e-5)
    assert torch.equal(runs[0], runs[1]) and torch.equal(runs[0], runs[2])


def test_embedding_rows_backward_long_runs_and_a_non_finite_row():
    """Runs longer than a chunk (64 positions) and longer than several (one id 5000 times: a padding
    token), ids at both ends of the table, against an fp64 scatter-add; and ONE non-finite gradient
    row stays in its own id's row (a single running sum carried it into every later
    id)."""
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(5)
    T, rows, width = 8192, 1000, 16
    ids = torch.randint(0, rows, [T], generator=gen)
    ids[100:5100] = 3                                          # 5000 copies: 79 chunks
    ids[6000:6070] = rows - 1
    ids[7000] = 0
    ids = ids[torch.randperm(T, generator=gen)].cuda()
    g = torch.randn([T, width], generator=gen).cuda()
    want = torch.zeros([rows, width], dtype=torch.float64, device='cuda').index_add_(0, ids, g.double())
    got = ext.embedding_rows_backward(g, ids, rows)
    assert torch.allclose(got.double(), want, rtol=1e-6, atol=1e-6)
    absent = torch.ones(rows, dtype=torch.bool, device='cuda')
    absent[ids] = False
    assert absent.any() and float(got[absent].abs().max()) == 0.0
    bad = int((ids == 5).nonzero()[0])                         # some position of id 5
    g[bad, 2] = float('inf')
    g[bad, 3] = float('nan')
    got2 = ext.embedding_rows_backward(g, ids, rows)
    others = torch.arange(rows, device='cuda') != 5
    assert torch.isfinite(got2[others]).all()
    assert torch.equal(got2[others], got[others])
    assert torch.isinf(got2[5, 2]) and torch.isnan(got2[5, 3])


def test_second_table_block_and_routing_from_logits():
    """spt_lora_down2: x @ table2.T for a row-major [n2, K] matrix as one more block of the x @ L pass
    (exact fp32), and spt_route_topk_logits: the routing of sigmoid(logits + bias) from that block --
    against the library's Linear + Sigmoid and spt_route_topk_coeff on its probabilities."""
    from naive_gpt import ext
    torch.manual_seed(0)
    T, K, G, k = 3000, 256, 4, 2
    x = torch.randn([T, K], device='cuda')
    table = torch.randn([K, 16], device='cuda')
    rw = torch.randn([G, K], device='cuda') / 16
    rb = torch.randn([G], device='cuda')
    pack, image, norms = ext.lora_down(x, table, want_image=True, want_norms=True, block_major=True,
                                       exact=True, table2=rw)
    assert pack.shape == (2, T, 16)
    u_ref, image_ref, norms_ref = ext.lora_down(x, table, want_image=True, want_norms=True, exact=True)
    assert torch.equal(pack[0], u_ref) and torch.equal(image.buffer, image_ref.buffer)
    assert torch.equal(norms, norms_ref)
    logits = x.double() @ rw.double().T
    assert torch.allclose(pack[1][:, :G].double(), logits, rtol=0, atol=1e-5 * float(logits.abs().max()))
    assert float(pack[1][:, G:].abs().max()) == 0.0
    # without by-products, and a table of 32 columns in front
    wide = torch.randn([K, 32], device='cuda')
    pack3 = ext.lora_down(x, wide, block_major=True, exact=True, table2=rw)
    assert pack3.shape == (3, T, 16)
    assert torch.allclose(pack3[2][:, :G].double(), logits, rtol=0, atol=1e-5 * float(logits.abs().max()))
    with pytest.raises(Exception):          # the second table exists in the exact form only
        ext.lora_down(x, wide, block_major=True, table2=rw)
    # routing from the logits block == routing of the probabilities torch forms from the same logits
    prob, token, block, offsets, pos, token64, block64, coeff = ext.route_topk_logits(pack[1], rb, G, k, 2.0)
    prob_ref = torch.sigmoid(pack[1][:, :G] + rb)
    assert torch.allclose(prob, prob_ref, rtol=2e-6, atol=1e-7)
    want = ext.route_topk_coeff(prob.contiguous(), k, 2.0)
    for a, b in zip((token, block, offsets, pos, token64, block64, coeff), want):
        assert torch.equal(a, b)


def test_rows_combine_with_a_side_product():
    from naive_gpt import ext
    torch.manual_seed(1)
    T, k, d, ns = 1000, 2, 256, 4
    rows = torch.randn([T * k, d], device='cuda')
    pos = torch.randperm(T * k, device='cuda').view(T, k).int()
    side = torch.randn([T, ns], device='cuda')
    w = torch.randn([ns, d], device='cuda')
    got = ext.rows_combine(rows, pos, side=side, side_weight=w)
    want = rows[pos.long()].double().sum(1) + side.double() @ w.double()
    assert torch.allclose(got.double(), want, rtol=1e-5, atol=1e-5)
