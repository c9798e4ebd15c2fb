"""spt_tall_tn (lora_side.hip): the LoRA table gradients wide^T . narrow, plain and grouped,
against the torch composition they replace (fp32; tolerance 1e-5 of the result's scale: both
sides are fp32 sums of the same exact-ish products in different orders)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _close(a, b, tol=2e-5):
    scale = max(float(b.abs().max()), 1e-20)
    assert float((a - b).abs().max()) <= tol * scale, float((a - b).abs().max()) / scale


@pytest.mark.parametrize('rows,width,n', [(8192, 1024, 16), (8192, 1024, 48), (8192, 1024, 4),
                                          (1000, 514, 16), (70, 64, 4), (16384, 4096, 16), (513, 2, 48)])
def test_plain_product_and_its_transposed_form(rows, width, n):
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(rows + n)
    wide = torch.randn([rows, width], generator=gen).cuda()
    narrow = torch.randn([rows, n], generator=gen).cuda()
    ref = torch.matmul(wide.double().t(), narrow.double()).float()
    out = ext.tall_tn(wide, narrow)
    assert out.shape == (1, width, n)
    _close(out[0], ref)
    _close(ext.tall_tn(wide, narrow, transposed=True)[0], ref.t())
    if n % 16 == 0:      # one contiguous [width, 16] matrix per rank-16 table
        split = ext.tall_tn(wide, narrow, split16=True)
        assert split.shape == (1, n // 16, width, 16)
        assert torch.equal(split[0].permute(1, 0, 2).reshape(width, n), out[0])
    # deterministic: the same bits twice
    assert torch.equal(out, ext.tall_tn(wide, narrow))


def test_strided_operands_and_the_layer_level_helper():
    from naive_gpt import ext
    from naive_gpt.layers.tuning import lora
    gen = torch.Generator().manual_seed(5)
    big = torch.randn([4096, 1100], generator=gen).cuda()
    cat = torch.randn([4096, 48], generator=gen).cuda()
    wide, narrow = big[:, 4:1028], cat[:, 16:32]                     # row strides 1100 / 48
    _close(ext.tall_tn(wide, narrow)[0], torch.matmul(wide.double().t(), narrow.double()).float())
    a, b = wide.contiguous(), narrow.contiguous()
    _close(lora.tall_tn(a, b), torch.matmul(a.double().t(), b.double()).float())
    _close(lora.tall_tn(b, a), torch.matmul(b.double().t(), a.double()).float())     # [16, 1024]
    odd = torch.randn([4096, 24], generator=gen).cuda()               # a rank the kernel does not take
    _close(lora.tall_tn(a, odd), torch.matmul(a.double().t(), odd.double()).float())


@pytest.mark.parametrize('sizes', [[4100, 4000, 4200, 4084], [0, 16384, 0, 0], [1, 63, 64, 65, 129, 0, 7, 5000]])
def test_grouped_product_with_a_gather(sizes):
    """Rows sorted by block, narrow picked per row through `gather`: block g's table gradient is
    wide[rows of g]^T . per_token[gather[rows of g]] -- against the scatter-into-zeros composition
    of layers/sparse/grouped.py (_in_own_block + one product + _block_major)."""
    from naive_gpt import ext
    from naive_gpt.layers.sparse import grouped
    gen = torch.Generator().manual_seed(len(sizes))
    P, G, bs, r, T = sum(sizes), len(sizes), 256, 16, 3000
    wide = torch.randn([P, bs], generator=gen).cuda()
    per_token = torch.randn([T, r], generator=gen).cuda()
    token = torch.randint(0, T, [P], generator=gen).int().cuda()
    offsets = torch.tensor([0] + sizes).cumsum(0).int().cuda()
    block = torch.repeat_interleave(torch.arange(G), torch.tensor(sizes)).cuda()
    out = ext.tall_tn(wide, per_token, gather=token, offsets=offsets)
    assert out.shape == (G, bs, r)
    rows = per_token.index_select(0, token.long())
    ref = grouped._block_major(
        torch.matmul(wide.double().t(), grouped._in_own_block(rows, block, G).double()).float(), G)
    _close(out.view(G * bs, r), ref)
    for g, size in enumerate(sizes):
        if size == 0:
            assert float(out[g].abs().max()) == 0.0


def test_bad_arguments_are_refused():
    from naive_gpt import ext
    wide = torch.randn([1024, 64]).cuda()
    with pytest.raises(Exception):
        ext.tall_tn(wide, torch.randn([1024, 24]).cuda())             # rank not compiled in
    with pytest.raises(Exception):
        ext.tall_tn(wide, torch.randn([1000, 16]).cuda())             # rows disagree
    with pytest.raises(Exception):
        ext.tall_tn(wide[:, :63], torch.randn([1024, 16]).cuda())      # odd width


@pytest.mark.parametrize('count', [2, 3, 4])
def test_batch_of_products_is_bit_equal_to_the_single_launches(count):
    """spt_tall_tn_batch: the same kernels with the problem in blockIdx.z -- every product bit-equal
    to its own spt_tall_tn call (plain and grouped with a shared gather)."""
    from naive_gpt import ext
    torch.manual_seed(count)
    rows, width, n = 4000, 384, 16
    wides = [torch.randn([rows, width], device='cuda') for _ in range(count)]
    block = torch.randn([count, rows, n], device='cuda')           # narrow sides: slices of one buffer
    narrows = [block[i] for i in range(count)]
    assert ext.tall_tn_batchable(wides, narrows)
    for got, w, nr in zip(ext.tall_tn_batch(wides, narrows), wides, narrows):
        assert torch.equal(got, ext.tall_tn(w, nr))
    # grouped: rows sorted by block, per-token narrow sides picked by a shared gather
    T = 1000
    token = torch.randint(0, T, [rows], device='cuda', dtype=torch.int32)
    offsets = torch.tensor([0, 900, 2100, 2100, rows], dtype=torch.int32, device='cuda')
    per_token = [torch.randn([T, n], device='cuda') for _ in range(count)]
    for got, w, nr in zip(ext.tall_tn_batch(wides, per_token, gather=token, offsets=offsets), wides, per_token):
        assert torch.equal(got, ext.tall_tn(w, nr, gather=token, offsets=offsets))
    assert not ext.tall_tn_batchable(wides, narrows[:-1] + [torch.randn([rows, 4], device='cuda')])
    assert not ext.tall_tn_batchable(wides * 3, narrows * 3)          # at most four


def test_lora_table_gradients_in_pairs():
    from naive_gpt import ext
    from naive_gpt.layers.tuning import lora
    torch.manual_seed(0)
    x, dy = torch.randn([2048, 256], device='cuda'), torch.randn([2048, 256], device='cuda')
    du, u = torch.randn([2048, 16], device='cuda'), torch.randn([2048, 16], device='cuda')
    ext.reset_paths()
    gl, none, gr = lora.tall_tn_many([(x, du), None, (dy, u)])
    assert none is None and ext.paths_taken('tall_tn') == {('tall_tn', 'kernel_batch'): 1}
    assert torch.equal(gl, lora.tall_tn(x, du)) and torch.equal(gr, lora.tall_tn(dy, u))
    # different shapes: one by one
    a, b = lora.tall_tn_many([(x, du), (dy[:, :128].contiguous(), u)])
    assert torch.equal(a, gl) and torch.equal(b, lora.tall_tn(dy[:, :128].contiguous(), u))
