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
                                      (4096, 2048, 32), (520, 4096, 48), (16, 1024, 16), (5, 1024, 16)])
def test_lora_down_is_x_times_table(rows, K, n):
    from naive_gpt import ext
    gen = torch.Generator().manual_seed(rows + n)
    x = torch.randn([rows, K], generator=gen)
    table = torch.randn([K, n], generator=gen)
    want = x.double() @ table.double()
    u = ext.lora_down(x.cuda(), table.cuda())
    assert u.shape == (rows, n)
    _close(u, want, 'u')
    # by-products of the same pass: the split image (bit-equal to spt_split_bf16's) and row norms
    u2, image, norms = ext.lora_down(x.cuda(), table.cuda(), want_image=True, want_norms=True)
    assert torch.equal(u2, u)
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
    x = torch.randn([64, 1000], device='cuda')
    assert not ext.lora_down_supported(x, torch.randn([1000, 16], device='cuda'))
    assert not ext.lora_down_supported(torch.randn([64, 1024], device='cuda'), torch.randn([1024, 8], device='cuda'))
    with pytest.raises(RuntimeError):
        ext.lora_down(x, torch.randn([1000, 16], device='cuda'))
