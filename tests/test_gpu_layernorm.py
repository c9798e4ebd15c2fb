"""spt_add_layernorm_forward / spt_layernorm_backward (csrc/layernorm.hip) and the pair-wise
pre-norm wiring built on them (layers/basic/fused_norm.py) against nn.LayerNorm and the reference's
wiring x + f(norm(x)) (naive_gpt/layers/basic/transformer.py:46-52)."""
import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('rows,d', [(8192, 1024), (1000, 1024), (77, 2048), (513, 512), (9, 256)])
@pytest.mark.parametrize('add', [False, True])
def test_add_layernorm_forward_backward(rows, d, add):
    from naive_gpt.layers.basic import fused_norm
    gen = torch.Generator().manual_seed(rows + d)
    norm = nn.LayerNorm(d).cuda()
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.1 * torch.randn([d], generator=gen))
        norm.bias.copy_(0.1 * torch.randn([d], generator=gen))
    x = (2.0 * torch.randn([rows, d], generator=gen) + 0.5).cuda()
    r = torch.randn([rows, d], generator=gen).cuda() if add else None
    ws, wy = [torch.randn([rows, d], generator=gen).cuda() for _ in range(2)]

    def run(fused):
        xi = x.clone().requires_grad_(True)
        ri = r.clone().requires_grad_(True) if add else None
        norm.zero_grad()
        if fused:
            assert fused_norm.usable(norm, xi)
            s, y = fused_norm.add_norm(norm, xi, ri)
        else:
            s = xi + ri if add else xi
            y = norm(s)
        ((s * ws).sum() + (y * wy).sum()).backward()        # both outputs are used downstream
        return (s.detach(), y.detach(), xi.grad, ri.grad if add else None,
                norm.weight.grad.clone(), norm.bias.grad.clone())

    got, want = run(True), run(False)
    names = ['s', 'y', 'grad_x', 'grad_r', 'grad_gamma', 'grad_beta']
    for a, b, name in zip(got, want, names):
        if b is None:
            continue
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item()), \
            (name, (a - b).abs().max().item(), b.abs().max().item())
    if add:
        assert torch.equal(got[2], got[3])                  # one gradient for both summands
    # reproducible: fixed summation order of the parameter gradients
    again = run(True)
    assert torch.equal(again[4], got[4]) and torch.equal(again[5], got[5])


def test_other_norms_and_widths_run_op_by_op():
    from naive_gpt.layers.basic import fused_norm
    x = torch.randn([4, 8, 96], device='cuda')
    r = torch.randn([4, 8, 96], device='cuda')
    norm = nn.LayerNorm(96).cuda()
    assert not fused_norm.usable(norm, x)
    s, y = fused_norm.add_norm(norm, x, r)
    assert torch.equal(s, x + r) and torch.equal(y, norm(x + r))
    plain = nn.LayerNorm(1024, elementwise_affine=False).cuda()
    assert not fused_norm.usable(plain, torch.randn([2, 4, 1024], device='cuda'))


@pytest.mark.parametrize('rows,d', [(2048, 4096), (777, 1024), (33, 2048)])
@pytest.mark.parametrize('add', [False, True])
def test_rmsnorm_forward_backward(rows, d, add):
    """LLaMA's RMSNorm (reference utils.py:22-37) through the same kernels (rms flag)."""
    from naive_gpt import layers
    from naive_gpt.layers.basic import fused_norm
    gen = torch.Generator().manual_seed(rows + d + 1)
    norm = layers.LlamaRMSNorm(d).cuda()
    with torch.no_grad():
        norm.weight.copy_(1.0 + 0.1 * torch.randn([d], generator=gen))
    x = (2.0 * torch.randn([rows, d], generator=gen) + 0.5).cuda()
    r = torch.randn([rows, d], generator=gen).cuda() if add else None
    ws, wy = [torch.randn([rows, d], generator=gen).cuda() for _ in range(2)]

    def run(fused):
        xi = x.clone().requires_grad_(True)
        ri = r.clone().requires_grad_(True) if add else None
        norm.zero_grad()
        if fused:
            assert fused_norm.usable(norm, xi)
            s, y = fused_norm.add_norm(norm, xi, ri)
        else:
            s = xi + ri if add else xi
            y = norm(s)
        ((s * ws).sum() + (y * wy).sum()).backward()
        return s.detach(), y.detach(), xi.grad, ri.grad if add else None, norm.weight.grad.clone()

    for a, b, name in zip(run(True), run(False), ['s', 'y', 'grad_x', 'grad_r', 'grad_gamma']):
        if b is None:
            continue
        assert torch.allclose(a, b, rtol=1e-4, atol=1e-5 * b.abs().max().item()), \
            (name, (a - b).abs().max().item(), b.abs().max().item())


def test_block_as_pairs_equals_the_reference_wiring():
    """TransformerBlock.forward_pair + the deferred addition == x + mha(norm1(x)) ... written out."""
    from naive_gpt import layers
    torch.manual_seed(0)
    d = 256
    block = layers.TransformerBlock(
        d_model=d, n_heads=4, layernorm_fn=nn.LayerNorm(d),
        attention_fn=layers.VanillaAttention(d_head=64, p_dropout=0.0),
        feedforward_fn=layers.Feedforward(d_model=d, d_feedforward=4 * d, activation=nn.ReLU(), p_dropout=0.0),
        attention_bias=True, pre_norm=True).cuda()
    x0 = torch.randn([2, 128, d], device='cuda')
    mask = torch.full([128, 128], float('-inf'), device='cuda').triu(1)

    def run(pairs):
        x = x0.clone().requires_grad_(True)
        block.zero_grad()
        if pairs:
            y = block(x, attn_mask=mask)
        else:
            h = block.norm1(x)
            t = x + block.mha(h, h, h, attn_mask=mask)
            y = t + block.ffd(block.norm2(t))
        y.square().sum().backward()
        return [y.detach(), x.grad] + [p.grad.clone() for p in block.parameters()]

    for a, b in zip(run(True), run(False)):
        # (a key bias' gradient is zero in exact arithmetic -- softmax ignores a per-row shift of the
        # scores -- so what both sides hold there is rounding noise: absolute floor)
        assert torch.allclose(a, b, rtol=1e-4, atol=max(1e-5 * b.abs().max().item(), 1e-3))
