"""Fused PQ training loss (spt_pq_loss_*, pq_loss.hip) against

* the golden produced by the IMPORTED REFERENCE's PQV2.forward('train')
  (tests/golden/pq.npz: loss, grad_z, grad_weight), and
* the reference-shaped composition of the same loss (quantizer.py:80-111) run on the CPU
  through the oracle's cdist, at the sizes of the layer (E = 64: M = 8, D = 8), at D = 4
  and with a non-unit upstream gradient.

Tolerance: fp32 within 1e-3 relative (north_star); the fused kernel sums in a different
order than torch and evaluates softmax(-log d) as normalised inverse distances.
"""
import os

import numpy as np
import pytest
import torch

from oracle import ext_stub

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
pytestmark = pytest.mark.gpu


def composed_cpu(z, weight, upstream):
    """quantizer.py:80-111 with the oracle's cdist (extension/cdist.cu restated)."""
    M, C, D = weight.shape

    class CDist(torch.autograd.Function):
        @staticmethod
        def forward(ctx, q, t):
            d, i = ext_stub.cdist_forward_cuda(q, t)
            ctx.save_for_backward(q, t)
            ctx.mark_non_differentiable(i)
            return d, i

        @staticmethod
        def backward(ctx, gd, _gi):
            q, t = ctx.saved_tensors
            return tuple(ext_stub.cdist_backward_cuda(q, t, gd.contiguous()))

    z = z.clone().requires_grad_(True)
    weight = weight.clone().requires_grad_(True)
    flat = z.flatten(end_dim=-2)
    flat = flat.view([flat.size(0), M, D]).transpose(0, 1).contiguous()
    distance, indices = CDist.apply(flat, weight)
    index = indices.long().unsqueeze(-1).expand([-1, -1, D])
    z_q = torch.gather(weight, dim=1, index=index)
    soft = torch.softmax(-torch.log(torch.clamp(distance, min=1e-5)), dim=-1)
    z_w = torch.matmul(soft, weight)
    mse = torch.nn.MSELoss()
    loss = mse(z_w, z_q) + mse(flat, z_q)
    (loss * upstream).backward()
    return loss.detach(), z.grad, weight.grad


def close(got, want, name, rtol=1e-3, atol=1e-6):
    got, want = got.detach().cpu().double(), want.detach().cpu().double()
    scale = want.abs().max().item()
    err = (got - want).abs().max().item()
    assert err <= atol + rtol * scale, '{}: max err {:.3e} vs scale {:.3e}'.format(name, err, scale)


def test_matches_imported_reference_golden():
    from naive_gpt import layers
    g = dict(np.load(os.path.join(GOLD, 'pq.npz'), allow_pickle=False))
    pq = layers.PQV2(d_codeword=8, n_codewords=16, n_subspaces=4)
    pq.weight.data.copy_(torch.from_numpy(g['v2.weight']))
    pq = pq.cuda()
    z = torch.from_numpy(g['v2.z']).cuda().requires_grad_(True)
    loss = pq.train_loss(z)
    assert loss.dim() == 0
    loss.backward()
    close(loss, torch.from_numpy(g['v2.loss']), 'loss')
    close(z.grad, torch.from_numpy(g['v2.grad_z']), 'grad_z')
    close(pq.weight.grad, torch.from_numpy(g['v2.grad_weight']), 'grad_weight')


@pytest.mark.parametrize('shape,M,D,upstream', [
    ([2, 48, 4, 64], 8, 8, 1.0),       # the layer's [N, S, H, E] head layout
    ([3, 40, 64], 8, 8, 0.01),         # [B, S, E]; the recipe's 1e-2 aux weight
    ([5, 33, 64], 16, 4, 1.0),         # D = 4
    ([7, 19, 16], 2, 8, -2.5),         # M = 2, ragged row count
    ([1, 1, 64], 8, 8, 1.0),           # a single vector: most lanes idle
])
def test_matches_composed_oracle(shape, M, D, upstream):
    from naive_gpt import kernels
    torch.manual_seed(sum(shape) + M)
    z = torch.randn(shape)
    weight = torch.randn([M, 16, D])
    want_loss, want_gz, want_gw = composed_cpu(z, weight, upstream)

    zg = z.cuda().requires_grad_(True)
    wg = weight.cuda().requires_grad_(True)
    loss = kernels.pq_loss(zg, wg)
    (loss * upstream).backward()
    close(loss, want_loss, 'loss')
    close(zg.grad, want_gz, 'grad_z')
    close(wg.grad, want_gw, 'grad_weight')


def test_exact_centroid_inputs_hit_the_clamp():
    """z equal to codewords: d = 0 < 1e-5, the clamp passes no gradient to that distance."""
    from naive_gpt import kernels
    torch.manual_seed(5)
    weight = torch.randn([8, 16, 8])
    pick = torch.randint(0, 16, [64, 8])
    z = torch.stack([weight[m, pick[:, m]] for m in range(8)], dim=1).reshape(64, 64)
    z[::2] += 0.25 * torch.randn(32, 64)          # half of the rows off-centroid
    want_loss, want_gz, want_gw = composed_cpu(z, weight, 1.0)
    zg, wg = z.cuda().requires_grad_(True), weight.cuda().requires_grad_(True)
    loss = kernels.pq_loss(zg, wg)
    loss.backward()
    assert torch.isfinite(zg.grad).all() and torch.isfinite(wg.grad).all()
    close(loss, want_loss, 'loss')
    close(zg.grad, want_gz, 'grad_z')
    close(wg.grad, want_gw, 'grad_weight')


def test_full_size_is_deterministic_and_matches_composed_gpu():
    """BASELINE size (N=16, S=512, H=16, E=64): two runs are bit-identical (fixed reduction
    order) and agree with the op-by-op composition over kernels.cdist on the GPU."""
    from naive_gpt import layers
    torch.manual_seed(0)
    pq = layers.PQV2(d_codeword=8, n_codewords=16, n_subspaces=8).cuda()
    z = torch.randn([16, 512, 16, 64], device='cuda')

    def fused():
        zz = z.clone().requires_grad_(True)
        pq.weight.grad = None
        loss = pq.train_loss(zz)
        loss.backward()
        return loss.detach().clone(), zz.grad.clone(), pq.weight.grad.clone()

    a, b = fused(), fused()
    for x, y in zip(a, b):
        assert torch.equal(x, y)
    zz = z.clone().requires_grad_(True)
    pq.weight.grad = None
    loss = pq('train', z=zz)[-1]
    loss.backward()
    close(a[0], loss, 'loss')
    close(a[1], zz.grad, 'grad_z')
    close(a[2], pq.weight.grad, 'grad_weight', rtol=2e-3)


def test_unsupported_shapes_fall_back_and_raise():
    from naive_gpt import ext, layers
    z = torch.randn([4, 8, 96], device='cuda')
    table = torch.randn([6, 16, 16], device='cuda')       # D = 16, M = 6
    assert not ext.pq_loss_supported(z, table)
    with pytest.raises(RuntimeError):
        ext.pq_loss_forward(z, table)
    pq = layers.PQV2(d_codeword=16, n_codewords=16, n_subspaces=6).cuda()
    loss = pq.train_loss(z.requires_grad_(True))           # composed path
    loss.backward()
    assert torch.isfinite(loss)


def test_loss_forward_leaves_the_pq_codes_of_the_encode_pass():
    """spt_pq_loss_forward_codes: the loss of the plain forward and, bit for bit, the codes of
    spt_pq_encode_heads (the loss's argmin is the code)."""
    from naive_gpt import ext, kernels
    gen = torch.Generator().manual_seed(11)
    N, S, H, M, C, D = 2, 256, 16, 8, 16, 8
    z = torch.randn([N, S, H, M * D], generator=gen).cuda()
    table = torch.randn([M, C, D], generator=gen).cuda()
    loss, codes = ext.pq_loss_forward(z, table, want_codes=True)
    assert torch.equal(loss, ext.pq_loss_forward(z, table))
    assert torch.equal(codes, ext.pq_encode_heads(z, table))
    zr = z.clone().requires_grad_(True)
    tr = table.clone().requires_grad_(True)
    l2, c2 = kernels.pq_loss_and_codes(zr, tr)
    assert not c2.requires_grad and torch.equal(c2, codes)
    l2.backward()
    z0 = z.clone().requires_grad_(True)
    t0 = table.clone().requires_grad_(True)
    kernels.pq_loss(z0, t0).backward()
    assert torch.equal(zr.grad, z0.grad) and torch.equal(tr.grad, t0.grad)


def test_q_and_k_in_one_buffer_take_one_pass():
    """spt_pq_loss_*_parts: q and k back to back (the joint projection's output) against one table:
    the sum of the two losses, both code tensors, both input gradients (added into their targets) and
    the summed table gradient from ONE forward and ONE backward launch -- against the calls per tensor."""
    from naive_gpt import ext
    torch.manual_seed(0)
    N, S, H, M, D = 2, 256, 4, 8, 8
    qk = torch.randn([2, N, S, H, M * D], device='cuda')
    q, k = qk[0], qk[1]
    table = torch.randn([M, 16, D], device='cuda')
    assert ext.back_to_back(q, k) and not ext.back_to_back(k, q) and not ext.back_to_back(q, q.clone())
    loss, cq, ck = ext.pq_loss_forward_pair(q, k, table)
    lq, cq1 = ext.pq_loss_forward(q, table, want_codes=True)
    lk, ck1 = ext.pq_loss_forward(k, table, want_codes=True)
    assert torch.equal(cq, cq1) and torch.equal(ck, ck1)
    assert torch.allclose(loss, lq + lk, rtol=1e-5)
    upstream = torch.tensor(0.7, device='cuda')
    base = torch.randn([2, N, S, H, M * D], device='cuda')
    got = base.clone()
    gt = ext.pq_loss_backward_pair(q, k, table, upstream, got[0], got[1])
    want = base.clone()
    _, gt_q = ext.pq_loss_backward(q, table, upstream, accumulate_into=want[0])
    _, gt_k = ext.pq_loss_backward(k, table, upstream, accumulate_into=want[1])
    assert torch.equal(got, want)
    assert torch.allclose(gt, gt_q + gt_k, rtol=1e-4, atol=1e-6 * float(gt.abs().max()))
    with pytest.raises(Exception):
        ext.pq_loss_forward_pair(q, k.clone(), table)
