"""debug: the ReLU GEMM of the routed FFN at BERT-large dims: sign of h against fp64 pre-activations"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
from naive_gpt.layers.sparse import grouped

d, d_ff, T, nb, r = 1024, 4096, 8192, 4, 16
bs = d_ff // nb
gen = torch.Generator().manual_seed(7)
x = torch.randn([T, d], generator=gen).cuda()
w1 = ((torch.rand([d_ff, d], generator=gen) * 2 - 1) / 32).cuda()
b1 = ((torch.rand([d_ff], generator=gen) * 2 - 1) / 32).cuda()
l1 = torch.randn([d, r], generator=gen).cuda()
r1 = (0.05 * torch.randn([d_ff, r], generator=gen)).cuda()
prob = torch.sigmoid(0.3 * torch.randn([T, nb], generator=gen)).cuda()
bk = grouped.make_buckets(prob, 2, scale=2.0)
rows = bk.token.numel()
for images in (True, False):
    for queue in (None, 0):
        u, x_img, x_norm = grouped._down(x, l1, images, True)
        h, s = ext.grouped_gemm_fused(
            x, w1, bk.offsets, nb, n=bs, k=d, w_group_stride=bs * d, w_ldn=d, w_ldk=1,
            n_rows=rows, gather=bk.token, bias=b1, rowscale=bk.coeff, a2=u, gather2=bk.token, b2=r1,
            b2_group_stride=bs * r, epilogue=ext.EPI_ACT, activation=ext.ACT_RELU, keep_preact=True,
            a_image=x_img, w_image=ext.weight_image(w1) if images else None,
            a_norm=x_norm, w_norm=ext.weight_row_norms(w1), relu_queue_entries=queue)
        torch.cuda.synchronize()
        # fp64 truth per row p
        tok, blk = bk.token_long, bk.block
        pre64 = torch.empty([rows, bs], dtype=torch.float64, device='cuda')
        for g in range(nb):
            sel = (blk == g).nonzero().flatten()
            xs = x[tok[sel]].double()
            c = bk.coeff[sel].double().unsqueeze(1)
            sl = slice(g * bs, (g + 1) * bs)
            pre64[sel] = c * (xs @ w1[sl].double().t() + b1[sl].double()) + (xs @ l1.double()) @ r1[sl].double().t()
        wrong = ((h > 0) != (pre64 > 0))
        print('images', images, 'queue', queue, ': sign mismatches', int(wrong.sum()), 'of', wrong.numel(),
              '| max |pre64| among them {:.3e}'.format(float(pre64[wrong].abs().max()) if wrong.any() else 0.0),
              '| max |s - pre64| {:.3e}'.format(float((s.double() - pre64).abs().max())),
              '| elements with |pre64| < 5e-5:', int((pre64.abs() < 5e-5).sum()))
        if wrong.any():
            idx = wrong.nonzero()[:8]
            for p, n in idx.tolist():
                print('    p', p, 'n', n, 'block', int(blk[p]), 'pre64 {:.3e} s {:.3e} h {:.3e}'.format(
                    float(pre64[p, n]), float(s[p, n]), float(h[p, n])))
        if ext.LAST_RELU_QUEUE is not None:
            q = ext.LAST_RELU_QUEUE
            counts = q[:256 * 64].view(torch.int32).view(256, 16)[:, 0]
            print('    queue counts: total', int(counts.sum()), 'max per segment', int(counts.max()))
