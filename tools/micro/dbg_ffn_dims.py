"""debug: fused LoRARoutedFFN vs its fp64 self at several dims: where do the input gradients differ"""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from torch import nn
from naive_gpt import ext, layers

for d, d_ff, T in [(1024, 4096, 8192), (2048, 8192, 4096), (2048, 8192, 2048)]:
    torch.manual_seed(0)
    ffn = layers.LoRARoutedFFN(d_lora=16, block_size=d_ff // 4, d_model=d, d_feedforward=d_ff,
                               activation=nn.ReLU()).cuda()
    gen = torch.Generator().manual_seed(7)
    for n, p in ffn.named_parameters():
        if n.endswith('lora.right.weight'):
            p.data.copy_(0.05 * torch.randn(p.shape, generator=gen).cuda())
    x = torch.randn([1, T, d], generator=gen).cuda()
    w = torch.randn([1, T, d], generator=gen).cuda()
    xi = x.clone().requires_grad_(True)
    y = ffn(xi)
    (y * w).sum().backward()
    ffn64 = copy.deepcopy(ffn).double()
    ffn64.zero_grad()
    x64 = x.double().requires_grad_(True)
    y64 = ffn64(x64)
    (y64 * w.double()).sum().backward()
    err = (xi.grad.double() - x64.grad).abs().amax(-1)[0]
    scale = float(x64.grad.abs().max())
    bad = (err > 1e-3 * scale).nonzero().flatten()
    print('d {} d_ff {} T {}: gx worst {:.2e} of scale {:.1f}; rows over 1e-3: {} of {}; y worst {:.2e}'.format(
        d, d_ff, T, float(err.max()) / scale, scale, bad.numel(), T,
        float((y.double() - y64).abs().max() / y64.abs().max())))
    print('   bad rows', bad[:12].tolist(), ['{:.3f}'.format(float(e) / scale) for e in err[bad[:12]]])
    # does the fused forward's h disagree in SIGN with fp64 anywhere?  (h is internal: recompute pre-activations)
    with torch.no_grad():
        prob = ffn.router(x[0])
        top = torch.topk(prob, 2, dim=-1).indices
        nb, bs = 4, d_ff // 4
        for t in bad[:3].tolist():
            xt = x[0, t].double()
            for i in top[t].tolist():
                sl = slice(i * bs, (i + 1) * bs)
                c = 2.0 * prob[t, i].double()
                pre = c * (ffn.fc1.weight.double()[sl] @ xt + ffn.fc1.bias.double()[sl]) \
                    + ffn.fc1.lora.right.weight.double()[sl] @ (ffn.fc1.lora.left.weight.double().t() @ xt)
                print('   token', t, 'block', i, 'min |pre|', float(pre.abs().min()), 'coeff', float(c))
