"""debug: fused vs op-by-op block paths with third consumers, switch by switch"""
import contextlib, io, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from torch import nn
from naive_gpt import ext, layers, utils
from naive_gpt.layers.sparse import attention as A, grouped

d, heads, d_ff, N, S = 1024, 16, 4096, 2, 512
torch.manual_seed(0)
block = layers.TransformerBlock(
    d_model=d, n_heads=heads, layernorm_fn=nn.LayerNorm(d),
    attention_fn=layers.VanillaAttention(d_head=d // heads, p_dropout=0.0),
    feedforward_fn=layers.Feedforward(d_model=d, d_feedforward=d_ff, activation=nn.GELU(), p_dropout=0.0),
    attention_bias=True, pre_norm=True)
with contextlib.redirect_stdout(io.StringIO()):
    for stage in ('lora', 'ffn', 'mha_v1', 'mha_v2'):
        block = utils.ModuleUpgrader(utils.SparseLoRAHandler(d_lora=16, stage=stage)).visit(block)
for name, p in block.named_parameters():
    if name.endswith('lora.right.weight'):
        p.data.normal_(0, 0.02)
block = block.cuda()
x0 = torch.randn([N, S, d], device='cuda')
wh = torch.randn([N, S, d], device='cuda')
wy = torch.randn([N, S, d], device='cuda')
taps = {}
block.mha.attn_fn.register_forward_pre_hook(lambda m, args: taps.__setitem__('q', args[0]))
block.ffd.register_forward_pre_hook(lambda m, args: taps.__setitem__('h', args[0]))
names = ['x'] + [n for n, p in block.named_parameters() if p.requires_grad]


def run(terms=(1, 1, 1, 1)):
    x = x0.clone().requires_grad_(True)
    block.zero_grad()
    block.mha.attn_fn.arm()
    y = block(x)
    loss = terms[0] * (y * wy).sum() + terms[1] * 50.0 * block.mha.attn_fn.loss \
        + terms[2] * 0.5 * taps['q'].square().sum() + terms[3] * (taps['h'] * wh).sum()
    loss.backward()
    return [x.grad.clone()] + [p.grad.clone() if p.grad is not None else torch.zeros_like(p)
                               for p in block.parameters() if p.requires_grad]


def cmp(tag, a, b):
    out = []
    for n, u, v in zip(names, a, b):
        e = float((u - v).abs().max() / v.abs().max().clamp_min(1e-30))
        if e > 1e-4:
            out.append('{} {:.1e}'.format(n, e))
    print(tag, '|', '; '.join(out) if out else 'all < 1e-4')


usable, supported = grouped.usable, ext.attention_mfma_supported
for terms in [(1, 1, 1, 1), (1, 0, 0, 0), (0, 1, 0, 0), (0, 0, 1, 0), (0, 0, 0, 1)]:
    base = run(terms)
    grouped.usable = lambda *a, **k: False
    cmp('{} ffn loop      '.format(terms), run(terms), base)
    grouped.usable = usable
    A.FOLD_PQ_LOSS = False
    cmp('{} pq loss apart '.format(terms), run(terms), base)
    ext.attention_mfma_supported = lambda *a: False
    cmp('{} gather attn   '.format(terms), run(terms), base)
    A.FOLD_PQ_LOSS = True
    ext.attention_mfma_supported = supported
    cmp('{} again         '.format(terms), run(terms), base)
