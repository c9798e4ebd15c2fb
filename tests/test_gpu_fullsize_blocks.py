"""BASELINE.json configs[3] and configs[4] at their BLOCK dimensions: the LoRA linears, the routed
FFN and the four-stage-upgraded TransformerBlock at OPT-1.3B dims (d 2048, 32 x 64 heads,
d_ff 8192 = 4 blocks of 2048, S 2048, N 2: reference default `--name opt-2048`,
script/0-profile.py:20-23) and LLaMA-7B dims (d 4096, 32 x 128 rotary heads, SiLU-gated d_ff
11008 = 4 blocks of 2752, S 2048, N 1: lora_ffn.py:164-225).  test_gpu_fullsize_long.py covers
the attention heads of the same two configurations.

Every case
* compares the fused engine (grouped split-bf16 GEMMs, spt_lora_down, spt_tall_tn, ...) with the
  per-block torch loop / library GEMM path of the same layer -- the reference-shaped code that the
  CPU goldens pin to the imported reference (tests/test_layers_golden.py) -- values, input
  gradient and every parameter gradient;
* checks a slice of the rows against fp64 algebra of the reference formula
  (lora.py:70-80, lora_ffn.py:87-111, :196-222);
* asserts WHICH engine ran (ext.PATH_COUNTS): a parity test that silently compared a fallback
  with itself would prove nothing.
"""
import contextlib
import io

import pytest
import torch
from torch import nn

pytestmark = pytest.mark.gpu

CONFIGS = {
    # BASELINE configs[3]: OPT-1.3B dims
    'opt1b3': dict(family='opt', d=2048, heads=32, d_ff=8192, N=2, S=2048),
    # BASELINE configs[4]: LLaMA-7B dims, one GPU's micro-batch
    'llama7b': dict(family='llama', d=4096, heads=32, d_ff=11008, N=1, S=2048),
}


def _scaled_close(got, want, rtol=1e-3, frac=3e-4):
    return torch.allclose(got, want, rtol=rtol, atol=frac * want.abs().max().item())


def _worst(got, want):
    return float((got.detach().double() - want.detach().double()).abs().max()) \
        / max(float(want.detach().abs().max()), 1e-30)


def _rows_close(got, want, rtol=1e-3, frac=3e-4, bad_rows=0.0):
    """_scaled_close per row (token), tolerating a fraction `bad_rows` of rows: a ReLU
    pre-activation within rounding of zero takes either sign in ANY fp32 evaluation (the library
    GEMM's included), and one flipped sign moves that token's input gradient by O(1)."""
    got, want = got.reshape(-1, got.size(-1)), want.reshape(-1, want.size(-1))
    tol = frac * float(want.abs().max()) + rtol * want.abs()
    bad = ((got - want).abs() > tol).any(dim=-1).float().mean()
    return float(bad) <= bad_rows


@pytest.fixture(autouse=True)
def _fresh_counts():
    from naive_gpt import ext
    ext.reset_paths()
    ext.drop_images()
    yield
    ext.drop_images()
    torch.cuda.empty_cache()


def _paths(site):
    from naive_gpt import ext
    return {k[1]: v for k, v in ext.paths_taken(site).items()}


def _randomise_adapters(module, gen, scale=0.05):
    """`right` tables start at zero (a fresh adapter is the identity): give them values so that
    every LoRA product and every table gradient is exercised."""
    for name, p in module.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.copy_(scale * torch.randn(p.shape, generator=gen).to(p.device))


# ------------------------------------------------------------------------------ LoRA linears

@pytest.mark.parametrize('name', list(CONFIGS))
def test_lora_linears_at_block_dims(name, monkeypatch):
    """`LoRALinear` (lora.py:70-80) d -> d at T = N * S rows: matrix-core GEMM + spt_lora_down +
    spt_tall_tn against the library path, an fp64 slice, and the joint q / k / v function against
    three separate layers."""
    from naive_gpt import ext, layers
    from naive_gpt.layers.tuning import lora
    c = CONFIGS[name]
    d, T = c['d'], c['N'] * c['S']
    gen = torch.Generator().manual_seed(11)
    bias = c['family'] == 'opt'
    lins = [layers.LoRALinear(d_lora=16, in_features=d, out_features=d, bias=bias).cuda() for _ in range(3)]
    for m in lins:
        _randomise_adapters(m, gen)
    x = torch.randn([c['N'], c['S'], d], generator=gen).cuda()
    w = torch.randn([c['N'], c['S'], d], generator=gen).cuda()

    def run(m, fused):
        m.zero_grad()
        xi = x.clone().requires_grad_(True)
        if not fused:
            monkeypatch.setattr(lora, '_mfma_linear_usable', lambda *a, **k: False)
            monkeypatch.setattr(ext, 'lora_down_supported', lambda *a, **k: False)
            monkeypatch.setattr(ext, 'tall_tn_supported', lambda *a, **k: False)
        y = m(xi)
        (y * w).sum().backward()
        monkeypatch.undo()
        return y.detach(), xi.grad, m.lora.left.weight.grad.clone(), m.lora.right.weight.grad.clone()

    ext.reset_paths()
    got = run(lins[0], True)
    assert _paths('lora_linear') == {'mfma': 1} and _paths('lora_linear_backward') == {'mfma': 1}
    assert _paths('lora_down') == {'kernel': 2}
    # (the two table gradients: one batched launch when the layer is square, else two)
    assert _paths('tall_tn') in ({'kernel': 2}, {'kernel_batch': 1}), _paths('tall_tn')
    # forward: the LDS-DMA k-loop on the fp32 rows (a lone layer makes no image: "A32"); backward dX:
    # the register path (its weight is read n-contiguous: no faster from an image, tools/bench_gemm.py)
    assert _paths('grouped_gemm') == {'a32': 1, 'register': 1}, _paths('grouped_gemm')
    want = run(lins[0], False)
    for g_, w_, what in zip(got, want, ('y', 'grad_x', 'grad_left', 'grad_right')):
        assert _scaled_close(g_, w_), (what, _worst(g_, w_))
    # fp64 algebra of the reference formula on a slice of the rows
    rows = torch.arange(0, T, T // 64, device='cuda')
    m = lins[0]
    x64 = x.view(T, d)[rows].double()
    y64 = x64 @ m.weight.double().t() + (x64 @ m.lora.left.weight.double()) @ m.lora.right.weight.double().t()
    if bias:
        y64 = y64 + m.bias.double()
    assert _worst(got[0].view(T, d)[rows], y64) <= 1e-3
    w64 = w.view(T, d)[rows].double()
    gx64 = w64 @ m.weight.double() + (w64 @ m.lora.right.weight.double()) @ m.lora.left.weight.double().t()
    assert _worst(got[1].view(T, d)[rows], gx64) <= 1e-3
    # the three projections as one function (lora._FrozenLoRAQKV) == three layers
    assert lora.qkv_usable(x, lins)
    ws = [torch.randn([c['N'], c['S'], d], generator=gen).cuda() for _ in range(3)]
    for m in lins:
        m.zero_grad()
    xi = x.clone().requires_grad_(True)
    outs = lora.qkv(xi, lins, v_apart=c['family'] == 'llama')
    sum((o * wi).sum() for o, wi in zip(outs, ws)).backward()
    joint = [o.detach() for o in outs] + [xi.grad] + [m.lora.left.weight.grad.clone() for m in lins] \
        + [m.lora.right.weight.grad.clone() for m in lins]
    for m in lins:
        m.zero_grad()
    xi = x.clone().requires_grad_(True)
    outs = [m(xi) for m in lins]
    sum((o * wi).sum() for o, wi in zip(outs, ws)).backward()
    apart = [o.detach() for o in outs] + [xi.grad] + [m.lora.left.weight.grad.clone() for m in lins] \
        + [m.lora.right.weight.grad.clone() for m in lins]
    for i, (a, b) in enumerate(zip(joint, apart)):
        assert _scaled_close(a, b), (i, _worst(a, b))


# ------------------------------------------------------------------------------ routed FFN

def _make_ffn(c, gen):
    from naive_gpt import layers
    bs = c['d_ff'] // 4                                   # adapter.py:155-183: block_size = d_ff // 4
    torch.manual_seed(0)
    if c['family'] == 'opt':
        ffn = layers.LoRARoutedFFN(d_lora=16, block_size=bs, d_model=c['d'], d_feedforward=c['d_ff'],
                                   activation=nn.ReLU()).cuda()
    else:
        ffn = layers.LoRARoutedLLaMaFFN(d_lora=16, block_size=bs, d_model=c['d'],
                                        d_feedforward=c['d_ff'], activation=nn.SiLU()).cuda()
    _randomise_adapters(ffn, gen)
    return ffn


def _ffn_fp64(ffn, x, family, w=None):
    """The reference formula (lora_ffn.py:87-111 / :196-222) in fp64 for a few tokens: every block
    evaluated, the token's top-k (taken from the layer's own fp32 router: the same decisions) kept.
    With `w` (the upstream gradient of these tokens) also d <w, y> / d x by fp64 autograd -- tokens
    are independent given the parameters."""
    nb, bs = ffn.n_blocks, ffn.block_size
    with torch.no_grad():
        prob32 = ffn.router(x)                                                # fp32, as the layer
    chosen = torch.zeros_like(prob32, dtype=torch.bool)
    chosen.scatter_(1, torch.topk(prob32, k=nb // 2, dim=-1, sorted=False).indices, True)
    x64 = x.double().requires_grad_(w is not None)
    lin = ffn.router[0]
    coeff = 2.0 * torch.sigmoid(x64 @ lin.weight.double().t() + lin.bias.double())
    y = torch.zeros_like(x64)

    def lora_of(linear):
        return linear.lora.left.weight.double(), linear.lora.right.weight.double()

    for i in range(nb):
        c_i = coeff[:, i:i + 1]
        sl = slice(i * bs, (i + 1) * bs)
        if family == 'opt':
            l1, r1 = lora_of(ffn.fc1)
            l2, r2 = lora_of(ffn.fc2)
            h = c_i * (x64 @ ffn.fc1.weight.double()[sl].t() + ffn.fc1.bias.double()[sl]) + (x64 @ l1) @ r1[sl].t()
            h = torch.relu(h)
            out = (h @ l2[sl]) @ r2.t() + c_i * (h @ ffn.fc2.weight.double()[:, sl].t())
        else:
            lg, rg = lora_of(ffn.gate)
            ls, rs = lora_of(ffn.side)
            ld, rd = lora_of(ffn.down)
            g = c_i * (x64 @ ffn.gate.weight.double()[sl].t()) + (x64 @ lg) @ rg[sl].t()
            s = c_i * (x64 @ ffn.side.weight.double()[sl].t()) + (x64 @ ls) @ rs[sl].t()
            h = nn.functional.silu(g) * s
            out = c_i * (h @ ffn.down.weight.double()[:, sl].t()) + (h @ ld[sl]) @ rd.t()
        y = y + out * chosen[:, i:i + 1]
    if family == 'opt':
        y = y + ffn.fc2.bias.double()
    if w is None:
        return y.detach()
    grad_x, = torch.autograd.grad(y, x64, w.double())
    return y.detach(), grad_x


@pytest.mark.parametrize('name', list(CONFIGS))
def test_routed_ffn_at_block_dims(name):
    from naive_gpt import ext
    from naive_gpt.layers.sparse import grouped
    c = CONFIGS[name]
    gen = torch.Generator().manual_seed(7)
    ffn = _make_ffn(c, gen)
    d = c['d']
    x = torch.randn([c['N'], c['S'], d], generator=gen).cuda()
    w = torch.randn([c['N'], c['S'], d], generator=gen).cuda()

    def run(fused):
        ffn.zero_grad()
        xi = x.clone().requires_grad_(True)
        saved = grouped.usable
        if not fused:
            grouped.usable = lambda *a, **k: False
        try:
            y = ffn(xi)
        finally:
            grouped.usable = saved
        (y * w).sum().backward()
        return y.detach(), xi.grad, {n: p.grad.clone() for n, p in ffn.named_parameters()
                                     if p.grad is not None}

    ext.reset_paths()
    y1, gx1, g1 = run(True)
    # which engine: fused routed FFN on the grouped GEMM's image path, every small product a kernel
    # of this library (no library small GEMM, no torch routing)
    assert _paths('routed_ffn') == {'fused': 1}
    assert _paths('route_topk') == {'kernel': 1}
    assert set(_paths('lora_down')) == {'kernel'}, _paths('lora_down')
    assert set(_paths('tall_tn')) <= {'kernel', 'kernel_batch'} and set(_paths('tall_tn_blocks')) <= {'kernel', 'kernel_batch'}
    if c['family'] == 'opt':
        assert set(_paths('lora_down_grouped')) == {'kernel'}
    gemms = _paths('grouped_gemm')
    # forward: the up GEMM(s) from pre-split images, the down GEMM from the image of its weight and
    # the fp32 rows of h ("A32": h has one reader, no image of it is made); backward: the derivative
    # GEMM(s) from images, the dX products from their fp32 operand (register path) -- DESIGN.md 5.1
    assert gemms.get('image', 0) >= (2 if c['family'] == 'opt' else 3) and gemms.get('a32', 0) == 1, gemms
    ext.reset_paths()
    y0, gx0, g0 = run(False)
    assert _paths('routed_ffn') == {'torch_loop': 1}
    assert set(g0) == set(g1) and len(g0) >= 6
    # The arbiter: the same layer in fp64 (its dtype sends it through the per-block torch loop) on
    # the whole batch.  The FUSED path must agree with it everywhere to the north-star bar; the
    # fp32 loop agrees per token except where a ReLU pre-activation lies within fp32 rounding of
    # zero -- there the library GEMM's summation order decides its sign (the fused path recomputes
    # those in exact fp32), which moves that token's input gradient by O(1).
    import copy
    ffn64 = copy.deepcopy(ffn).double()
    ffn64.zero_grad()
    x64 = x.double().requires_grad_(True)
    y64 = ffn64(x64)
    (y64 * w.double()).sum().backward()
    g64 = {n: p.grad for n, p in ffn64.named_parameters() if p.grad is not None}
    relu = c['family'] == 'opt'
    assert _scaled_close(y1.double(), y64.detach()), _worst(y1, y64)
    # (one token in a few thousand may own a pre-activation within fp32 rounding of zero, whose sign
    # no fp32 evaluation can promise: 1 of 4096 at the OPT dims, |pre| = 9e-7)
    assert _rows_close(gx1.double(), x64.grad, bad_rows=0.001 if relu else 0.0), _worst(gx1, x64.grad)
    assert set(g64) == set(g1)
    for n in g1:
        # (ReLU: the same one-in-millions pre-activation shows in the table gradients as ONE row --
        # hidden unit -- of fc1.lora.right: which element it is depends on the last bit of the router
        # coefficient, e.g. on whether the sigmoid ran in torch or inside the routing launch)
        if relu and g1[n].dim() == 2 and g1[n].size(0) >= 1024:
            assert _rows_close(g1[n].double(), g64[n], frac=1e-3, bad_rows=2.0 / g1[n].size(0)), \
                (n, _worst(g1[n], g64[n]))
        else:
            # (... and in the router's gradients as that one token's share: 1.7e-3 of the largest
            # element at these dimensions, measured)
            assert _scaled_close(g1[n].double(), g64[n], frac=3e-3 if relu else 3e-4), (n, _worst(g1[n], g64[n]))
    assert _scaled_close(y0.double(), y64.detach()), _worst(y0, y64)
    assert _rows_close(gx0.double(), x64.grad, bad_rows=0.01 if relu else 0.0), _worst(gx0, x64.grad)
    for n in g0:
        assert _scaled_close(g0[n].double(), g64[n], rtol=2e-3, frac=5e-3 if relu else 3e-4), \
            (n, _worst(g0[n], g64[n]))
    # and an independent restatement of the formula (not the layer's own code) on a slice
    T = c['N'] * c['S']
    rows = torch.arange(5, T, T // 96, device='cuda')
    want_y, want_gx = _ffn_fp64(ffn, x.view(T, d)[rows], c['family'], w.view(T, d)[rows])
    assert _worst(y1.view(T, d)[rows], want_y) <= 1e-3
    assert _worst(gx1.view(T, d)[rows], want_gx) <= 1e-3


# ------------------------------------------------------------------------------ the block

def _make_block(c):
    from naive_gpt import layers, utils
    d, heads, d_ff, S = c['d'], c['heads'], c['d_ff'], c['S']
    torch.manual_seed(0)
    if c['family'] == 'llama':
        block = layers.TransformerBlock(
            d_model=d, n_heads=heads, layernorm_fn=layers.LlamaRMSNorm(d),
            attention_fn=layers.RotaryAttention(d_head=d // heads, p_dropout=0.0, max_length=S),
            feedforward_fn=layers.LLaMaFeedforward(d_model=d, d_feedforward=d_ff, activation=nn.SiLU()),
            attention_bias=False, pre_norm=True)
    else:
        block = layers.TransformerBlock(
            d_model=d, n_heads=heads, layernorm_fn=nn.LayerNorm(d),
            attention_fn=layers.VanillaAttention(d_head=d // heads, p_dropout=0.0),
            feedforward_fn=layers.Feedforward(d_model=d, d_feedforward=d_ff, activation=nn.ReLU(),
                                              p_dropout=0.0),
            attention_bias=True, pre_norm=True)
    with contextlib.redirect_stdout(io.StringIO()):
        for stage in ['lora', 'ffn', 'mha_v1', 'mha_v2']:         # script/0-profile.py:183-189
            block = utils.ModuleUpgrader(utils.SparseLoRAHandler(d_lora=16, stage=stage)).visit(block)
    return block.cuda()


@pytest.mark.parametrize('name', list(CONFIGS))
def test_upgraded_block_step_at_block_dims(name):
    """One four-stage-upgraded TransformerBlock, forward + backward (the step of
    script/0-profile.py:203-211) with the PQ loss armed: every engine choice asserted; two runs
    bit-identical in the output; finite everywhere; the directional derivative along the last
    adapter table (in which the block is exactly linear: lora_ffn.py:109-111 / :220-222) equals the
    finite difference."""
    from naive_gpt import ext
    c = CONFIGS[name]
    block = _make_block(c)
    gen = torch.Generator().manual_seed(3)
    _randomise_adapters(block, gen)
    d = c['d']
    x = torch.randn([c['N'], c['S'], d], generator=gen).cuda()
    w = torch.randn([c['N'], c['S'], d], generator=gen).cuda()

    def run(arm):
        block.zero_grad()
        if arm:
            block.mha.attn_fn.arm()
        xi = x.clone().requires_grad_(True)
        y = block(xi)
        loss = (y * w).sum()
        if arm:
            loss = loss + 1e-2 * block.mha.attn_fn.loss
        loss.backward()
        return y.detach(), xi.grad, {n: p.grad.clone() for n, p in block.named_parameters()
                                     if p.grad is not None}

    ext.reset_paths()
    y1, gx1, g1 = run(True)
    assert _paths('qkv_projection') == {'joint': 1}
    assert _paths('attention') == {'mfma': 1}
    assert set(_paths('norm')) == {'kernel'}        # (forward twice, recomputed twice in the backward)
    assert _paths('lora_linear') == {'mfma': 1} and _paths('lora_linear_backward') == {'mfma': 1}   # linear_o
    assert _paths('routed_ffn') == {'fused': 1} and _paths('route_topk') == {'kernel': 1}
    assert 'library' not in _paths('lora_down'), _paths('lora_down')
    assert set(_paths('tall_tn')) <= {'kernel', 'kernel_batch'} and set(_paths('tall_tn_blocks')) <= {'kernel', 'kernel_batch'}
    trainable = {n for n, p in block.named_parameters() if p.requires_grad}
    assert set(g1) == trainable and 'mha.attn_fn.quantizer.weight' in g1
    for t in [y1, gx1] + list(g1.values()):
        assert torch.isfinite(t).all()
    y2, gx2, g2 = run(True)
    assert torch.equal(y1, y2)
    assert _worst(gx2, gx1) <= 1e-5
    for n in g1:
        assert _worst(g2[n], g1[n]) <= 1e-4, n
    # exact linear direction: the FFN's last `right` table
    last = block.ffd.fc2 if c['family'] == 'opt' else block.ffd.down
    table = last.lora.right.weight
    direction = torch.randn(table.shape, generator=gen).cuda()
    y0, _, g0 = run(False)
    with torch.no_grad():
        table.add_(direction)
    ya, _, _ = run(False)
    with torch.no_grad():
        table.sub_(direction)
    fd = ((ya - y0).double() * w.double()).sum()
    an = (g0['ffd.{}.lora.right.weight'.format('fc2' if c['family'] == 'opt' else 'down')].double()
          * direction.double()).sum()
    assert abs(fd - an) <= 2e-3 * abs(an) + 1e-2, (float(fd), float(an))
