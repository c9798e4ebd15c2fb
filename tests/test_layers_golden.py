"""This repo's `naive_gpt.layers` / `naive_gpt.utils` against goldens produced by the
IMPORTED REFERENCE (tests/golden/make_golden.py, build container).

CPU tier: pure-PyTorch layers run as they are; layers that reach `naive_gpt.ext` get
the CPU oracle patched in *by the test* (the product never imports it), which checks
the host logic -- autograd wiring, layouts, orchestration -- against the reference's.
GPU tier (`-m gpu`): the same goldens with the real HIP extension on cuda:0.
"""
import os

import numpy as np
import pytest
import torch
from torch import nn

from oracle import ext_stub

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')


def load(name):
    return dict(np.load(os.path.join(GOLD, name + '.npz'), allow_pickle=False))


def T(a, device='cpu'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def load_sd(module, g, prefix, device='cpu'):
    sd = {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}
    missing = module.load_state_dict(sd, strict=False)
    # only the constant rotary tables may be absent from the fixture
    assert all('cached' in k for k in missing.missing_keys), missing
    assert not missing.unexpected_keys, missing
    return module.to(device)


def check(got: torch.Tensor, want: np.ndarray, rtol, atol, what=''):
    got = got.detach().cpu().numpy()
    assert got.shape == want.shape, what
    err = np.abs(got - want)
    assert (err <= atol + rtol * np.abs(want)).all(), '{}: max err {}'.format(what, err.max())


def check_grads(module, g, prefix, rtol, atol):
    n = 0
    for name, p in module.named_parameters():
        key = prefix + name
        if key in g:
            assert p.grad is not None, name
            check(p.grad, g[key], rtol, atol, name)
            n += 1
        else:
            assert p.grad is None, name
    assert n > 0


# ------------------------------------------------------------------ routed FFN

def build_ffn(tag):
    from naive_gpt import layers
    d_model, d_ff, bs = 16, 64, 16
    if tag == 'routed':
        return layers.RoutedFFN(d_model=d_model, d_feedforward=d_ff, block_size=bs,
                                activation=nn.ReLU())
    if tag == 'routed_llama':
        return layers.RoutedLLaMaFFN(d_model=d_model, d_feedforward=d_ff, block_size=bs // 2,
                                     activation=nn.SiLU())
    if tag == 'lora_routed':
        return layers.LoRARoutedFFN(d_lora=4, block_size=bs, d_model=d_model,
                                    d_feedforward=d_ff, activation=nn.ReLU())
    return layers.LoRARoutedLLaMaFFN(d_lora=4, block_size=bs, d_model=d_model,
                                     d_feedforward=d_ff, activation=nn.SiLU())


FFN_TAGS = ['routed', 'routed_llama', 'lora_routed', 'lora_routed_llama']


def run_ffn(tag, device, rtol, atol):
    g = load('ffn')
    model = load_sd(build_ffn(tag), g, tag + '.sd.', device)
    x = T(g['x'], device).requires_grad_(True)
    y = model(x)
    (y * T(g['w'], device)).sum().backward()
    check(y, g[tag + '.y'], rtol, atol, 'y')
    check(x.grad, g[tag + '.grad_x'], rtol, atol, 'grad_x')
    check_grads(model, g, tag + '.grad.', rtol, atol)


@pytest.mark.parametrize('tag', FFN_TAGS)
def test_routed_ffn_matches_reference_cpu(tag):
    run_ffn(tag, 'cpu', 1e-5, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', FFN_TAGS)
def test_routed_ffn_matches_reference_gpu(tag):
    run_ffn(tag, 'cuda', 1e-3, 1e-4)


def test_routed_ffn_is_the_masked_dense_ffn():
    """The reference's own identity (test/layer/test_sparse_ffn.py:9-38 with k matched):
    RoutedFFN == fc2(act(mask * fc1(x)))."""
    from naive_gpt import layers
    torch.manual_seed(0)
    ffn = layers.RoutedFFN(d_model=8, d_feedforward=64, block_size=4, activation=nn.ReLU())
    x = torch.randn([4, 16, 8])
    xf = x.view(-1, 8)
    idx = torch.topk(ffn.router(xf), k=ffn.n_blocks // 2, dim=-1).indices
    mask = torch.zeros([xf.size(0), ffn.n_blocks]).scatter(-1, idx, 1.0)
    mask = mask.repeat_interleave(ffn.block_size, dim=-1)
    want = ffn.fc2(ffn.activation(mask * ffn.fc1(xf))).view_as(x)
    assert torch.allclose(ffn(x), want, atol=1e-5)


# ------------------------------------------------------------------ PQ

def run_pq(tag, device, exact_float):
    from naive_gpt import layers
    g = load('pq')
    cls = layers.PQV1 if tag == 'v1' else layers.PQV2
    pq = cls(d_codeword=8, n_codewords=16, n_subspaces=4)
    pq.weight.data.copy_(T(g[tag + '.weight']))
    pq = pq.to(device)
    z = T(g[tag + '.z'], device).requires_grad_(True)
    codes = pq('encode', z=z)
    assert codes.dtype == (torch.int64 if tag == 'v1' else torch.int32)
    assert np.array_equal(codes.cpu().numpy(), g[tag + '.codes'])        # bit-exact
    z_q, loss = pq('train', z=z)
    loss.backward()
    rtol, atol = (1e-6, 1e-6) if exact_float else (1e-3, 1e-5)
    check(z_q, g[tag + '.z_q'], 0, 0, 'z_q')                             # a gather: exact
    check(loss, g[tag + '.loss'], rtol, atol, 'loss')
    check(pq('quantize', z=z.detach()), g[tag + '.quantize'], 0, 0)
    check(pq('decode', z=codes), g[tag + '.decode'], 0, 0)
    check(z.grad, g[tag + '.grad_z'], rtol, atol, 'grad_z')
    check(pq.weight.grad, g[tag + '.grad_weight'], rtol, 1e-5, 'grad_weight')


def test_pq_v1_matches_reference_cpu():
    run_pq('v1', 'cpu', True)


def test_pq_v2_matches_reference_cpu(oracle_ext):
    run_pq('v2', 'cpu', True)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', ['v1', 'v2'])
def test_pq_matches_reference_gpu(tag):
    run_pq(tag, 'cuda', False)


# ------------------------------------------------------------------ attention

def build_attn(tag):
    from naive_gpt import layers
    E = 64
    if tag == 'vanilla':
        return layers.VanillaAttention(d_head=E, p_dropout=0.0)
    if tag == 'rotary':
        return layers.RotaryAttention(d_head=E, p_dropout=0.0, max_length=256)
    if tag == 'sparse_rotary_v2':
        return layers.SparseRotaryAttentionV2(d_head=E, p_dropout=0.0, d_codeword=8,
                                              n_codewords=16)
    return layers.SparseVanillaAttentionV2(d_head=E, d_codeword=8, n_codewords=16,
                                           p_dropout=0.0)


def run_attn(tag, device, rtol, atol):
    g = load('attention')
    model = load_sd(build_attn(tag), g, tag + '.sd.', device)
    q, k, v = [T(g[n], device).requires_grad_(True) for n in 'qkv']
    S = q.size(1)
    mask = None
    if tag in ('vanilla', 'rotary'):
        mask = torch.full([S, S], float('-inf'), device=device).triu(1)
    train = tag.endswith('_train')
    if train:
        model.trigger.fill_(True)
    y = model(q, k, v, attn_mask=mask)
    loss = (y * T(g['w'], device)).sum()
    if train:
        loss = loss + 1e-2 * model.loss
        check(model.loss, g[tag + '.pq_loss'], rtol, atol, 'pq_loss')
        assert bool(model.trigger.item()) == bool(g[tag + '.trigger_after'])
    loss.backward()
    check(y, g[tag + '.y'], rtol, atol, 'y')
    check(q.grad, g[tag + '.grad_q'], rtol, atol, 'grad_q')
    check(k.grad, g[tag + '.grad_k'], rtol, atol, 'grad_k')
    check(v.grad, g[tag + '.grad_v'], rtol, atol, 'grad_v')
    if train:
        check_grads(model, g, tag + '.grad.', rtol, 1e-5)


DENSE_TAGS = ['vanilla', 'rotary']
SPARSE_TAGS = ['sparse_v2', 'sparse_v2_train', 'sparse_rotary_v2']


@pytest.mark.parametrize('tag', DENSE_TAGS)
def test_dense_attention_matches_reference_cpu(tag):
    run_attn(tag, 'cpu', 1e-5, 1e-5)


@pytest.mark.parametrize('tag', SPARSE_TAGS)
def test_sparse_attention_matches_reference_cpu(tag, oracle_ext):
    run_attn(tag, 'cpu', 1e-5, 2e-5)


def test_sparse_attention_csr_structure_cpu(oracle_ext):
    g = load('attention')
    model = load_sd(build_attn('sparse_v2'), g, 'sparse_v2.sd.')
    indptr, indices, values = model._get_attn(T(g['q']), T(g['k']), attn_mask=None)
    assert np.array_equal(indptr.numpy(), g['sparse_v2.indptr'])
    assert np.array_equal(indices.numpy(), g['sparse_v2.indices'])
    check(values, g['sparse_v2.values'], 1e-5, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('tag', DENSE_TAGS + SPARSE_TAGS)
def test_attention_matches_reference_gpu(tag):
    run_attn(tag, 'cuda', 1e-3, 2e-4)


@pytest.mark.gpu
def test_sparse_attention_csr_structure_gpu():
    g = load('attention')
    model = load_sd(build_attn('sparse_v2'), g, 'sparse_v2.sd.', 'cuda')
    indptr, indices, values = model._get_attn(T(g['q'], 'cuda'), T(g['k'], 'cuda'),
                                              attn_mask=None)
    assert np.array_equal(indptr.cpu().numpy(), g['sparse_v2.indptr'])
    assert np.array_equal(indices.cpu().numpy(), g['sparse_v2.indices'])     # bit-exact
    check(values, g['sparse_v2.values'], 1e-3, 1e-6)


def test_all_ones_plumbing_check_of_the_reference(oracle_ext):
    """test/layer/test_sparse_mha.py:7-43: q = k = v = 1 -> dense and sparse agree."""
    from naive_gpt import layers
    q = torch.ones([1, 128, 2, 32])
    dense = layers.VanillaAttention(d_head=32, p_dropout=0.0)
    sparse = layers.SparseVanillaAttentionV2(d_head=32, d_codeword=8, n_codewords=16,
                                             p_dropout=0.0)
    assert torch.allclose(dense(q, q, q), sparse(q, q, q), atol=1e-3)


# ------------------------------------------------------------------ upgraded block

def build_block(kind):
    from naive_gpt import layers
    d_model, n_heads, d_ff = 64, 1, 128
    if kind == 'opt':
        return layers.TransformerBlock(
            d_model=d_model, n_heads=n_heads, layernorm_fn=nn.LayerNorm(d_model),
            attention_fn=layers.VanillaAttention(d_head=d_model // n_heads, p_dropout=0.0),
            feedforward_fn=layers.Feedforward(d_model=d_model, d_feedforward=d_ff,
                                              activation=nn.ReLU(), p_dropout=0.0),
            attention_bias=True, pre_norm=True)
    return layers.TransformerBlock(
        d_model=d_model, n_heads=n_heads, layernorm_fn=layers.LlamaRMSNorm(d_model),
        attention_fn=layers.RotaryAttention(d_head=d_model // n_heads, p_dropout=0.0,
                                            max_length=256),
        feedforward_fn=layers.LLaMaFeedforward(d_model=d_model, d_feedforward=d_ff,
                                               activation=nn.SiLU()),
        attention_bias=False, pre_norm=True)


def upgrade(model, capsys=None):
    from naive_gpt import utils
    for stage in ['lora', 'ffn', 'mha_v1', 'mha_v2']:
        model = utils.ModuleUpgrader(
            handler=utils.SparseLoRAHandler(d_lora=4, stage=stage)).visit(model)
    return model


def run_block(kind, device, rtol, atol):
    g = load('block')
    prefix = kind + '.sparse.sd.'
    # dense block first: its weights are the frozen subset of the upgraded state_dict
    dense = build_block(kind)
    dense_keys = set(dense.state_dict())
    sd = {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}
    dense.load_state_dict({k: v for k, v in sd.items() if k in dense_keys}, strict=False)
    S = g['x'].shape[1]
    x = T(g['x'], device).requires_grad_(True)
    mask = torch.full([S, S], float('-inf'), device=device).triu(1)
    y = dense.to(device)(x, attn_mask=mask)
    (y * T(g['w'], device)).sum().backward()
    check(y, g[kind + '.dense.y'], rtol, atol, 'dense y')
    check(x.grad, g[kind + '.dense.grad_x'], rtol, atol, 'dense grad_x')

    model = upgrade(dense.cpu())
    trainable = sorted(n for n, p in model.named_parameters() if p.requires_grad)
    assert trainable == list(g[kind + '.sparse.trainable'])
    assert set(model.state_dict()) - {k for k in model.state_dict() if 'cached' in k} \
        == set(sd)                                        # identical state_dict keys
    model = load_sd(model, g, prefix, device)
    model.zero_grad()
    x = T(g['x'], device).requires_grad_(True)
    y = model(x)
    (y * T(g['w'], device)).sum().backward()
    check(y, g[kind + '.sparse.y'], rtol, atol, 'sparse y')
    check(x.grad, g[kind + '.sparse.grad_x'], rtol, atol * 10, 'sparse grad_x')
    check_grads(model, g, kind + '.sparse.grad.', rtol, atol * 10)


@pytest.mark.parametrize('kind', ['opt', 'llama'])
def test_upgraded_block_matches_reference_cpu(kind, oracle_ext, capsys):
    run_block(kind, 'cpu', 1e-4, 2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['opt', 'llama'])
def test_upgraded_block_matches_reference_gpu(kind):
    run_block(kind, 'cuda', 2e-3, 5e-4)


# ------------------------------------------------------------------ N * H >= 32 (the matrix-core path)
# tests/golden/attention_mfma.npz, block_mfma.npz: the imported reference's orchestration at
# the shape class bench.py and real training run (n * h >= 32, d_head 64 and 128), where this
# repo's V2 layers take `_MfmaAttention`.  Inputs / weights are seeded by name (helpers.seeded).

from helpers import fingerprint, seeded, seeded_fill  # noqa: E402

MFMA_CASES = {
    'v64': dict(kind='vanilla', N=2, H=16, S=256, E=64, heads=[3, 12]),
    'r128': dict(kind='rotary', N=1, H=32, S=256, E=128, heads=[7, 29]),
}


class _Spy:
    """Counts the calls of the matrix-core attention entry points."""

    def __init__(self, monkeypatch):
        from naive_gpt import ext
        self.calls = {'attention_mfma_forward': 0, 'attention_mfma_backward': 0}
        for name in self.calls:
            monkeypatch.setattr(ext, name, self._wrap(name, getattr(ext, name)))

    def _wrap(self, name, fn):
        def counted(*a, **kw):
            self.calls[name] += 1
            return fn(*a, **kw)
        return counted


def _slice_sums(t):
    return t.detach().double().sum(dim=(1, 3)).cpu().numpy(), \
        t.detach().double().abs().sum(dim=(1, 3)).cpu().numpy()


def run_attn_mfma(tag, variant, device, rtol, atol):
    from naive_gpt import layers
    g, c = load('attention_mfma'), MFMA_CASES[tag]
    N, H, S, E, heads = c['N'], c['H'], c['S'], c['E'], c['heads']
    arrays = {n: seeded('attn_mfma.{}.{}'.format(tag, n), [N, S, H, E]) for n in 'qkvw'}
    for n, a in arrays.items():
        assert np.allclose(fingerprint(a), g['{}.print.{}'.format(tag, n)], rtol=1e-12), \
            'the seeded generator drifted: regenerate tests/golden with make_golden.py'
    if c['kind'] == 'vanilla':
        model = layers.SparseVanillaAttentionV2(d_head=E, d_codeword=8, n_codewords=16,
                                                p_dropout=0.0)
    else:
        model = layers.SparseRotaryAttentionV2(d_head=E, p_dropout=0.0, d_codeword=8,
                                               n_codewords=16)
    model = load_sd(model, g, tag + '.sd.', device)
    q, k, v = [T(arrays[n], device).requires_grad_(True) for n in 'qkv']
    if variant == 'train':
        model.trigger.fill_(True)
    y = model(q, k, v, attn_mask=None)
    loss = (y * T(arrays['w'], device)).sum()
    key = '{}.{}.'.format(tag, variant)
    if variant == 'train':
        loss = loss + 1e-2 * model.loss
        check(model.loss, g[key + 'pq_loss'], rtol, atol, 'pq_loss')
    loss.backward()
    for name, t in (('y', y), ('grad_q', q.grad), ('grad_k', k.grad), ('grad_v', v.grad)):
        if variant == 'plain':
            check(t[:, :, heads], g[key + name + '.heads'], rtol, atol, name)
        sums, abs_sums = _slice_sums(t)
        # every (n, h) slice: the sum of S * E terms against the same sum of the reference
        want, scale = g[key + name + '.sums'], g[key + name + '.abs']
        assert (np.abs(sums - want) <= rtol * scale).all(), (name, np.abs(sums - want).max())
        assert np.allclose(abs_sums, scale, rtol=rtol), name
    if variant == 'train':
        check_grads(model, g, key + 'grad.', rtol, 1e-5)


def _mfma_structure(tag, device):
    """CSR indices of the layer's lookup against the reference's, bit for bit."""
    from naive_gpt import layers
    g, c = load('attention_mfma'), MFMA_CASES[tag]
    N, H, S, E, heads = c['N'], c['H'], c['S'], c['E'], c['heads']
    q, k = [T(seeded('attn_mfma.{}.{}'.format(tag, n), [N, S, H, E]), device) for n in 'qk']
    if c['kind'] == 'vanilla':
        model = layers.SparseVanillaAttentionV2(d_head=E, d_codeword=8, n_codewords=16,
                                                p_dropout=0.0)
    else:
        model = layers.SparseRotaryAttentionV2(d_head=E, p_dropout=0.0, d_codeword=8,
                                               n_codewords=16)
    model = load_sd(model, g, tag + '.sd.', device)
    attn = model._get_attn(q, k, None)
    indices = attn[2] if isinstance(attn[0], str) else attn[1]
    idx = indices.cpu().numpy().reshape(N, H, -1)
    assert np.array_equal(idx[:, heads], g[tag + '.indices.heads'])
    weights = (np.arange(idx.shape[-1], dtype=np.int64) % 8191) + 1
    assert np.array_equal((idx.astype(np.int64) * weights).sum(-1), g[tag + '.indices.checksum'])
    return attn


@pytest.mark.parametrize('variant', ['plain', 'train'])
@pytest.mark.parametrize('tag', list(MFMA_CASES))
def test_attention_heads32_matches_reference_cpu(tag, variant, oracle_ext):
    run_attn_mfma(tag, variant, 'cpu', 1e-4, 2e-5)


@pytest.mark.parametrize('tag', list(MFMA_CASES))
def test_attention_heads32_csr_structure_cpu(tag, oracle_ext):
    _mfma_structure(tag, 'cpu')


@pytest.mark.gpu
@pytest.mark.parametrize('variant', ['plain', 'train'])
@pytest.mark.parametrize('tag', list(MFMA_CASES))
def test_attention_heads32_matches_reference_gpu(tag, variant, monkeypatch):
    """The matrix-core kernels (d_head 64 and the spt::e128 build) directly against the
    imported reference's orchestration -- and the test insists that they ran."""
    spy = _Spy(monkeypatch)
    run_attn_mfma(tag, variant, 'cuda', 1e-3, 2e-4)
    assert spy.calls == {'attention_mfma_forward': 1, 'attention_mfma_backward': 1}


@pytest.mark.gpu
@pytest.mark.parametrize('tag', list(MFMA_CASES))
def test_attention_heads32_csr_structure_gpu(tag):
    attn = _mfma_structure(tag, 'cuda')
    assert attn[0] == 'mfma'


BLOCK_MFMA = {
    'opt': dict(d_model=1024, n_heads=16, d_ff=256, N=2, S=128),
    'llama': dict(d_model=2048, n_heads=16, d_ff=512, N=2, S=128),
}


def run_block_mfma(kind, device, rtol, atol):
    from naive_gpt import layers
    g, c = load('block_mfma'), BLOCK_MFMA[kind]
    d_model, n_heads, d_ff, N, S = c['d_model'], c['n_heads'], c['d_ff'], c['N'], c['S']
    if kind == 'opt':
        block = layers.TransformerBlock(
            d_model=d_model, n_heads=n_heads, layernorm_fn=nn.LayerNorm(d_model),
            attention_fn=layers.VanillaAttention(d_head=d_model // n_heads, p_dropout=0.0),
            feedforward_fn=layers.Feedforward(d_model=d_model, d_feedforward=d_ff,
                                              activation=nn.ReLU(), p_dropout=0.0),
            attention_bias=True, pre_norm=True)
    else:
        block = layers.TransformerBlock(
            d_model=d_model, n_heads=n_heads, layernorm_fn=layers.LlamaRMSNorm(d_model),
            attention_fn=layers.RotaryAttention(d_head=d_model // n_heads, p_dropout=0.0,
                                                max_length=256),
            feedforward_fn=layers.LLaMaFeedforward(d_model=d_model, d_feedforward=d_ff,
                                                   activation=nn.SiLU()),
            attention_bias=False, pre_norm=True)
    model = upgrade(block)
    tag = 'block_mfma.{}.'.format(kind)
    prints = seeded_fill(model, tag)
    wanted = {k[len(kind + '.print.'):]: v for k, v in g.items() if k.startswith(kind + '.print.')}
    assert set(prints) == set(wanted)                     # identical state_dict keys
    for key, fp in prints.items():
        assert np.allclose(fp, wanted[key], rtol=1e-12), 'seeded generator drifted: ' + key
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad) \
        == list(g[kind + '.trainable'])
    model = model.to(device)
    x = T(seeded(tag + 'x', [N, S, d_model]), device).requires_grad_(True)
    w = T(seeded(tag + 'w', [N, S, d_model]), device)
    y = model(x)
    (y * w).sum().backward()
    for name, t in (('y', y), ('grad_x', x.grad)):
        want = g['{}.{}.sub'.format(kind, name)]
        scale = float(np.abs(want).max())
        check(t[:, ::4], want, rtol, atol * scale, name)
        sums = t.detach().double().sum(-1).cpu().numpy()
        assert (np.abs(sums - g['{}.{}.sums'.format(kind, name)])
                <= rtol * d_model ** 0.5 * scale * 4).all(), name
    n = 0
    for name, p in model.named_parameters():
        key = kind + '.grad.' + name
        if key in g:
            scale = max(float(np.abs(g[key]).max()), 1e-6)
            check(p.grad, g[key], rtol, atol * scale, name)
            n += 1
        else:
            assert p.grad is None, name
    assert n >= 10


@pytest.mark.parametrize('kind', list(BLOCK_MFMA))
def test_upgraded_block_heads32_matches_reference_cpu(kind, oracle_ext, capsys):
    run_block_mfma(kind, 'cpu', 1e-4, 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', list(BLOCK_MFMA))
def test_upgraded_block_heads32_matches_reference_gpu(kind, monkeypatch, capsys):
    """Whole upgraded blocks (split-bf16 LoRA linears, routed FFN, matrix-core attention)
    at BERT-large / LLaMA head shapes against the imported reference."""
    spy = _Spy(monkeypatch)
    run_block_mfma(kind, 'cuda', 2e-3, 1e-3)
    assert spy.calls == {'attention_mfma_forward': 1, 'attention_mfma_backward': 1}


# ------------------------------------------------------------------ sparse chain == dense attention
# tests/golden/dense_chain.npz: the reference's dense VanillaAttention (causal mask) on small
# inputs.  On the full causal pattern (every row lists all S columns, softmax masks col > row)
# with scores inside the clamp the sparse chain sddmm -> scale, clamp -> softmax -> spmm and
# its autograd wiring (kernels/{sddmm,softmax,spmm}.py) compute exactly that function: this is
# the reference-run pin of the oracle's sddmm / softmax / spmm legs.

def _dense_chain(device):
    from naive_gpt import kernels
    g = load('dense_chain')
    q, k, v = [T(g[n], device) for n in 'qkv']
    N, S, H, E = q.shape
    flat = lambda t: t.transpose(1, 2).contiguous().view(N * H, S, E).requires_grad_(True)   # noqa: E731
    qf, kf, vf = flat(q), flat(k), flat(v)
    indptr = torch.arange(0, S * S + 1, S, dtype=torch.int32, device=device)
    indices = torch.arange(S, dtype=torch.int32, device=device).repeat(N * H, S).contiguous()
    scores = kernels.sddmm(indptr, indices, query=qf, key=kf)
    scores = torch.clamp(E ** -0.5 * scores, min=-10.0, max=10.0)
    assert float(scores.detach().abs().max()) < 10.0
    attn = kernels.softmax(indptr, indices, values=scores)
    y = kernels.spmm(indptr, indices, attn, vf)
    heads = lambda t: t.view(N, H, S, E).transpose(1, 2)                    # noqa: E731
    (heads(y) * T(g['w'], device)).sum().backward()
    return g, heads(y), heads(qf.grad), heads(kf.grad), heads(vf.grad)


def test_oracle_sparse_chain_equals_reference_dense_attention(oracle_ext):
    g, y, gq, gk, gv = _dense_chain('cpu')
    for got, name in ((y, 'y'), (gq, 'grad_q'), (gk, 'grad_k'), (gv, 'grad_v')):
        check(got, g[name], 1e-5, 2e-6, name)


@pytest.mark.gpu
def test_hip_sparse_chain_equals_reference_dense_attention():
    g, y, gq, gk, gv = _dense_chain('cuda')
    for got, name in ((y, 'y'), (gq, 'grad_q'), (gk, 'grad_k'), (gv, 'grad_v')):
        check(got, g[name], 1e-3, 1e-5, name)
