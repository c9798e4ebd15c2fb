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
