"""Generate golden vectors by IMPORTING THE REFERENCE (build container only).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_golden.py

The reference's pure-PyTorch layers run as they are (CPU).  Its native module
`naive_gpt.ext` cannot be built here (CUDA + cuSPARSE), so for the layers that reach
it (`PQV2`, the V2 sparse attentions) the CPU oracle (`oracle/ext_stub.py`) is
installed as `naive_gpt.ext`: the goldens of those layers then pin the reference's
*orchestration* (`naive_gpt/layers/sparse/attention.py:84-142`, `quantizer.py`,
`kernels/*.py` autograd wiring) on top of the oracle's operator semantics.
`naive_gpt.loaders` needs torchdata/torchtext/lightning (absent) and is stubbed empty.

Outputs: tests/golden/*.npz -- inputs, state_dicts, outputs, gradients (data only;
no reference source is copied).  Nothing here runs on the GPU box.
"""
import os
import sys
import types

import numpy as np
import torch
from torch import nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REFERENCE = '/root/reference'

sys.dont_write_bytecode = True
sys.path.insert(0, ROOT)
from oracle import ext_stub  # noqa: E402

sys.modules['naive_gpt.ext'] = ext_stub
sys.modules['naive_gpt.loaders'] = types.ModuleType('naive_gpt.loaders')
sys.path.insert(0, REFERENCE)
import naive_gpt  # noqa: E402
from naive_gpt import layers, utils  # noqa: E402

assert naive_gpt.__file__.startswith(REFERENCE)
naive_gpt.ext = ext_stub


class _NoStream:
    """The reference's sddmm/spmm backward forks a CUDA stream around its second
    SpMM (kernels/sddmm.py:36-50).  That is scheduling, not arithmetic: on this
    GPU-less host the three stream entry points become no-ops."""

    def __init__(self, *args, **kwargs):
        pass

    def wait_stream(self, other):
        pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        return False


torch.cuda.current_stream = lambda *a, **k: _NoStream()
torch.cuda.Stream = _NoStream
torch.cuda.stream = lambda s: s


def t2n(t):
    # a copy: golden_models() steps the optimiser in place after taking its snapshots
    return t.detach().cpu().numpy().copy()


def sd2n(module, prefix='sd.'):
    return {prefix + k: t2n(v) for k, v in module.state_dict().items()}


def grads2n(module, prefix='grad.'):
    return {prefix + k: t2n(p.grad) for k, p in module.named_parameters()
            if p.grad is not None}


def save(name, **arrays):
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **arrays)
    print('wrote', path, '{:.1f} KB'.format(os.path.getsize(path) / 1024))


def quiet_upgrade(model, d_lora):
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        for stage in ['lora', 'ffn', 'mha_v1', 'mha_v2']:
            upgrader = utils.ModuleUpgrader(
                handler=utils.SparseLoRAHandler(d_lora=d_lora, stage=stage)
            )
            model = upgrader.visit(model)
    return model


def randomize_lora(model, gen):
    # LoRA `right` tables start at zero (lora.py:25-26): make the side path count
    for name, p in model.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.copy_(0.05 * torch.randn(p.shape, generator=gen))


# --------------------------------------------------------------------------- FFN

def golden_ffn():
    gen = torch.Generator().manual_seed(1234)
    d_model, d_ff, bs = 16, 64, 16
    x = torch.randn([2, 24, d_model], generator=gen)
    w = torch.randn([2, 24, d_model], generator=gen)
    out = {'x': t2n(x), 'w': t2n(w)}

    def run(tag, model):
        xi = x.clone().requires_grad_(True)
        y = model(xi)
        (y * w).sum().backward()
        out.update({tag + '.y': t2n(y), tag + '.grad_x': t2n(xi.grad)})
        out.update(sd2n(model, tag + '.sd.'))
        out.update(grads2n(model, tag + '.grad.'))

    torch.manual_seed(1)
    run('routed', layers.RoutedFFN(d_model=d_model, d_feedforward=d_ff,
                                   block_size=bs, activation=nn.ReLU()))
    torch.manual_seed(2)
    run('routed_llama', layers.RoutedLLaMaFFN(d_model=d_model, d_feedforward=d_ff,
                                              block_size=bs // 2, activation=nn.SiLU()))
    torch.manual_seed(3)
    m = layers.LoRARoutedFFN(d_lora=4, block_size=bs, d_model=d_model,
                             d_feedforward=d_ff, activation=nn.ReLU())
    randomize_lora(m, gen)
    run('lora_routed', m)
    torch.manual_seed(4)
    m = layers.LoRARoutedLLaMaFFN(d_lora=4, block_size=bs, d_model=d_model,
                                  d_feedforward=d_ff, activation=nn.SiLU())
    randomize_lora(m, gen)
    run('lora_routed_llama', m)
    save('ffn', **out)


# --------------------------------------------------------------------------- PQ

def golden_pq():
    gen = torch.Generator().manual_seed(99)
    out = {}
    for tag, cls in [('v1', layers.PQV1), ('v2', layers.PQV2)]:
        torch.manual_seed(5)
        pq = cls(d_codeword=8, n_codewords=16, n_subspaces=4)
        z = torch.randn([3, 16, 32], generator=torch.Generator().manual_seed(7))
        zi = z.clone().requires_grad_(True)
        codes = pq('encode', z=zi)
        z_q, loss = pq('train', z=zi)
        loss.backward()
        out.update({tag + '.z': t2n(z), tag + '.codes': t2n(codes),
                    tag + '.z_q': t2n(z_q), tag + '.loss': t2n(loss),
                    tag + '.quantize': t2n(pq('quantize', z=z)),
                    tag + '.decode': t2n(pq('decode', z=codes)),
                    tag + '.grad_z': t2n(zi.grad),
                    tag + '.grad_weight': t2n(pq.weight.grad),
                    tag + '.weight': t2n(pq.weight)})
    # the reference's two quantiser methods agree on codes (same L1 + argmin)
    assert np.array_equal(out['v1.codes'], out['v2.codes'])
    save('pq', **out)


# --------------------------------------------------------------------------- attention

def golden_attention():
    out = {}
    N, S, H, E = 1, 128, 1, 64
    gen = torch.Generator().manual_seed(2024)
    q = torch.randn([N, S, H, E], generator=gen)
    k = torch.randn([N, S, H, E], generator=gen)
    v = torch.randn([N, S, H, E], generator=gen)
    w = torch.randn([N, S, H, E], generator=gen)
    out.update({'q': t2n(q), 'k': t2n(k), 'v': t2n(v), 'w': t2n(w)})

    def run(tag, model, trigger=False, mask=None):
        qi, ki, vi = [t.clone().requires_grad_(True) for t in (q, k, v)]
        if trigger:
            model.trigger.fill_(True)
        y = model(qi, ki, vi, attn_mask=mask)
        loss = (y * w).sum()
        if trigger:
            loss = loss + 1e-2 * model.loss
            out[tag + '.pq_loss'] = t2n(model.loss)
            out[tag + '.trigger_after'] = t2n(model.trigger)
        loss.backward()
        out.update({tag + '.y': t2n(y), tag + '.grad_q': t2n(qi.grad),
                    tag + '.grad_k': t2n(ki.grad), tag + '.grad_v': t2n(vi.grad)})
        out.update(sd2n(model, tag + '.sd.'))
        out.update(grads2n(model, tag + '.grad.'))

    causal = torch.full([S, S], float('-inf')).triu(1)
    run('vanilla', layers.VanillaAttention(d_head=E, p_dropout=0.0), mask=causal)
    run('rotary', layers.RotaryAttention(d_head=E, p_dropout=0.0, max_length=256), mask=causal)
    torch.manual_seed(11)
    run('sparse_v2', layers.SparseVanillaAttentionV2(
        d_head=E, d_codeword=8, n_codewords=16, p_dropout=0.0))
    torch.manual_seed(12)
    run('sparse_v2_train', layers.SparseVanillaAttentionV2(
        d_head=E, d_codeword=8, n_codewords=16, p_dropout=0.0), trigger=True)
    torch.manual_seed(13)
    m = layers.SparseRotaryAttentionV2(d_head=E, p_dropout=0.0, d_codeword=8, n_codewords=16)
    run('sparse_rotary_v2', m)
    # the CSR structure the V2 layer built (bit-exact contract)
    torch.manual_seed(11)
    m = layers.SparseVanillaAttentionV2(d_head=E, d_codeword=8, n_codewords=16, p_dropout=0.0)
    indptr, indices, values = m._get_attn(q, k, attn_mask=None)
    out.update({'sparse_v2.indptr': t2n(indptr), 'sparse_v2.indices': t2n(indices),
                'sparse_v2.values': t2n(values)})
    # drop the big constant buffers of the rotary tables from the fixture
    for key in list(out):
        if 'cos_cached' in key or 'sin_cached' in key or 'cached_ids' in key:
            del out[key]
    save('attention', **out)


# --------------------------------------------------------------------------- block

def golden_block():
    out = {}
    d_model, n_heads, d_ff, S, N = 64, 1, 128, 128, 1

    def build(kind):
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

    gen = torch.Generator().manual_seed(77)
    x = torch.randn([N, S, d_model], generator=gen)
    w = torch.randn([N, S, d_model], generator=gen)
    out.update({'x': t2n(x), 'w': t2n(w)})
    causal = torch.full([S, S], float('-inf')).triu(1)
    for kind in ['opt', 'llama']:
        torch.manual_seed(21 if kind == 'opt' else 22)
        dense = build(kind)
        # (the dense weights are a subset of the upgraded state_dict stored below)
        xi = x.clone().requires_grad_(True)
        y = dense(xi, attn_mask=causal)
        (y * w).sum().backward()
        out.update({kind + '.dense.y': t2n(y), kind + '.dense.grad_x': t2n(xi.grad)})

        model = quiet_upgrade(dense, d_lora=4)
        randomize_lora(model, gen)
        model.zero_grad()
        out.update(sd2n(model, kind + '.sparse.sd.'))
        out[kind + '.sparse.trainable'] = np.array(
            sorted(n for n, p in model.named_parameters() if p.requires_grad))
        xi = x.clone().requires_grad_(True)
        y = model(xi)
        (y * w).sum().backward()
        out.update({kind + '.sparse.y': t2n(y), kind + '.sparse.grad_x': t2n(xi.grad)})
        out.update(grads2n(model, kind + '.sparse.grad.'))
    for key in list(out):
        if 'cos_cached' in key or 'sin_cached' in key or 'cached_ids' in key:
            del out[key]
    save('block', **out)


# --------------------------------------------------------------------------- N * H >= 32

sys.path.insert(0, os.path.join(ROOT, 'tests'))
from helpers import fingerprint, seeded, seeded_fill  # noqa: E402

# the shape class of the benchmark path: this repo's V2 layers run their matrix-core
# kernels from n * h >= 32 (layers/sparse/attention.py: _mfma_ok); the N * H = 1 goldens
# above take the separate-operator path
MFMA_CASES = {
    'v64': dict(kind='vanilla', N=2, H=16, S=256, E=64, heads=[3, 12]),
    'r128': dict(kind='rotary', N=1, H=32, S=256, E=128, heads=[7, 29]),
}


def golden_attention_mfma():
    """The reference's V2 attentions (orchestration attention.py:84-142 / :241-299 on the
    oracle) at N * H = 32.  Inputs are seeded by name (tests/helpers.py: seeded); stored:
    the outputs and gradients of two heads in full, per-(n, h) sums of all of them, the CSR
    indices of those heads and a position-weighted checksum of every slice."""
    out = {}
    for tag, c in MFMA_CASES.items():
        N, H, S, E, heads = c['N'], c['H'], c['S'], c['E'], c['heads']
        q, k, v, w = [torch.from_numpy(seeded('attn_mfma.{}.{}'.format(tag, n), [N, S, H, E]))
                      for n in 'qkvw']
        for n, t in zip('qkvw', (q, k, v, w)):
            out['{}.print.{}'.format(tag, n)] = fingerprint(t.numpy())

        def build():
            torch.manual_seed(41)
            if c['kind'] == 'vanilla':
                return layers.SparseVanillaAttentionV2(d_head=E, d_codeword=8, n_codewords=16,
                                                       p_dropout=0.0)
            return layers.SparseRotaryAttentionV2(d_head=E, p_dropout=0.0, d_codeword=8,
                                                  n_codewords=16)

        for variant in ('plain', 'train'):
            model = build()
            key = '{}.{}.'.format(tag, variant)
            qi, ki, vi = [t.clone().requires_grad_(True) for t in (q, k, v)]
            if variant == 'train':
                model.trigger.fill_(True)
            y = model(qi, ki, vi, attn_mask=None)
            loss = (y * w).sum()
            if variant == 'train':
                loss = loss + 1e-2 * model.loss
                out[key + 'pq_loss'] = t2n(model.loss)
            loss.backward()
            for name, t in (('y', y), ('grad_q', qi.grad), ('grad_k', ki.grad),
                            ('grad_v', vi.grad)):
                if variant == 'plain':
                    out[key + name + '.heads'] = t2n(t[:, :, heads])
                out[key + name + '.sums'] = t2n(t.double().sum(dim=(1, 3)))          # [N, H]
                out[key + name + '.abs'] = t2n(t.double().abs().sum(dim=(1, 3)))
            out.update(grads2n(model, key + 'grad.'))
            if variant == 'plain':
                out.update(sd2n(model, tag + '.sd.'))
                indptr, indices, values = model._get_attn(q, k, attn_mask=None)
                idx = t2n(indices).reshape(N, H, -1)
                out[tag + '.indices.heads'] = idx[:, heads]
                weights = (np.arange(idx.shape[-1], dtype=np.int64) % 8191) + 1
                out[tag + '.indices.checksum'] = (idx.astype(np.int64) * weights).sum(-1)
    for key in list(out):
        if 'cos_cached' in key or 'sin_cached' in key or 'cached_ids' in key:
            del out[key]
    save('attention_mfma', **out)


BLOCK_MFMA = {
    'opt': dict(d_model=1024, n_heads=16, d_ff=256, N=2, S=128),      # 16 heads x 64
    'llama': dict(d_model=2048, n_heads=16, d_ff=512, N=2, S=128),    # 16 heads x 128
}


def golden_block_mfma():
    """Four-stage upgraded TransformerBlocks with N * H = 32 and BERT-large / LLaMA head
    shapes.  Weights and inputs are seeded by name (4-16 MB per matrix: not stored)."""
    out = {}
    for kind, c in BLOCK_MFMA.items():
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
        model = quiet_upgrade(block, d_lora=4)
        tag = 'block_mfma.{}.'.format(kind)
        for key, fp in seeded_fill(model, tag).items():
            out[kind + '.print.' + key] = fp
        x = torch.from_numpy(seeded(tag + 'x', [N, S, d_model]))
        w = torch.from_numpy(seeded(tag + 'w', [N, S, d_model]))
        xi = x.clone().requires_grad_(True)
        y = model(xi)
        (y * w).sum().backward()
        out[kind + '.trainable'] = np.array(
            sorted(n for n, p in model.named_parameters() if p.requires_grad))
        for name, t in (('y', y), ('grad_x', xi.grad)):
            out['{}.{}.sub'.format(kind, name)] = t2n(t[:, ::4])
            out['{}.{}.sums'.format(kind, name)] = t2n(t.double().sum(-1))            # [N, S]
        out.update(grads2n(model, kind + '.grad.'))
    save('block_mfma', **out)


def golden_dense_chain():
    """Pins the oracle's sddmm -> softmax -> spmm leg (and the reference's autograd wiring
    of it, kernels/{sddmm,softmax,spmm}.py) to the reference's own DENSE attention: on the
    full causal pattern (every row lists all S columns; softmax masks col > row) with
    scores far inside the clamp, the sparse chain IS `VanillaAttention` with a causal mask
    (naive_gpt/layers/basic/attention.py:41-57)."""
    gen = torch.Generator().manual_seed(808)
    N, S, H, E = 1, 64, 2, 32
    q, k, v, noise = [0.5 * torch.randn([N, S, H, E], generator=gen) for _ in range(4)]
    model = layers.VanillaAttention(d_head=E, p_dropout=0.0)
    qi, ki, vi = [t.clone().requires_grad_(True) for t in (q, k, v)]
    causal = torch.full([S, S], float('-inf')).triu(1)
    # The upstream gradient w keeps <w_i, y_i> > 0 in every row: softmax_backward clamps
    # sum_j y_ij dy_ij = <dY_i, Y_i> at >= 1e-9 (softmax.cu:69, a reference quirk that its own
    # test hides behind torch.max, test_softmax.py:71,78), so only there is the sparse
    # backward the true softmax VJP that the dense attention differentiates.
    with torch.no_grad():
        w = model(q, k, v, attn_mask=causal) + 0.1 * noise
        assert float((w * model(q, k, v, attn_mask=causal)).sum(-1).min()) > 1e-3
    y = model(qi, ki, vi, attn_mask=causal)
    (y * w).sum().backward()
    save('dense_chain', q=t2n(q), k=t2n(k), v=t2n(v), w=t2n(w), y=t2n(y),
         grad_q=t2n(qi.grad), grad_k=t2n(ki.grad), grad_v=t2n(vi.grad))


# --------------------------------------------------------------------------- models + one tuning step

def golden_models():
    """OPTModel / LLaMAModel of the reference (naive_gpt/models): dense logits, then the
    four-stage upgrade and ONE optimisation step restated from
    script/4-sparse-tuning-0.py:45-93,183-187 (Lightning itself is not installed: the
    step below is its training_step + AdamW + clip 1.0, written out)."""
    from naive_gpt import models
    from torch import optim
    out = {}
    config = dict(d_model=64, n_heads=1, n_layers=2, max_length=128, vocab_size=96,
                  d_feedforward=128, p_dropout=0.0)
    out['config'] = np.array([config[k] for k in
                              ['d_model', 'n_heads', 'n_layers', 'max_length', 'vocab_size',
                               'd_feedforward']])
    gen = torch.Generator().manual_seed(123)
    batch = torch.randint(3, 96, [2, 130], generator=gen)
    batch[:, 0] = torch.tensor([17, 90])          # MMLU answer positions
    out['batch'] = t2n(batch)
    skip = ('attn_mask', 'cos_cached', 'sin_cached', 'cached_ids')
    for kind, cls in [('opt', models.OPTModel), ('llama', models.LLaMAModel)]:
        torch.manual_seed(31 if kind == 'opt' else 32)
        model = cls(**config)
        model.eval()
        with torch.no_grad():
            out[kind + '.dense.logits'] = t2n(model(batch[:, 1:-1]))
        model = quiet_upgrade(model, d_lora=4)
        randomize_lora(model, gen)
        model.train()
        for k, v in sd2n(model, kind + '.sd.').items():
            if not k.endswith(skip):
                out[k] = v
        params = [p for p in model.parameters() if p.requires_grad]
        optimizer = optim.AdamW(model.parameters(), lr=1e-4, weight_decay=1e-1)
        true_tensor = torch.scalar_tensor(True, dtype=torch.bool)
        for name, trigger in model.named_buffers():
            if name.endswith('.trigger'):
                trigger.fill_(true_tensor)
        logits = model(batch[:, 1:-1])
        loss = nn.CrossEntropyLoss()(logits.flatten(end_dim=-2), target=batch[:, 2:].flatten())
        aux = 0.0
        for name, buffer in model.named_buffers():
            if name.endswith('.loss'):
                aux += buffer
        total = loss + 1e-2 * aux
        total.backward()
        out.update({kind + '.logits': t2n(logits), kind + '.ce': t2n(loss),
                    kind + '.aux': t2n(aux), kind + '.loss': t2n(total)})
        out.update(grads2n(model, kind + '.grad.'))
        norm = nn.utils.clip_grad_norm_(params, 1.0)
        out[kind + '.grad_norm'] = t2n(norm)
        optimizer.step()
        for name, p in model.named_parameters():
            if p.requires_grad:
                out[kind + '.stepped.' + name] = t2n(p)
    save('models', **out)


if __name__ == '__main__':
    torch.set_num_threads(4)
    golden_ffn()
    golden_pq()
    golden_attention()
    golden_block()
    golden_attention_mfma()
    golden_block_mfma()
    golden_dense_chain()
    golden_models()
