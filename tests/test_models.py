"""naive_gpt.models, the checkpoint format and the Lightning-free tuning step.

Pins, in order of independence:
* HuggingFace `transformers` (installed here, random-init from a config: no download) --
  OPTForCausalLM / LlamaForCausalLM logits after the weight map of utils.checkpoint
  (the check script/1-convert.py:162-181 makes);
* tests/golden/models.npz from the IMPORTED REFERENCE: dense logits, and ONE
  optimisation step of the four-stage-upgraded model (loss, aux loss, every gradient,
  clip norm, every stepped parameter) restated from script/4-sparse-tuning-0.py.
CPU tier runs the sparse attentions through the oracle (conftest.oracle_ext); the GPU
tier runs the same goldens through the HIP library.
"""
import os

import numpy as np
import pytest
import torch

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
CONFIG_KEYS = ['d_model', 'n_heads', 'n_layers', 'max_length', 'vocab_size', 'd_feedforward']


def golden():
    g = dict(np.load(os.path.join(GOLD, 'models.npz'), allow_pickle=False))
    config = {k: int(v) for k, v in zip(CONFIG_KEYS, g['config'])}
    config['p_dropout'] = 0.0
    return g, config


def T(a, device='cpu'):
    return torch.from_numpy(np.ascontiguousarray(a)).to(device)


def close(got, want, rtol, atol, what):
    got = got.detach().cpu().numpy()
    assert got.shape == want.shape, what
    err = np.abs(got - want)
    bound = atol + rtol * np.abs(want)
    assert (err <= bound).all(), '{}: max err {:.3e}'.format(what, err.max())


def upgraded_from_golden(kind, device):
    from naive_gpt import utils
    from naive_gpt.utils import checkpoint
    g, config = golden()
    model = checkpoint.build_model(kind, config)
    model = utils.upgrade_sparse(model, d_lora=4)
    prefix = kind + '.sd.'
    sd = {k[len(prefix):]: T(v) for k, v in g.items() if k.startswith(prefix)}
    report = model.load_state_dict(sd, strict=False)
    assert not report.unexpected_keys
    assert all(k.endswith(('attn_mask', 'cos_cached', 'sin_cached', 'cached_ids'))
               for k in report.missing_keys), report.missing_keys
    return g, model.to(device)


def run_tuning_step(kind, device, rtol, atol):
    from naive_gpt import utils
    g, model = upgraded_from_golden(kind, device)
    batch = T(g['batch'], device)
    tuner = utils.SparseTuner(model)          # recipe defaults: AdamW 1e-4 / 0.1, clip 1.0
    assert sorted(n for n, p in model.named_parameters() if p.requires_grad)

    # the step, piece by piece (training_step == these pieces, checked below)
    model.train()
    tuner.arm_triggers()
    logits, ce = tuner.shared_step(batch[:, 1:-1], target=batch[:, 2:])
    aux = tuner.aux_loss()
    loss = ce + tuner.aux_weight * aux
    loss.backward()
    close(logits, g[kind + '.logits'], rtol, atol * 10, 'logits')
    close(ce, g[kind + '.ce'], rtol, atol, 'ce')
    close(aux, g[kind + '.aux'], rtol, atol, 'aux')
    close(loss, g[kind + '.loss'], rtol, atol, 'loss')
    n = 0
    for name, p in model.named_parameters():
        key = kind + '.grad.' + name
        if key in g:
            assert p.grad is not None, name
            close(p.grad, g[key], rtol, atol, 'grad ' + name)
            n += 1
        else:
            assert p.grad is None, name
    assert n > 10
    tuner.apply_gradients()
    close(tuner.last_grad_norm, g[kind + '.grad_norm'], rtol, atol, 'grad_norm')
    for name, p in model.named_parameters():
        if p.requires_grad:
            close(p, g[kind + '.stepped.' + name], rtol, atol, 'stepped ' + name)
    assert all(p.grad is None for p in tuner.params)


# ------------------------------------------------------------------ dense models

@pytest.mark.parametrize('kind', ['opt', 'llama'])
def test_dense_logits_match_imported_reference(kind):
    """Seeded construction draws the same initial weights as the reference's model, so
    the logits agree without loading anything."""
    from naive_gpt.utils import checkpoint
    g, config = golden()
    torch.manual_seed(31 if kind == 'opt' else 32)
    model = checkpoint.build_model(kind, config).eval()
    with torch.no_grad():
        logits = model(T(g['batch'])[:, 1:-1])
    close(logits, g[kind + '.dense.logits'], 1e-5, 1e-5, 'logits')


def test_parameter_names_are_the_checkpoint_format():
    from naive_gpt import models
    opt = models.OPTModel(d_model=32, n_heads=2, n_layers=1, max_length=16, vocab_size=20,
                          d_feedforward=64, p_dropout=0.0)
    names = set(opt.state_dict())
    for key in ['embedding.weight', 'learned_pe.weight', 'final_norm.bias', 'lm_output.weight',
                'attn_mask', 'decoders.0.mha.linear_q.bias', 'decoders.0.ffd.fc1.weight',
                'decoders.0.norm2.weight']:
        assert key in names, key
    assert opt.learned_pe.weight.shape == (16 + 2, 32) and opt.attn_mask.shape == (18, 18)
    llama = models.LLaMAModel(d_model=32, n_heads=2, n_layers=1, max_length=16, vocab_size=20,
                              d_feedforward=64, p_dropout=0.0)
    names = set(llama.state_dict())
    assert 'learned_pe.weight' not in names and 'final_norm.bias' not in names
    for key in ['decoders.0.ffd.gate.weight', 'decoders.0.ffd.side.weight',
                'decoders.0.ffd.down.weight', 'final_norm.weight']:
        assert key in names, key
    assert llama.attn_mask.shape == (16, 16)
    mask = llama.attn_mask
    assert mask[3, 3] == 0 and mask[3, 2] == 0 and mask[2, 3] == float('-inf')


@pytest.mark.parametrize('kind', ['opt', 'llama'])
def test_logits_match_huggingface_after_weight_map(kind):
    T5 = pytest.importorskip('transformers')
    from naive_gpt.utils import checkpoint
    torch.manual_seed(0)
    if kind == 'opt':
        cfg = T5.OPTConfig(vocab_size=96, hidden_size=64, num_hidden_layers=2, ffn_dim=128,
                           num_attention_heads=4, max_position_embeddings=64,
                           word_embed_proj_dim=64, dropout=0.0, attention_dropout=0.0,
                           activation_function='relu', do_layer_norm_before=True)
        hf = T5.OPTForCausalLM(cfg).eval()
    else:
        cfg = T5.LlamaConfig(vocab_size=96, hidden_size=64, num_hidden_layers=2,
                             intermediate_size=176, num_attention_heads=4,
                             num_key_value_heads=4, max_position_embeddings=64,
                             rms_norm_eps=1e-6, attention_dropout=0.0)
        hf = T5.LlamaForCausalLM(cfg).eval()
    model = checkpoint.build_model(kind, checkpoint.config_from_hf(cfg)).eval()
    checkpoint.load_hf_state(model, kind, hf.state_dict())
    x = torch.randint(0, 96, [3, 40])
    with torch.no_grad():
        want, got = hf(x)['logits'], model(x)
    assert torch.allclose(got, want, atol=1e-5), (got - want).abs().max()   # 1-convert.py:181: 1e-3


def test_weight_map_rejects_unknown_tensors():
    from naive_gpt.utils import checkpoint
    with pytest.raises(RuntimeError):
        checkpoint.opt_state_from_hf({'model.decoder.layers.0.mystery.weight': torch.zeros(1)})


def test_checkpoint_round_trip(tmp_path):
    from naive_gpt import utils
    from naive_gpt.utils import checkpoint
    _, config = golden()
    torch.manual_seed(3)
    model = checkpoint.build_model('llama', config)
    path = str(tmp_path / 'tiny-llama.ckpt')
    utils.save_checkpoint(path, config, model)
    loaded_config, state = utils.load_checkpoint(path)
    assert loaded_config == config and set(state) == set(model.state_dict())
    again = utils.model_from_checkpoint(path)          # family from the file name
    assert type(again).__name__ == 'LLaMAModel'
    x = torch.randint(0, config['vocab_size'], [1, 32])
    with torch.no_grad():
        assert torch.equal(again(x), model(x))
    with pytest.raises(RuntimeError):
        checkpoint.model_family('bert-large.ckpt')
    torch.save({'weights': 1}, str(tmp_path / 'opt-bad.ckpt'))
    with pytest.raises(RuntimeError):
        utils.load_checkpoint(str(tmp_path / 'opt-bad.ckpt'))


# ------------------------------------------------------------------ the tuning step

@pytest.mark.parametrize('kind', ['opt', 'llama'])
def test_tuning_step_matches_imported_reference_cpu(kind, oracle_ext):
    run_tuning_step(kind, 'cpu', 1e-4, 1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize('kind', ['opt', 'llama'])
def test_tuning_step_matches_imported_reference_gpu(kind):
    run_tuning_step(kind, 'cuda', 2e-3, 2e-5)


def test_training_step_is_the_pieces_and_accumulates(oracle_ext):
    from naive_gpt import utils
    g, model_a = upgraded_from_golden('opt', 'cpu')
    _, model_b = upgraded_from_golden('opt', 'cpu')
    batch = T(g['batch'])
    a = utils.SparseTuner(model_a)
    loss = a.training_step(batch)
    close(loss, g['opt.loss'], 1e-4, 1e-6, 'loss')
    for name, p in model_a.named_parameters():
        if p.requires_grad:
            close(p, g['opt.stepped.' + name], 1e-4, 1e-6, name)
    # two half-weight micro-batches of the same data == one full step
    b = utils.SparseTuner(model_b, n_accumulate=2)
    b.training_step(batch)
    assert all(p.grad is not None for p in b.params)
    before = [p.detach().clone() for p in b.params]
    assert all(torch.equal(x, y) for x, y in zip(before, b.params))     # no step yet
    b.training_step(batch)
    for pa, pb in zip(a.params, b.params):
        assert torch.allclose(pa, pb, rtol=1e-4, atol=1e-6)
    # the scheduler decays per epoch
    lr0 = a.lr
    a.end_epoch()
    assert abs(a.lr - 0.9 * lr0) < 1e-12


def test_flat_parameter_store_takes_the_per_tensor_step(oracle_ext, monkeypatch):
    """SparseTuner keeps the trainable parameters and their gradients in one flat buffer (one
    tensor for the exchange, the clip and AdamW); above FLAT_LIMIT elements it keeps the
    per-tensor path.  Both take the same steps: same loss, same clip norm, same parameters, with
    a parameter that receives no gradient (its slice of the flat gradient is zero) included."""
    from naive_gpt import utils
    g, model_a = upgraded_from_golden('opt', 'cpu')
    _, model_b = upgraded_from_golden('opt', 'cpu')
    batch = T(g['batch'])
    flat = utils.SparseTuner(model_a)
    assert flat._flat is not None and flat._flat.numel() >= sum(p.numel() for p in flat.params)
    # every trainable parameter is a view of the flat buffer, in order, on a 16-byte boundary
    offset = 0
    for p in flat.params:
        assert p.data_ptr() == flat._flat.data_ptr() + 4 * offset and (4 * offset) % 16 == 0
        offset += (p.numel() + 3) // 4 * 4
    assert offset == flat._flat.numel()
    monkeypatch.setattr(utils.SparseTuner, 'FLAT_LIMIT', 0)
    plain = utils.SparseTuner(model_b)
    assert plain._flat is None
    for step in range(3):
        pq = step != 1                       # step 1: the quantizer tables get no gradient
        la, lb = flat.training_step(batch, pq_loss=pq), plain.training_step(batch, pq_loss=pq)
        assert torch.allclose(la, lb, rtol=1e-6, atol=0)
        assert torch.allclose(flat.last_grad_norm, plain.last_grad_norm, rtol=1e-5, atol=0)
        for (n, pa), pb in zip(model_a.named_parameters(), model_b.parameters()):
            assert torch.allclose(pa, pb, rtol=1e-5, atol=1e-7), (step, n)
    assert all(p.grad is None for p in flat.params)
    assert flat._flat is None        # step 1 (a parameter without gradient) ended the flat form, exactly
    # the state dict reads through the views
    sd = model_a.state_dict()
    for n, p in model_a.named_parameters():
        assert torch.equal(sd[n], p)


def test_flat_store_survives_repointed_parameters_and_refuses_a_second_owner(oracle_ext):
    """Round-2 ADVICE: the optimiser only owns the flat buffer, so a parameter whose `.data` was
    re-pointed (`model.float()`, `p.data = ...`) would silently stop being trained, and a second
    tuner on the same model would leave the first one updating dead memory.  The tuner now checks
    the views before every update (re-homes a moved parameter with its current values), pads every
    slice to 16 bytes (float4 kernels read the views) and refuses a second live owner."""
    from naive_gpt import utils
    g, model_a = upgraded_from_golden('opt', 'cpu')
    _, model_b = upgraded_from_golden('opt', 'cpu')
    batch = T(g['batch'])
    # a trainable parameter whose length is not a multiple of four elements: padded slice
    odd_a, odd_b = [torch.nn.Parameter(torch.full([7], 0.5)) for _ in range(2)]
    model_a.register_parameter('odd', odd_a)
    model_b.register_parameter('odd', odd_b)
    a, b = utils.SparseTuner(model_a), utils.SparseTuner(model_b)
    assert a._pads and all(off % 4 == 0 for off in a._offsets)
    for tuner, odd in ((a, odd_a), (b, odd_b)):
        hook = tuner.model.embedding.register_forward_hook(lambda m, i, o, odd=odd: o + odd.sum() * 1e-3)
        tuner._hook = hook
    a.training_step(batch)
    b.training_step(batch)
    # re-point two parameters of model_b the way `model.float()` / a manual assignment would
    moved = [p for p in b.params if p.dim() == 2][:2]
    for p in moved:
        p.data = p.data.clone() * 1.0
    assert any(p.data_ptr() != b._flat.data_ptr() + 4 * off for p, off in zip(b.params, b._offsets))
    a.training_step(batch)
    b.training_step(batch)                                  # heals the views, then updates
    for p, off in zip(b.params, b._offsets):
        assert p.data_ptr() == b._flat.data_ptr() + 4 * off
    for (n, pa), pb in zip(model_a.named_parameters(), model_b.parameters()):
        assert torch.allclose(pa, pb, rtol=1e-6, atol=1e-8), n
    assert not torch.equal(odd_a.detach(), torch.full([7], 0.5))         # the padded slice is trained
    # a second owner is refused while the first is alive; `release()` hands the parameters back
    with pytest.raises(RuntimeError):
        utils.SparseTuner(model_a)
    a.release()
    c = utils.SparseTuner(model_a)
    assert c._flat is not None
    a.training_step(batch)                                   # the released tuner still steps (per tensor)
    # another dtype cannot be expressed by the flat buffer: loud, not silent
    c.params[0].data = c.params[0].data.double()
    with pytest.raises(RuntimeError):
        c.apply_gradients()


def test_aux_loss_sees_modules_replaced_after_the_first_step(oracle_ext):
    """`aux_loss` caches the module list; a layer replaced later (same parent, same name) must
    still be found (round-2 ADVICE)."""
    import copy
    from naive_gpt import utils
    g, model = upgraded_from_golden('opt', 'cpu')
    tuner = utils.SparseTuner(model)
    batch = T(g['batch'])
    tuner.training_step(batch)
    block = model.decoders[0]
    old = block.mha.attn_fn
    old._buffers.pop('loss', None)                           # (a graph output: not deep-copyable)
    block.mha.add_module('attn_fn', copy.deepcopy(old))
    assert block.mha.attn_fn is not old
    tuner._armable = [m for m in model.modules() if hasattr(m, 'arm')]
    tuner.arm_triggers()
    model(batch[:, 1:-1])
    found = tuner.aux_loss()
    want = sum(m._buffers['loss'] for m in model.modules() if m._buffers.get('loss') is not None)
    assert torch.allclose(found, want)
    assert block.mha.attn_fn._buffers.get('loss') is not None


def test_hooked_blocks_and_checkpointed_blocks_take_the_same_step(oracle_ext):
    """`DecoderLM.hidden` runs the pre-norm stack as (stream, addend) pairs through
    `block.forward_pair`, past `nn.Module.__call__`: a block with hooks must still see them fire
    (it takes the plain loop), and a block whose forward is re-run inside the backward (activation
    checkpointing: `recompute.tag` drops its one-entry cache on every forward) must produce the same
    gradients (round-2 ADVICE)."""
    from torch.utils import checkpoint as ckpt
    from naive_gpt import utils
    g, model = upgraded_from_golden('opt', 'cpu')
    batch = T(g['batch'])
    src, target = batch[:, 1:-1], batch[:, 2:]

    def grads():
        model.zero_grad()
        out = model(src)
        torch.nn.functional.cross_entropy(out.flatten(end_dim=-2), target.flatten()).backward()
        return out.detach(), {n: p.grad.clone() for n, p in model.named_parameters() if p.grad is not None}

    out0, g0 = grads()
    seen = []
    handle = model.decoders[0].register_forward_hook(lambda m, i, o: seen.append(tuple(o.shape)))
    out1, g1 = grads()
    handle.remove()
    assert seen == [tuple(out0.shape[:2]) + (model.embedding.embedding_dim,)]
    assert torch.allclose(out1, out0, rtol=1e-5, atol=1e-6)
    for n in g0:
        assert torch.allclose(g1[n], g0[n], rtol=1e-4, atol=1e-7), n
    # every block under torch.utils.checkpoint (its forward runs again in the backward)
    block = model.decoders[0]
    original = type(block).forward_pair

    def checkpointed(self, x, pending, attn_mask=None):
        if pending is None:
            pending = torch.zeros_like(x)
        return ckpt.checkpoint(lambda a, b: original(self, a, b, attn_mask=attn_mask), x, pending,
                               use_reentrant=False)

    type(block).forward_pair = checkpointed
    try:
        out2, g2 = grads()
    finally:
        type(block).forward_pair = original
    assert torch.allclose(out2, out0, rtol=1e-5, atol=1e-6)
    assert set(g2) == set(g0)
    for n in g0:
        assert torch.allclose(g2[n], g0[n], rtol=1e-4, atol=1e-7), n


def test_unarmed_step_registers_no_pq_loss(oracle_ext):
    from naive_gpt import utils
    g, model = upgraded_from_golden('opt', 'cpu')
    tuner = utils.SparseTuner(model)
    tuner.training_step(T(g['batch']), pq_loss=False)
    assert not [n for n, _ in model.named_buffers() if n.endswith('.loss')]
    quant = [p for n, p in model.named_parameters() if n.endswith('quantizer.weight')]
    assert quant and all(p.grad is None for p in quant)


def test_validation_step_metrics(oracle_ext):
    from naive_gpt import utils
    g, model = upgraded_from_golden('llama', 'cpu')
    tuner = utils.SparseTuner(model)
    batch = T(g['batch'])
    out = tuner.validation_step(batch)
    assert abs(out['ppl'].item() - float(np.exp(out['loss'].item()))) < 1e-3 * out['ppl'].item()
    # the reference's [B, B] accuracy grid (script/4-sparse-tuning-0.py:110-121)
    with torch.no_grad():
        logits = model(batch[:, 1:-1])
    pos = batch[:, 0]
    grid = torch.stack([torch.stack([(logits[i, pos[j] - 2].argmax() == batch[i, pos[j]]).float()
                                     for j in range(2)]) for i in range(2)])
    assert abs(out['accuracy'].item() - grid.mean().item()) < 1e-6


@pytest.mark.gpu
@pytest.mark.parametrize('n_seq', [4, 16])
def test_captured_step_replays_the_eager_step(n_seq):
    """SparseTuner.capture(): the whole training step (forward through the HIP kernels, backward,
    clip, AdamW) as one HIP graph.  Three replays on fresh batches leave the parameters where
    three eager steps of an identical tuner with the same (capturable) optimiser leave them.
    (Against the default AdamW the comparison is ill-posed after the first step: its 2e-7
    rounding difference flips PQ codes / top-k choices, and the gradient norm moves by 4e-4.)
    n_seq = 16: 4096 tokens -- the matrix-core linears, the fused head loss, and more than the 3072
    ids above which torch's own embedding backward sizes its launches from a host-side count of
    the capture batch's distinct tokens (layers/tuning/lora.py: _LookupRows replaces it)."""
    from naive_gpt import models, utils
    config = dict(d_model=1024, n_heads=16, n_layers=2, max_length=256, vocab_size=512,
                  d_feedforward=4096, p_dropout=0.0)

    def build():
        torch.manual_seed(3)
        model = models.OPTModel(**config)
        model = utils.upgrade_sparse(model, d_lora=16)
        for name, p in model.named_parameters():
            if name.endswith('lora.right.weight'):
                p.data.normal_(0, 0.02)
        return utils.SparseTuner(model.cuda())

    gen = torch.Generator().manual_seed(9)
    batches = [torch.randint(3, 512, [n_seq, 258], generator=gen).cuda() for _ in range(3)]
    eager, graphed = build(), build()
    # capture() warms up with three eager steps (library initialisations must not fall into the
    # capture): the eager twin takes the same three.  (Not on the default all-zero batch: with
    # every token equal most gradients are rounding noise, whose sign Adam turns into +-lr.)
    warm = torch.randint(3, 512, [n_seq, 258], generator=gen).cuda()
    eager.use_capturable_optimizer()
    graphed.capture(batches[0].shape, pq_loss=True, warmup=3, example=warm)
    for _ in range(3):
        eager.training_step(warm, pq_loss=True)
    losses = []
    for b in batches:
        le = eager.training_step(b, pq_loss=True)
        lg = graphed.training_step(b, pq_loss=True)
        losses.append((float(le), float(lg)))
    for le, lg in losses:
        assert abs(le - lg) <= 1e-6 * abs(le), losses
    def apart():
        return sorted(((float((pe - pg).abs().max()), n)
                       for (n, pe), pg in zip(eager.model.named_parameters(), graphed.model.parameters())
                       if pe.requires_grad and not torch.allclose(pe, pg, rtol=1e-6, atol=1e-7)), reverse=True)

    assert not apart(), (len(apart()), apart()[:4])
    # back-to-back replays without a host synchronisation in between (bench.py's timed loop), then
    # the same steps of the eager twin
    if n_seq == 16:
        for b in batches + batches:
            graphed.training_step(b, pq_loss=True)
        torch.cuda.synchronize()
        for b in batches + batches:
            eager.training_step(b, pq_loss=True)
        torch.cuda.synchronize()
        assert not apart(), (len(apart()), apart()[:4])
        # and INTERLEAVED with the twin's eager launches, no host synchronisation of the test's own:
        # the form round 3 saw diverge for one allocation pattern.  A replay
        # that finds eager launches of the library on the stream since its last one drains the
        # stream first (SparseTuner.training_step); dropping the package's caches under the live
        # graph must not pull anything from under it (capture() pins what the kernels read)
        from naive_gpt import ext
        ext.drop_images()
        for b in batches + batches:
            eager.training_step(b, pq_loss=True)
            graphed.training_step(b, pq_loss=True)
        torch.cuda.synchronize()
        assert not apart(), (len(apart()), apart()[:4])
    # a different batch shape falls back to the eager path
    other = torch.randint(3, 512, [2, 130], generator=gen).cuda()
    assert torch.isfinite(graphed.training_step(other, pq_loss=True))
    graphed.end_epoch()
    assert abs(graphed.lr - 0.9e-4) < 1e-9


def test_arm_is_visible_through_armed_and_does_not_travel_with_copies():
    """`module.arm()` is a host-side note (no device write): `module.trigger` stays False, which is
    where it differs observably from the reference's `trigger.fill_(True)` (attention.py:98-104).
    `module.armed` ORs the two; a deepcopy or a state_dict round trip of an armed module starts
    disarmed (the note is this module's next forward's, not model state); the reference's protocol
    -- a write to the buffer -- still arms."""
    import copy
    from naive_gpt import layers
    attn = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0)
    assert not attn.armed
    attn.arm()
    assert attn.armed and not bool(attn.trigger)
    twin = copy.deepcopy(attn)
    assert not twin.armed and attn.armed
    fresh = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0)
    fresh.load_state_dict(attn.state_dict())
    assert not fresh.armed
    assert attn._take_trigger() and not attn.armed          # one shot
    attn.trigger.fill_(True)                                # the reference's way
    assert attn.armed and bool(attn.trigger)
    assert copy.deepcopy(attn).armed                        # device state DOES travel
    assert attn._take_trigger() and not attn.armed and not bool(attn.trigger)
