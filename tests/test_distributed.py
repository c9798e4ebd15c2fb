"""Data parallel path on CPU: world_size 2, gloo backend (the N > 1 path of bench.py).

Each rank runs the upgraded (LoRA + routed FFN) block on its own micro-batch, the
trainable gradients are averaged with one flat all-reduce, and the result must equal
the single-process gradient of the mean loss over both micro-batches."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp
from torch import nn


def _free_port():
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return s.getsockname()[1]


def _build():
    from naive_gpt import layers, utils
    torch.manual_seed(0)
    ffn = layers.Feedforward(d_model=16, d_feedforward=64, p_dropout=0.0, activation=nn.ReLU())
    model = nn.Sequential(ffn)
    import contextlib
    import io
    with contextlib.redirect_stdout(io.StringIO()):
        for stage in ['lora', 'ffn']:
            model = utils.ModuleUpgrader(utils.SparseLoRAHandler(d_lora=4, stage=stage)).visit(model)
    gen = torch.Generator().manual_seed(1)
    for name, p in model.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.copy_(0.1 * torch.randn(p.shape, generator=gen))
    return model


def _batch(rank):
    return torch.randn([3, 8, 16], generator=torch.Generator().manual_seed(100 + rank))


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naive_gpt import utils
    model = _build()
    if rank == 1:                         # replicas start different; broadcast must fix it
        for p in model.parameters():
            p.data.add_(1.0)
    utils.broadcast_parameters(model, src=0)
    model(_batch(rank)).square().mean().backward()
    n = utils.allreduce_gradients(utils.trainable_parameters(model), world_size=world)
    grads = {k: p.grad.clone() for k, p in model.named_parameters() if p.requires_grad}
    if rank == 0:
        torch.save({'n': n, 'grads': grads}, out)
    dist.barrier()
    dist.destroy_process_group()


def test_flat_allreduce_matches_single_process(tmp_path):
    world = 2
    out = str(tmp_path / 'rank0.pt')
    mp.spawn(_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    got = torch.load(out, weights_only=True)

    model = _build()
    loss = sum(model(_batch(r)).square().mean() for r in range(world)) / world
    loss.backward()
    want = {k: p.grad for k, p in model.named_parameters() if p.requires_grad}
    assert set(want) == set(got['grads'])
    assert got['n'] == sum(v.numel() for v in want.values())
    # frozen base weights never travel
    assert not any(k.endswith('fc1.weight') or k.endswith('fc2.weight') for k in want)
    for k in want:
        assert torch.allclose(got['grads'][k], want[k], atol=1e-6), k


def test_allreduce_is_a_noop_for_one_rank():
    from naive_gpt import utils
    model = _build()
    model(_batch(0)).sum().backward()
    assert utils.allreduce_gradients(utils.trainable_parameters(model), world_size=1) == 0


# ------------------------------------------------------------------ the tuner under DP

def _tiny_lm():
    from naive_gpt import models, utils
    torch.manual_seed(4)
    model = models.OPTModel(d_model=16, n_heads=2, n_layers=1, max_length=24, vocab_size=32,
                            d_feedforward=64, p_dropout=0.0)
    model = utils.upgrade_sparse(model, d_lora=4, stages=('lora', 'ffn'))   # pure PyTorch on CPU
    gen = torch.Generator().manual_seed(5)
    for name, p in model.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.copy_(0.1 * torch.randn(p.shape, generator=gen))
    return model


def _tokens(rank):
    return torch.randint(0, 32, [2, 20], generator=torch.Generator().manual_seed(200 + rank))


def _tuner_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naive_gpt import utils
    model = _tiny_lm()
    if rank == 1:
        for p in model.parameters():
            p.data.mul_(1.5)               # the tuner's broadcast must undo this
    tuner = utils.SparseTuner(model, clip_norm=0.05)      # small enough to clip for real
    assert tuner.world_size == world
    tuner.allreduce_every = 1          # bench.py's measurement hook: HIP events on CUDA, inert on CPU
    norms = []
    for _ in range(2):
        tuner.training_step(_tokens(rank), pq_loss=False)
        norms.append(tuner.last_grad_norm.detach().clone())
    assert tuner._updates == 2 and tuner.allreduce_events == []
    saved = {k: p.detach().clone() for k, p in model.named_parameters() if p.requires_grad}
    saved['__clip_norms'] = torch.stack(norms)
    torch.save(saved, out.format(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_tuner_replicas_stay_identical_and_match_the_global_batch(tmp_path):
    """Two ranks, two steps: both ranks end with the same parameters, equal to a single
    process that takes the mean gradient of both micro-batches, clips it by the same norm
    and applies the same AdamW step."""
    world = 2
    out = str(tmp_path / 'rank{}.pt')
    mp.spawn(_tuner_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    r0 = torch.load(out.format(0), weights_only=True)
    r1 = torch.load(out.format(1), weights_only=True)
    assert set(r0) == set(r1) and len(r0) > 5
    for k in r0:
        assert torch.equal(r0[k], r1[k]), k
    # (bench.py asserts this at N > 1: the ranks clip by the same norm, taken after the exchange)
    norms = r0.pop('__clip_norms')
    r1.pop('__clip_norms')
    assert bool((norms > 0.05).all())

    from naive_gpt import utils
    model = _tiny_lm()
    tuner = utils.SparseTuner(model, clip_norm=0.05)
    for _ in range(2):
        for r in range(world):           # mean of the per-rank losses == global-batch loss
            batch = _tokens(r)
            loss = tuner.shared_step(batch[:, 1:-1], target=batch[:, 2:])[-1]
            (loss / world).backward()
        tuner.apply_gradients()
        assert float(tuner.last_grad_norm) > 0.05        # the clip was active
    for k, p in model.named_parameters():
        if p.requires_grad:
            assert torch.allclose(p, r0[k], rtol=1e-5, atol=1e-7), k


@pytest.mark.gpu
def test_rccl_path_on_one_gpu():
    """The N > 1 code path of bench.py / SparseTuner with the real backend: a one-rank RCCL
    group on cuda:0 (the pool gives this session a single GPU; the 2..8-rank runs are the
    driver's).  Broadcast of parameters and buffers (bool `trigger` included) and the flat
    gradient all-reduce must run on device tensors and leave the values unchanged."""
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(_free_port())
    dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
    try:
        from naive_gpt import layers, utils
        attn = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16,
                                               p_dropout=0.0).cuda()
        attn.trigger.fill_(True)
        before = attn.quantizer.weight.detach().clone()
        utils.broadcast_parameters(attn, src=0)
        assert torch.equal(attn.quantizer.weight, before) and bool(attn.trigger)
        params = utils.trainable_parameters(attn)
        for p in params:
            p.grad = torch.randn_like(p)
        want = [p.grad.clone() for p in params]
        n = utils.allreduce_gradients(params, world_size=1, force=True)
        assert n == sum(p.numel() for p in params)
        for p, w in zip(params, want):
            assert torch.allclose(p.grad, w)
        t = torch.tensor([1.5], device='cuda', dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dist.barrier()
        assert t.item() == 1.5
    finally:
        dist.destroy_process_group()


# ------------------------------------------------------------------ triggers under broadcast

def _trigger_worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from naive_gpt import layers, utils
    attn = layers.SparseVanillaAttentionV2(d_head=16, d_codeword=8, n_codewords=16, p_dropout=0.0)
    first = attn._take_trigger()          # an earlier forward: caches "disarmed" per version
    if rank == 0:
        attn.trigger.fill_(True)          # the reference's way of arming (4-sparse-tuning-0.py:71-78)
    utils.broadcast_parameters(attn, src=0)
    armed = attn._take_trigger()
    again = attn._take_trigger()          # one-shot: disarmed by the look above
    torch.save({'first': first, 'armed': armed, 'again': again}, out.format(rank))
    dist.barrier()
    dist.destroy_process_group()


def test_broadcast_of_an_armed_trigger_is_seen_by_every_rank(tmp_path):
    """`broadcast_parameters` must invalidate the layers' cached reading of `trigger`
    (keyed on the buffer's version counter): a trigger armed on rank 0 only and then
    broadcast has to make EVERY rank compute the PQ loss on its next forward."""
    world = 2
    out = str(tmp_path / 'trigger{}.pt')
    mp.spawn(_trigger_worker, args=(world, _free_port(), out), nprocs=world, join=True)
    for r in range(world):
        got = torch.load(out.format(r), weights_only=True)
        assert got == {'first': False, 'armed': True, 'again': False}, (r, got)


# ------------------------------------------------------------------ bench.py starts its own ranks

def _run_bench(*argv, env=None):
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    full_env = {k: v for k, v in os.environ.items()
                if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_PORT')}
    full_env.update(env or {})
    return subprocess.run([sys.executable, os.path.join(root, 'bench.py')] + list(argv),
                          env=full_env, capture_output=True, text=True, timeout=600)


def test_bench_gpus_n_launches_its_own_ranks():
    """`python bench.py --gpus 2` with NO launcher around it (the way the driver starts N = 1): the
    parent starts two child ranks before touching any device, rank 0's ONE JSON line comes back on
    the parent's stdout and says two ranks exchanged gradients.  CPU tensors over gloo on the dense
    tiny model -- the launcher and the DP plumbing, not the sparse kernels."""
    import json
    res = _run_bench('--gpus', '2', '--rehearse-cpu', '--config', 'tiny-rehearsal',
                     '--steps', '2', '--warmup', '1')
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [l for l in res.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line['ranks'] == 2 and line['backend'] == 'gloo' and line['steps'] == 2
    assert line['clip_norm_equal_on_all_ranks'] is True
    assert 'rehearsal' in line and line['n_gpus'] == 0      # can never be read as a measurement


def test_bench_launcher_returns_a_failing_ranks_code():
    """A rank that dies takes the others down with it and the parent exits non-zero (no hang in a
    collective, no JSON line)."""
    res = _run_bench('--gpus', '2', '--rehearse-cpu', '--config', 'tiny-rehearsal',
                     '--steps', '1', '--warmup', '0', '--batch', '-1')
    assert res.returncode != 0
    assert not [l for l in res.stdout.splitlines() if l.startswith('{')]


def test_bench_under_a_launcher_still_checks_the_world_size():
    res = _run_bench('--gpus', '2', '--rehearse-cpu', '--config', 'tiny-rehearsal',
                     env={'WORLD_SIZE': '1', 'RANK': '0', 'LOCAL_RANK': '0'})
    assert res.returncode != 0 and 'WORLD_SIZE=1 but --gpus 2' in res.stderr
