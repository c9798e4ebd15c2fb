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
