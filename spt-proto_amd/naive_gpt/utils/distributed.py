"""Data-parallel gradient exchange for SPT fine-tuning.

The reference trains through Lightning's DDP (``script/4-sparse-tuning-0.py:183-187``):
one process per GPU, NCCL all-reduce of the trainable gradients.  Here the same
exchange is explicit: every rank holds the full frozen replica, only LoRA / router /
PQ-codebook / norm parameters carry gradients (36 MB fp32 at BERT-large dimensions,
SURVEY.md 8e), and they travel as ONE flat fp32 buffer in ONE all-reduce per step --
on MI355X the backend name ``nccl`` is RCCL, which spreads a single large message over
the 7 xGMI links of a GPU, where many small per-parameter messages would be latency
bound.
"""
from typing import Iterable, List, Optional

import torch
import torch.distributed as dist


def trainable_parameters(module: torch.nn.Module) -> List[torch.nn.Parameter]:
    return [p for p in module.parameters() if p.requires_grad]


def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every replica identical to rank ``src`` (parameters and buffers)."""
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            if t.dtype == torch.bool:      # e.g. the `trigger` buffers: not a collective dtype
                tmp = t.to(torch.uint8)
                dist.broadcast(tmp, src=src, group=group)
                # written through `t` itself, not `t.data`: `.data` has a version counter of
                # its own, and the sparse attentions trust `t._version` to tell whether their
                # cached "disarmed" reading of the trigger is still valid
                t.copy_(tmp.to(torch.bool))
            else:
                dist.broadcast(t.data, src=src, group=group)
    # host-side readings of the triggers (layers/sparse/attention.py) describe the OLD values;
    # a pending `module.arm()` is an intent of this rank's training loop, not device state: kept
    for m in module.modules():
        m.__dict__.pop('_trigger_seen', None)


def allreduce_gradients(params: Iterable[torch.nn.Parameter], group=None,
                        world_size: Optional[int] = None, force: bool = False) -> int:
    """Average ``p.grad`` over the ranks of ``group`` with a single flat all-reduce.

    Parameters whose gradient is ``None`` on this rank contribute zeros, so every rank
    must pass the same parameter list (true for replicas).  Returns the number of
    elements exchanged."""
    params = [p for p in params if p.requires_grad]
    if not params:
        return 0
    if world_size is None:
        world_size = dist.get_world_size(group)
    if world_size == 1 and not force:      # force: run the collective anyway (RCCL smoke test)
        return 0
    device = params[0].device
    sizes = [p.numel() for p in params]
    # gather with torch.cat (one launch per 128 tensors, not one copy per parameter), exchange,
    # scatter back with one multi-tensor copy
    zeros = {}
    pieces = []
    for p, n in zip(params, sizes):
        if p.grad is not None:
            pieces.append(p.grad.reshape(-1).to(torch.float32))
        else:
            if n not in zeros:
                zeros[n] = torch.zeros([n], dtype=torch.float32, device=device)
            pieces.append(zeros[n])
    flat = torch.cat(pieces)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    flat.div_(world_size)
    views = [v.view_as(p) for v, p in zip(flat.split(sizes), params)]
    have = [(p.grad, v) for p, v in zip(params, views) if p.grad is not None]
    if have:
        torch._foreach_copy_([g for g, _ in have], [v for _, v in have])
    for p, v in zip(params, views):
        if p.grad is None:
            p.grad = v.clone()
    return int(flat.numel())
