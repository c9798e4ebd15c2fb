"""Token -> block routing shared by the routed FFN layers.

The reference evaluates block i on ``x[mask_i]`` with a boolean mask and
scatter-adds with ``y[mask_i] += ...`` inside a Python loop
(``layers/sparse/feedforward.py:66-85``, ``layers/tuning/lora_ffn.py:87-111``): one
``nonzero`` (a device->host sync) per block.  Here the routing is turned once into
a bucket permutation: every token picks exactly k blocks, so the T*k (token, block)
pairs are sorted by block id on the device; bucket i is a contiguous run of that
order.  Only the k*T row count is static -- bucket boundaries are data dependent --
so a single ``bincount`` read-back per call replaces the per-block syncs.
Accumulation order per token is ascending block id, the same as the reference's
loop, so results are bit-identical to it on the same GEMM library.
"""
from typing import Callable, NamedTuple

import torch


class Route(NamedTuple):
    prob: torch.Tensor      # [T, n_blocks] router probabilities
    token: torch.Tensor     # [T*k] token id of each (token, block) pair, sorted by block
    counts: list            # python ints: pairs per block
    n_blocks: int


def route_topk(prob: torch.Tensor, k: int) -> Route:
    n_tokens, n_blocks = prob.shape
    indices = torch.topk(prob, k=k, dim=-1, sorted=False).indices      # [T, k]
    block = indices.reshape(-1)
    order = torch.argsort(block, stable=True)      # ascending block, then token order
    token = torch.div(order, k, rounding_mode='floor')
    counts = torch.bincount(block, minlength=n_blocks).tolist()        # one sync
    return Route(prob=prob, token=token, counts=counts, n_blocks=n_blocks)


def routed_apply(x: torch.Tensor, route: Route,
                 block_fn: Callable[[int, torch.Tensor, torch.Tensor], torch.Tensor],
                 weighted: bool) -> torch.Tensor:
    """y[t] = sum over the blocks i chosen by token t of block_fn(i, x[t], coeff[t, i]);
    coeff = 2 * prob (LoRA variants) when ``weighted``."""
    y = torch.zeros_like(x)
    start = 0
    for i, n in enumerate(route.counts):
        tok = route.token[start:start + n]
        start += n
        x_i = x.index_select(0, tok)
        coeff = None
        if weighted:
            coeff = 2.0 * route.prob.index_select(0, tok)[:, i].unsqueeze(-1)
        y.index_add_(0, tok, block_fn(i, x_i, coeff))
    return y
