"""Transformer block (reference: ``naive_gpt/layers/basic/transformer.py``).

The block owns a private ``MultiheadAttention`` with a ``bias`` switch
(``transformer.py:6-21`` of the reference) and deep copies of the feed-forward and
norm prototypes; pre-norm (OPT/LLaMA) or post-norm (BERT) residual wiring.
"""
import copy

import torch
from torch import nn

from . import fused_norm
from .multihead import project_heads
from ..tuning import recompute


class MultiheadAttention(nn.Module):
    def __init__(self, d_model: int, n_heads: int, attention_fn: nn.Module, bias: bool):
        super().__init__()
        self.d_model = d_model
        self.n_heads = n_heads
        self.attn_fn = attention_fn
        self.linear_q = nn.Linear(d_model, d_model, bias=bias)
        self.linear_k = nn.Linear(d_model, d_model, bias=bias)
        self.linear_v = nn.Linear(d_model, d_model, bias=bias)
        self.linear_o = nn.Linear(d_model, d_model, bias=bias)

    def forward(self, q, k, v, attn_mask: torch.Tensor = None):
        return project_heads(self, q, k, v, attn_mask)


class TransformerBlock(nn.Module):
    def __init__(self, d_model: int, n_heads: int, layernorm_fn: nn.Module,
                 attention_fn: nn.Module, feedforward_fn: nn.Module,
                 attention_bias: bool, pre_norm: bool):
        super().__init__()
        self.pre_norm = pre_norm
        self.mha = MultiheadAttention(
            d_model=d_model, n_heads=n_heads,
            attention_fn=attention_fn, bias=attention_bias
        )
        self.ffd = copy.deepcopy(feedforward_fn)
        self.norm1 = copy.deepcopy(layernorm_fn)
        self.norm2 = copy.deepcopy(layernorm_fn)

    def forward_pair(self, x: torch.Tensor, pending, attn_mask: torch.Tensor = None):
        """The pre-norm block on (residual stream, addend not yet added): returns the same pair --
        every addition of the stream then sits in front of a norm and is fused with it
        (fused_norm.py).  forward(x) == x' + pending' of forward_pair(x, None)."""
        assert x.dim() == 3 and self.pre_norm
        # (the norm outputs are tagged so that the LoRA layers behind them can recompute
        # them in the backward instead of keeping them: layers/tuning/recompute.py)
        x, h = fused_norm.add_norm(self.norm1, x, pending)
        h = recompute.tag(h, self.norm1, x)
        x, h = fused_norm.add_norm(self.norm2, x, self.mha(h, h, h, attn_mask=attn_mask))
        return x, self.ffd(recompute.tag(h, self.norm2, x))

    def forward(self, x: torch.Tensor, attn_mask: torch.Tensor = None):
        assert x.dim() == 3
        if self.pre_norm:
            x, pending = self.forward_pair(x, None, attn_mask=attn_mask)
            return x + pending
        x = self.norm1(x + self.mha(x, x, x, attn_mask=attn_mask))
        return self.norm2(x + self.ffd(x))
