"""Rotary position embedding (reference: ``naive_gpt/layers/basic/position.py:5-48``).

Tables are built on the host once (cos/sin of t * base^(-2i/d), duplicated over the
two halves) and kept as buffers ``cos_cached`` / ``sin_cached`` -- the state_dict
keys of the reference.
"""
import torch
from torch import nn


class RotaryEmbedding(nn.Module):
    def __init__(self, n_embeddings: int, d_model: int, base: float = 10000.0):
        super().__init__()
        assert d_model % 2 == 0
        exponent = torch.arange(0, d_model, step=2) / d_model
        inv_freq = 1.0 / (base ** exponent)
        angle = torch.outer(torch.arange(n_embeddings), inv_freq)   # [S, E/2]
        angle = torch.cat((angle, angle), dim=-1)                   # [S, E]
        self.cos_cached: torch.Tensor
        self.sin_cached: torch.Tensor
        self.register_buffer('cos_cached', angle.cos())
        self.register_buffer('sin_cached', angle.sin())

    def rotate_half(self, x: torch.Tensor):
        lo, hi = x.chunk(2, dim=-1)
        return torch.cat([-hi, lo], dim=-1)

    def forward(self, x: torch.Tensor, ids: torch.Tensor):
        # x [N, S, H, E], ids [S]
        assert x.dim() == 4
        assert ids.dim() == 1
        cos = self.cos_cached[ids][None, :, None, :]
        sin = self.sin_cached[ids][None, :, None, :]
        return cos * x + sin * self.rotate_half(x)
