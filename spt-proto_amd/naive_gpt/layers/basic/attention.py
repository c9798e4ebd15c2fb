"""Dense attention cores (reference: ``naive_gpt/layers/basic/attention.py``).

Plumbing only: these are the base classes the sparse layers subclass and the dense
baseline of the benchmark.  Tensors are ``[N, S, H, E]``; a module returns the
attention *object* from ``_get_attn`` and consumes it in ``_apply_attn`` so that
subclasses can swap the representation (dense matrix here, CSR triple in
``layers/sparse/attention.py``).
"""
import torch
from torch import nn

from naive_gpt import layers


class VanillaAttention(nn.Module):
    def __init__(self, d_head: int, p_dropout: float):
        super().__init__()
        self.d_head = d_head
        self.p_dropout = p_dropout
        self.scaling = float(d_head) ** -0.5
        self.dropout = nn.Dropout(p_dropout)

    def _get_attn(self, q: torch.Tensor, k: torch.Tensor, attn_mask: torch.Tensor):
        # scores [N, H, S_q, S_k]; the additive mask is applied before scaling
        # (reference: attention.py:21-31)
        attn = torch.einsum('niae, njae -> naij', q, k)
        if attn_mask is not None:
            attn += attn_mask
        attn = torch.softmax(self.scaling * attn, dim=-1)
        return self.dropout(attn)

    def _apply_attn(self, attn: torch.Tensor, v: torch.Tensor):
        y = torch.einsum('naij, njae -> niae', attn, v)
        return y.contiguous()

    def forward(self, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                attn_mask: torch.Tensor = None):
        assert q.dim() == 4 and k.dim() == 4 and v.dim() == 4
        assert q.size(0) == k.size(0) == v.size(0)
        attn = self._get_attn(q=q, k=k, attn_mask=attn_mask)
        return self._apply_attn(attn, v=v)


class RotaryAttention(VanillaAttention):
    def __init__(self, d_head: int, p_dropout: float, max_length: int = 2048):
        super().__init__(d_head=d_head, p_dropout=p_dropout)
        self.embedding = layers.RotaryEmbedding(
            n_embeddings=max_length, d_model=d_head
        )
        self.cached_ids: torch.Tensor
        self.register_buffer('cached_ids', torch.arange(max_length))
        self.max_length = max_length

    def _rotate(self, x: torch.Tensor):
        return self.embedding(x, ids=self.cached_ids[:x.size(1)])

    def _get_attn(self, q: torch.Tensor, k: torch.Tensor, attn_mask: torch.Tensor):
        return VanillaAttention._get_attn(
            self, self._rotate(q), self._rotate(k), attn_mask=attn_mask
        )
