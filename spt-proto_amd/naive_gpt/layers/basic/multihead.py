"""Stand-alone multi-head wrapper, always with bias
(reference: ``naive_gpt/layers/basic/multihead.py:5-50``; ``TransformerBlock`` uses
its own variant with a ``bias`` switch, see ``transformer.py``)."""
import torch
from torch import nn


def project_heads(module, q, k, v, attn_mask):
    """Shared forward of both MultiheadAttention flavours: project, split heads
    ``[N,S,d] -> [N,S,H,E]``, run the attention core, merge, project out."""
    assert q.size(0) == k.size(0) == v.size(0)
    n_heads = module.n_heads
    fused = False
    if q is k and k is v:
        # self-attention: frozen projections with adapters run as one function of the shared input
        from ..tuning import lora
        linears = (module.linear_q, module.linear_k, module.linear_v)
        from naive_gpt import ext
        joint = lora.qkv_usable(q, linears)
        if all(isinstance(m, lora.LoRALinear) for m in linears):
            ext.note_path('qkv_projection', 'joint' if joint else 'separate', fallback=not joint and q.is_cuda,
                          why=lambda: 'x {} {}, W {}: one grouped launch needs frozen rank-16 adapters, fp32, '
                              '>= 2048 rows, k % 32 == 0'.format(tuple(q.shape), q.dtype,
                                                                 tuple(linears[0].weight.shape)))
        if joint:
            # (a rotary attention replaces q and k by rotated copies: v gets a buffer of its own)
            q, k, v = lora.qkv(q, linears, v_apart=hasattr(module.attn_fn, '_rotate'))
            fused = True
        else:
            lora.prepare_shared(q, linears)      # (at least one pass for the three down products)
    if not fused:
        q = module.linear_q(q)
        k = module.linear_k(k)
        v = module.linear_v(v)
    q = q.view([q.size(0), q.size(1), n_heads, -1])
    k = k.view([k.size(0), k.size(1), n_heads, -1])
    v = v.view([v.size(0), v.size(1), n_heads, -1])
    y: torch.Tensor = module.attn_fn(q, k, v, attn_mask=attn_mask)
    y = y.view([y.size(0), y.size(1), -1])
    return module.linear_o(y)


class MultiheadAttention(nn.Module):
    def __init__(self, d_model: int, n_heads: int, attention_fn: nn.Module):
        super().__init__()
        self.d_model = d_model
        self.n_heads = n_heads
        self.attn_fn = attention_fn
        self.linear_q = nn.Linear(d_model, d_model)
        self.linear_k = nn.Linear(d_model, d_model)
        self.linear_v = nn.Linear(d_model, d_model)
        self.linear_o = nn.Linear(d_model, d_model)

    def forward(self, q, k, v, attn_mask: torch.Tensor = None):
        return project_heads(self, q, k, v, attn_mask)
