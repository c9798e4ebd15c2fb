"""Shared skeleton of the decoder-only language models.

Reference: ``naive_gpt/models/opt.py:8-82`` (``OPTBase``) and
``naive_gpt/models/llama.py:7-62`` (``LLaMABase``) are the same pipeline -- token
embedding (+ learned positions for OPT), ``n_layers`` copies of one
``layers.TransformerBlock``, a final norm, an untied LM head, and an additive causal
mask kept as a buffer.  Parameter / buffer names are the checkpoint format
(``script/1-convert.py:188-195``): ``embedding``, ``learned_pe`` (OPT only),
``decoders.<i>.*``, ``final_norm``, ``lm_output``, ``attn_mask``.
"""
import copy

import torch
from torch import nn


def _has_hooks(module: nn.Module) -> bool:
    """Forward (pre-)hooks or backward hooks on `module`, or registered for every module: the pair
    form below calls `block.forward_pair` directly, which `nn.Module.__call__`'s hooks never see --
    a hooked block (profilers, activation checkpoint wrappers that hook) takes the plain loop."""
    from torch.nn.modules import module as M
    return bool(module._forward_hooks or module._forward_pre_hooks or module._backward_hooks
                or module._backward_pre_hooks or M._global_forward_hooks
                or M._global_forward_pre_hooks or M._global_backward_hooks
                or M._global_backward_pre_hooks)


class DecoderLM(nn.Module):
    #: OPT stores positions shifted by two (a fairseq padding convention); 0 = no
    #: learned positions at all (LLaMA: rotary, inside the attention)
    position_offset = 0

    def __init__(self, d_model: int, n_layers: int, max_length: int, vocab_size: int,
                 block: nn.Module, final_norm: nn.Module, learned_positions: bool):
        super().__init__()
        self.embedding = nn.Embedding(vocab_size, embedding_dim=d_model)
        mask_length = max_length
        if learned_positions:
            mask_length = max_length + self.position_offset
            self.learned_pe = nn.Embedding(mask_length, embedding_dim=d_model)
        self.decoders = nn.ModuleList(copy.deepcopy(block) for _ in range(n_layers))
        self.final_norm = final_norm
        self.lm_output = nn.Linear(d_model, vocab_size, bias=False)
        # 0 on and below the diagonal, -inf above it
        causal = torch.full([mask_length, mask_length], float('-inf')).triu(diagonal=1)
        self.attn_mask: torch.Tensor
        self.register_buffer('attn_mask', causal)

    def embed(self, tokens: torch.Tensor) -> torch.Tensor:
        h = self.embedding(tokens)
        if hasattr(self, 'learned_pe'):
            positions = torch.arange(tokens.size(-1), device=tokens.device)
            h = h + self.learned_pe(positions + self.position_offset).unsqueeze(0)
        return h

    def hidden(self, x: torch.Tensor) -> torch.Tensor:
        """tokens [batch, seq] -> the LM head's input [batch, seq, d_model]."""
        assert x.dim() == 2
        seq_length = x.size(-1)
        mask = self.attn_mask[:seq_length, :seq_length]
        h = self.embed(x)
        if all(getattr(block, 'pre_norm', False) and hasattr(block, 'forward_pair')
               and not _has_hooks(block) for block in self.decoders):
            # pre-norm stack as (stream, pending addend) pairs: each residual addition is fused
            # with the norm behind it, the last one with the final norm (layers/basic/fused_norm.py)
            from naive_gpt.layers.basic import fused_norm
            pending = None
            for block in self.decoders:
                h, pending = block.forward_pair(h, pending, attn_mask=mask)
            return fused_norm.add_norm(self.final_norm, h, pending)[1]
        for block in self.decoders:
            h = block(h, attn_mask=mask)
        return self.final_norm(h)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """tokens [batch, seq] (integer) -> logits [batch, seq, vocab]."""
        return self.lm_output(self.hidden(x))
