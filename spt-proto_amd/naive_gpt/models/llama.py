"""LLaMA (reference: ``naive_gpt/models/llama.py:65-93``): pre-norm blocks with RMSNorm,
bias-free rotary attention, SiLU-gated feed-forward, no learned positions."""
from torch import nn

from naive_gpt import layers
from .decoder import DecoderLM


class LLaMAModel(DecoderLM):
    def __init__(self, d_model: int, n_heads: int, n_layers: int, max_length: int,
                 vocab_size: int, d_feedforward: int, p_dropout: float):
        block = layers.TransformerBlock(
            d_model=d_model, n_heads=n_heads,
            layernorm_fn=layers.LlamaRMSNorm(d_model),
            attention_fn=layers.RotaryAttention(d_head=d_model // n_heads,
                                                p_dropout=p_dropout),
            feedforward_fn=layers.LLaMaFeedforward(d_model=d_model,
                                                   d_feedforward=d_feedforward,
                                                   activation=nn.SiLU()),
            attention_bias=False, pre_norm=True)
        super().__init__(d_model=d_model, n_layers=n_layers, max_length=max_length,
                         vocab_size=vocab_size, block=block,
                         final_norm=layers.LlamaRMSNorm(d_model), learned_positions=False)
