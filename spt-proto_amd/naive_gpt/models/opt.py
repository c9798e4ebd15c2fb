"""OPT (reference: ``naive_gpt/models/opt.py:85-111``): pre-norm blocks with LayerNorm,
biased attention projections, ReLU feed-forward, learned positions offset by 2."""
from torch import nn

from naive_gpt import layers
from .decoder import DecoderLM


class OPTModel(DecoderLM):
    position_offset = 2
    PE_OFFSET = position_offset      # the reference's name for it (opt.py:9)

    def __init__(self, d_model: int, n_heads: int, n_layers: int, max_length: int,
                 vocab_size: int, d_feedforward: int, p_dropout: float):
        block = layers.TransformerBlock(
            d_model=d_model, n_heads=n_heads,
            layernorm_fn=nn.LayerNorm(d_model),
            attention_fn=layers.VanillaAttention(d_head=d_model // n_heads,
                                                 p_dropout=p_dropout),
            feedforward_fn=layers.Feedforward(d_model=d_model, d_feedforward=d_feedforward,
                                              activation=nn.ReLU(), p_dropout=p_dropout),
            attention_bias=True, pre_norm=True)
        super().__init__(d_model=d_model, n_layers=n_layers, max_length=max_length,
                         vocab_size=vocab_size, block=block,
                         final_norm=nn.LayerNorm(d_model), learned_positions=True)
