"""LoRA adapters (reference: ``naive_gpt/layers/tuning/lora.py``).

The frozen base layer keeps its ``weight`` / ``bias``; the trainable rank-r side
path lives in ``lora.left.weight [in, r]`` and ``lora.right.weight [out, r]`` (two
``nn.Embedding`` tables, so that the embedding variant can index ``left`` by token
id).  ``right`` starts at zero, so a fresh adapter is the identity.  No 1/r scaling
is applied (``scaling`` is stored but unused, as in the reference).
"""
import torch
from torch import nn


class LoRABase(nn.Module):
    def __init__(self, d_lora: int, in_features: int, out_features: int,
                 device: any = None, dtype: any = None):
        super().__init__()
        self.left = nn.Embedding(in_features, embedding_dim=d_lora,
                                 device=device, dtype=dtype)
        self.right = nn.Embedding(out_features, embedding_dim=d_lora,
                                  device=device, dtype=dtype)
        self.scaling = 1.0 / d_lora
        self.reset_parameters()

    def reset_parameters(self):
        nn.init.zeros_(self.right.weight)


def _freeze(module: nn.Module):
    for param in module.parameters():
        param.requires_grad = False


def _load_base(model: nn.Module, source: nn.Module):
    # only the two LoRA tables may be absent from the source
    report = model.load_state_dict(source.state_dict(), strict=False)
    if len(report.missing_keys) != 2:
        raise RuntimeError
    return model


class LoRALinear(nn.Linear):
    def __init__(self, d_lora: int, in_features: int, out_features: int,
                 bias: bool = True, *args, **kwargs):
        super().__init__(in_features=in_features, out_features=out_features,
                         bias=bias, *args, **kwargs)
        _freeze(self)
        self.lora = LoRABase(d_lora=d_lora, in_features=in_features,
                             out_features=out_features)

    @staticmethod
    def from_pretrained(d_lora: int, source: nn.Linear):
        model = LoRALinear(d_lora=d_lora, in_features=source.in_features,
                           out_features=source.out_features,
                           bias=source.bias is not None)
        return _load_base(model, source)

    def forward(self, x: torch.Tensor):
        y = nn.functional.linear(x, bias=self.bias, weight=self.weight)
        y += torch.matmul(torch.matmul(x, self.lora.left.weight),
                          self.lora.right.weight.T)
        return y


class LoRAEmbedding(nn.Embedding):
    def __init__(self, d_lora: int, num_embeddings: int, embedding_dim: int,
                 *args, **kwargs):
        super().__init__(num_embeddings=num_embeddings,
                         embedding_dim=embedding_dim, *args, **kwargs)
        _freeze(self)
        self.lora = LoRABase(d_lora=d_lora, in_features=num_embeddings,
                             out_features=embedding_dim)

    @staticmethod
    def from_pretrained(d_lora: int, source: nn.Embedding):
        model = LoRAEmbedding(d_lora=d_lora, num_embeddings=source.num_embeddings,
                              embedding_dim=source.embedding_dim)
        return _load_base(model, source)

    def forward(self, x: torch.Tensor):
        side = torch.matmul(self.lora.left(x), self.lora.right.weight.T)
        return nn.functional.embedding(x, weight=self.weight) + side
