"""Routed feed-forward networks (reference: ``naive_gpt/layers/sparse/feedforward.py``).

The hidden dimension is cut into ``n_blocks = d_ff / block_size`` blocks; a sigmoid
router scores the blocks per token and only the top-k blocks are evaluated:
k = n_blocks // 2 for ``RoutedFFN`` (``feedforward.py:59-62``), n_blocks // 4 for
``RoutedLLaMaFFN`` (``:155-158``).  The plain variants do not weight the blocks by
the router probability (the LoRA variants in ``tuning/lora_ffn.py`` do).

Evaluation goes through ``routing.routed_apply``: tokens are bucketed per block
(every token appears in exactly k buckets) and each bucket runs a dense block
GEMM; see that module for the MI355X mapping.
"""
import torch
from torch import nn

from naive_gpt import layers
from . import routing


def _make_router(d_model: int, n_blocks: int):
    return nn.Sequential(nn.Linear(d_model, n_blocks), nn.Sigmoid())


class RoutedFFN(layers.Feedforward):
    def __init__(self, d_model: int, d_feedforward: int, block_size: int,
                 activation: nn.Module, p_dropout: float = 0.0):
        super().__init__(d_model, d_feedforward, p_dropout=p_dropout,
                         activation=activation)
        self.block_size = block_size
        assert d_feedforward % block_size == 0
        self.n_blocks = d_feedforward // block_size
        self.router = _make_router(d_model, self.n_blocks)

    @staticmethod
    def from_pretrained(block_size: int, source: layers.Feedforward):
        assert isinstance(source, layers.Feedforward)
        model = RoutedFFN(block_size=block_size, d_model=source.d_model,
                          d_feedforward=source.d_feedforward,
                          activation=source.activation, p_dropout=source.p_dropout)
        report = model.load_state_dict(source.state_dict(), strict=False)
        if len(report.missing_keys) != 2:      # router weight + bias
            raise RuntimeError
        return model

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        x = x.view([-1, self.d_model])
        prob = self.router(x)
        route = routing.route_topk(prob, k=self.n_blocks // 2)
        w1 = self.fc1.weight.view([self.n_blocks, self.block_size, -1])
        b1 = self.fc1.bias.view([self.n_blocks, self.block_size])
        w2 = self.fc2.weight.view([-1, self.n_blocks, self.block_size])

        def block_fn(i, x_i, coeff_i):
            h = self.activation(torch.addmm(b1[i], x_i, w1[i].T))
            return torch.matmul(h, w2[:, i, :].T)

        y = routing.routed_apply(x, route, block_fn, weighted=False)
        y = y + self.fc2.bias.view([1, -1])
        return y.view(x_size)


class RoutedLLaMaFFN(layers.LLaMaFeedforward):
    def __init__(self, d_model: int, d_feedforward: int, block_size: int,
                 activation: nn.Module):
        super().__init__(d_model, d_feedforward, activation)
        self.block_size = block_size
        assert d_feedforward % block_size == 0
        self.n_blocks = d_feedforward // block_size
        self.router = _make_router(d_model, self.n_blocks)

    @staticmethod
    def from_pretrained(block_size: int, source: layers.LLaMaFeedforward):
        assert isinstance(source, layers.LLaMaFeedforward)
        model = RoutedLLaMaFFN(block_size=block_size, d_model=source.d_model,
                               d_feedforward=source.d_feedforward,
                               activation=source.activation)
        report = model.load_state_dict(source.state_dict(), strict=False)
        if len(report.missing_keys) != 2:
            raise RuntimeError
        return model

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        x = x.view([-1, self.d_model])
        prob = self.router(x)
        route = routing.route_topk(prob, k=self.n_blocks // 4)
        gate = self.gate.weight.view([self.n_blocks, self.block_size, -1])
        side = self.side.weight.view([self.n_blocks, self.block_size, -1])
        down = self.down.weight.view([-1, self.n_blocks, self.block_size])

        def block_fn(i, x_i, coeff_i):
            h = self.activation(torch.matmul(x_i, gate[i].T)) * torch.matmul(x_i, side[i].T)
            return torch.matmul(h, down[:, i, :].T)

        y = routing.routed_apply(x, route, block_fn, weighted=False)
        return y.view(x_size)
