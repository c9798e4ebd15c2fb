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

from naive_gpt import ext, layers
from . import routing


def _make_router(d_model: int, n_blocks: int):
    return nn.Sequential(nn.Linear(d_model, n_blocks), nn.Sigmoid())


def _grouped_usable(x: torch.Tensor, *frozen) -> bool:
    from . import grouped
    return grouped.usable(x, *frozen)


def _combine(ys: torch.Tensor, bk, x: torch.Tensor) -> torch.Tensor:
    """y[t] = sum of the token's rows of ys, in block order (the reference's `y[mask] += ...` loop
    order): a fixed-order gather (spt_rows_combine) where that applies."""
    if ys.is_cuda and ys.dtype == torch.float32 and ys.size(1) % 4 == 0 and not ys.requires_grad:
        return ext.rows_combine(ys.contiguous(), bk.pos)
    return torch.zeros_like(x).index_add_(0, bk.token_long, ys)


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

    def _forward_grouped(self, x: torch.Tensor):
        """Frozen base weights on the GPU (a stage-'ffn'-only model, inference): the same device-side
        bucketing and grouped matrix-core GEMMs as the LoRA variant (layers/sparse/grouped.py), no
        host synchronisation -- the reference's loop reads every block's row count on the host
        (feedforward.py:66-85).  Trainable weights keep the loop (their gradients are per-bucket
        products this library does not form)."""
        from . import grouped
        nb, bs, d = self.n_blocks, self.block_size, self.d_model
        prob = self.router(x)
        bk = grouped.make_buckets(prob.detach(), k=nb // 2)
        h = grouped.grouped_linear(x, self.fc1.weight, bk.offsets, (bs, d, bs * d, d, 1),
                                   gather=bk.token, gather_long=bk.token_long,
                                   bias=self.fc1.bias.view([nb, bs]))
        ys = grouped.grouped_linear(self.activation(h), self.fc2.weight, bk.offsets,
                                    (d, bs, bs, self.d_feedforward, 1))
        return _combine(ys, bk, x) + self.fc2.bias.view([1, -1])

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        x = x.view([-1, self.d_model])
        if _grouped_usable(x, self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias) \
                and self.d_model % 4 == 0 and self.block_size % 4 == 0:
            ext.note_path('routed_ffn_plain', 'grouped')
            return self._forward_grouped(x.contiguous()).view(x_size)
        ext.note_path('routed_ffn_plain', 'torch_loop', fallback=x.is_cuda,
                      why=lambda: 'x {} {}, trainable weights: {}'.format(
                          tuple(x.shape), x.dtype, self.fc1.weight.requires_grad))
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

    def _forward_grouped(self, x: torch.Tensor):
        """see RoutedFFN._forward_grouped; top-k = n_blocks // 4 (feedforward.py:155-158)"""
        from . import grouped
        nb, bs, d = self.n_blocks, self.block_size, self.d_model
        prob = self.router(x)
        bk = grouped.make_buckets(prob.detach(), k=nb // 4)

        def up(linear):
            return grouped.grouped_linear(x, linear.weight, bk.offsets, (bs, d, bs * d, d, 1),
                                          gather=bk.token, gather_long=bk.token_long)

        h = self.activation(up(self.gate)) * up(self.side)
        ys = grouped.grouped_linear(h, self.down.weight, bk.offsets, (d, bs, bs, self.d_feedforward, 1))
        return _combine(ys, bk, x)

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        x = x.view([-1, self.d_model])
        if self.n_blocks >= 4 and _grouped_usable(x, self.gate.weight, self.side.weight, self.down.weight) \
                and self.d_model % 4 == 0 and self.block_size % 4 == 0:
            ext.note_path('routed_ffn_plain', 'grouped')
            return self._forward_grouped(x.contiguous()).view(x_size)
        ext.note_path('routed_ffn_plain', 'torch_loop', fallback=x.is_cuda,
                      why=lambda: 'x {} {}, trainable weights: {}'.format(
                          tuple(x.shape), x.dtype, self.gate.weight.requires_grad))
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
