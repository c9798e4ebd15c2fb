"""LoRA + routed feed-forward (reference: ``naive_gpt/layers/tuning/lora_ffn.py``).

These are the FFN layers the upgrader installs (``utils/adapter.py``), so they
define parity for the routed-FFN half of the hot path.  With c = 2 * sigmoid(router(x))_i
for a chosen block i:

OPT / BERT  (``lora_ffn.py:87-111``)
    h = act( c * (x W1_i^T + b1_i) + (x L1) R1_i^T )          (the bias is scaled too)
    y += c * (h W2_i) + (h L2_i) R2^T ;   finally y += b2
LLaMA       (``lora_ffn.py:196-222``)
    h = act( c * x G_i^T + (x Lg) Rg_i^T ) * ( c * x S_i^T + (x Ls) Rs_i^T )
    y += c * (h D_i) + (h Ld_i) Rd^T

Both use top-k = n_blocks // 2 (also the LLaMA one, unlike its plain base class).
"""
import torch
from torch import nn

from naive_gpt import layers
from naive_gpt.layers.sparse import routing


def _load_routed(model: nn.Module, source: nn.Module):
    report = model.load_state_dict(source.state_dict(), strict=False)
    if len(report.missing_keys) != 2:      # router weight + bias only: the source
        raise RuntimeError                 # must already be LoRA-upgraded
    return model


class LoRARoutedFFN(layers.RoutedFFN):
    def __init__(self, d_lora: int, block_size: int, d_model: int,
                 d_feedforward: int, activation: nn.Module):
        super().__init__(block_size=block_size, d_model=d_model,
                         d_feedforward=d_feedforward, activation=activation,
                         p_dropout=0.0)
        self.fc1 = layers.LoRALinear(d_lora=d_lora, in_features=d_model,
                                     out_features=d_feedforward)
        self.fc2 = layers.LoRALinear(d_lora=d_lora, in_features=d_feedforward,
                                     out_features=d_model)

    @staticmethod
    def from_pretrained(d_lora: int, block_size: int, source: layers.Feedforward):
        assert isinstance(source, layers.Feedforward)
        model = LoRARoutedFFN(d_lora=d_lora, block_size=block_size,
                              d_model=source.d_model,
                              d_feedforward=source.d_feedforward,
                              activation=source.activation)
        return _load_routed(model, source)

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        x = x.view([-1, self.d_model])
        prob = self.router(x)
        route = routing.route_topk(prob, k=self.n_blocks // 2)

        nb, bs = self.n_blocks, self.block_size
        b1 = self.fc1.bias.view([nb, bs])
        w1 = self.fc1.weight.view([nb, bs, -1])
        r1 = self.fc1.lora.right.weight.view([nb, bs, -1])
        l1 = self.fc1.lora.left.weight
        w2 = self.fc2.weight.view([-1, nb, bs])
        l2 = self.fc2.lora.left.weight.view([nb, bs, -1])
        r2 = self.fc2.lora.right.weight

        def block_fn(i, x_i, coeff):
            h = coeff * torch.addmm(b1[i], x_i, w1[i].T)
            h = h + torch.matmul(torch.matmul(x_i, l1), r1[i].T)
            h = self.activation(h)
            side = torch.matmul(torch.matmul(h, l2[i]), r2.T)
            return side + coeff * torch.matmul(h, w2[:, i, :].T)

        y = routing.routed_apply(x, route, block_fn, weighted=True)
        y = y + self.fc2.bias.view([1, -1])
        return y.view(x_size)


class LoRARoutedLLaMaFFN(layers.RoutedLLaMaFFN):
    def __init__(self, d_lora: int, block_size: int, d_model: int,
                 d_feedforward: int, activation: nn.Module):
        super().__init__(d_model, d_feedforward, block_size=block_size,
                         activation=activation)
        self.gate = layers.LoRALinear(d_lora=d_lora, in_features=d_model,
                                      out_features=d_feedforward, bias=False)
        self.side = layers.LoRALinear(d_lora=d_lora, in_features=d_model,
                                      out_features=d_feedforward, bias=False)
        self.down = layers.LoRALinear(d_lora=d_lora, in_features=d_feedforward,
                                      out_features=d_model, bias=False)

    @staticmethod
    def from_pretrained(d_lora: int, block_size: int, source: layers.LLaMaFeedforward):
        assert isinstance(source, layers.LLaMaFeedforward)
        model = LoRARoutedLLaMaFFN(d_lora=d_lora, block_size=block_size,
                                   d_model=source.d_model,
                                   d_feedforward=source.d_feedforward,
                                   activation=source.activation)
        return _load_routed(model, source)

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        x = x.view([-1, self.d_model])
        prob = self.router(x)
        route = routing.route_topk(prob, k=self.n_blocks // 2)

        nb, bs = self.n_blocks, self.block_size
        wg = self.gate.weight.view([nb, bs, -1])
        rg = self.gate.lora.right.weight.view([nb, bs, -1])
        lg = self.gate.lora.left.weight
        ws = self.side.weight.view([nb, bs, -1])
        rs = self.side.lora.right.weight.view([nb, bs, -1])
        ls = self.side.lora.left.weight
        wd = self.down.weight.view([-1, nb, bs])
        ld = self.down.lora.left.weight.view([nb, bs, -1])
        rd = self.down.lora.right.weight

        def block_fn(i, x_i, coeff):
            h_gate = coeff * torch.matmul(x_i, wg[i].T) \
                + torch.matmul(torch.matmul(x_i, lg), rg[i].T)
            h_side = coeff * torch.matmul(x_i, ws[i].T) \
                + torch.matmul(torch.matmul(x_i, ls), rs[i].T)
            h = self.activation(h_gate) * h_side
            return coeff * torch.matmul(h, wd[:, i, :].T) \
                + torch.matmul(torch.matmul(h, ld[i]), rd.T)

        y = routing.routed_apply(x, route, block_fn, weighted=True)
        return y.view(x_size)
