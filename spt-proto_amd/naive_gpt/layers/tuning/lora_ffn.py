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

from naive_gpt import ext, layers
from naive_gpt.layers.sparse import grouped, routing
from . import recompute


def _plain_router(router) -> bool:
    """The router as `sparse/feedforward.py:_make_router` builds it (Linear + Sigmoid): what the
    fused function restates inside its own autograd node."""
    return (isinstance(router, nn.Sequential) and len(router) == 2
            and type(router[0]) is nn.Linear and type(router[1]) is nn.Sigmoid
            and router[0].bias is not None)


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

    def _router(self, x: torch.Tensor, origin):
        """router probabilities; a tagged norm output is not kept for the router's weight
        gradient either (layers/tuning/recompute.py)"""
        if origin is None:
            return self.router(x)
        linear = self.router[0]
        return self.router[1](recompute.RecomputedLinear.apply(
            x, origin.input, linear.weight, linear.bias, origin.module))

    def _forward_grouped(self, x: torch.Tensor, origin=None):
        """MI355X path: device-side bucketing + grouped MFMA GEMMs for the frozen base
        weights (layers/sparse/grouped.py); LoRA side paths as dense K = n_blocks * r
        matmuls on the same row space.  No host synchronisation."""
        nb, bs, d, r = self.n_blocks, self.block_size, self.d_model, self.fc1.lora.left.weight.size(1)
        act = grouped.activation_code(self.activation)
        if act is not None and grouped.fused_usable(r) and self.fc2.bias is not None \
                and not self.fc2.bias.requires_grad and _plain_router(self.router):
            # four fused block GEMMs, the router and the bucketing of its top-k as ONE autograd
            # node (layers/sparse/grouped.py: RoutedLoRAFFN); coeff = 2 prob (lora_ffn.py:96)
            ext.note_path('routed_ffn', 'fused')
            return grouped.routed_lora_ffn(
                x, self.router[0].weight, self.router[0].bias,
                self.fc1.lora.left.weight, self.fc1.lora.right.weight,
                self.fc2.lora.left.weight, self.fc2.lora.right.weight,
                self.fc1.weight, self.fc1.bias, self.fc2.weight, self.fc2.bias, nb, nb // 2, act,
                origin=origin)
        prob = self._router(x, origin)
        bk = grouped.make_buckets(prob, k=nb // 2, scale=2.0)       # coeff = 2 prob (lora_ffn.py:96)
        coeff = bk.coeff
        ext.note_path('routed_ffn', 'grouped_linear', fallback=True,
                      why=lambda: 'activation {} rank {}: the fused epilogues take ReLU / GELU / SiLU, '
                          'rank % 4 == 0, rank <= 32, a frozen fc2.bias'.format(type(self.activation).__name__, r))
        rows = torch.arange(bk.block.numel(), device=x.device)

        g1 = grouped.grouped_linear(
            x, self.fc1.weight, bk.offsets, (bs, d, bs * d, d, 1), gather=bk.token,
            gather_long=bk.token_long, bias=self.fc1.bias.view([nb, bs]), rowscale=coeff)
        # (x L1) R1_g^T as one dense product: the row's r coefficients sit in slice g
        u = torch.matmul(x, self.fc1.lora.left.weight).index_select(0, bk.token_long)
        u_exp = torch.zeros([u.size(0), nb, r], dtype=u.dtype, device=u.device)
        u_exp[rows, bk.block] = u
        r1cat = self.fc1.lora.right.weight.view([nb, bs, r]).permute(1, 0, 2).reshape(bs, nb * r)
        h = self.activation(g1 + torch.matmul(u_exp.view(-1, nb * r), r1cat.T))

        y2 = grouped.grouped_linear(
            h, self.fc2.weight, bk.offsets, (d, bs, bs, self.d_feedforward, 1), rowscale=coeff)
        l2cat = self.fc2.lora.left.weight.view([nb, bs, r]).permute(1, 0, 2).reshape(bs, nb * r)
        z = torch.matmul(h, l2cat).view(-1, nb, r)[rows, bk.block]
        ys = y2 + torch.matmul(z, self.fc2.lora.right.weight.T)
        y = torch.zeros_like(x).index_add_(0, bk.token_long, ys)
        return y + self.fc2.bias.view([1, -1])

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        origin = recompute.origin_of(x) if x.is_contiguous() else None
        x = x.view([-1, self.d_model])
        if grouped.usable(x, self.fc1.weight, self.fc1.bias, self.fc2.weight) \
                and self.d_model % 4 == 0 and self.block_size % 4 == 0:
            return self._forward_grouped(x.contiguous(), origin).view(x_size)
        ext.note_path('routed_ffn', 'torch_loop', fallback=x.is_cuda,
                      why=lambda: 'x {} {}: the grouped GEMMs take fp32 CUDA activations and frozen base weights'.format(
                          tuple(x.shape), x.dtype))
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

    _router = LoRARoutedFFN._router

    def _forward_grouped(self, x: torch.Tensor, origin=None):
        """MI355X path, see LoRARoutedFFN._forward_grouped."""
        nb, bs, d, r = self.n_blocks, self.block_size, self.d_model, self.gate.lora.left.weight.size(1)
        if grouped.fused_usable(r) and _plain_router(self.router):
            # three fused block GEMMs forward, router and bucketing inside
            # (layers/sparse/grouped.py: RoutedLoRALLaMAFFN)
            ext.note_path('routed_ffn', 'fused')
            return grouped.routed_lora_llama_ffn(
                x, self.router[0].weight, self.router[0].bias,
                self.gate.lora.left.weight, self.gate.lora.right.weight,
                self.side.lora.left.weight, self.side.lora.right.weight,
                self.down.lora.left.weight, self.down.lora.right.weight,
                self.gate.weight, self.side.weight, self.down.weight, nb, nb // 2, self.activation,
                origin=origin)
        prob = self._router(x, origin)
        bk = grouped.make_buckets(prob, k=nb // 2, scale=2.0)       # coeff = 2 prob (lora_ffn.py:96)
        coeff = bk.coeff
        ext.note_path('routed_ffn', 'grouped_linear', fallback=True, why=lambda: 'rank {}'.format(r))
        rows = torch.arange(bk.block.numel(), device=x.device)

        def up(linear):
            base = grouped.grouped_linear(
                x, linear.weight, bk.offsets, (bs, d, bs * d, d, 1), gather=bk.token,
                gather_long=bk.token_long, rowscale=coeff)
            u = torch.matmul(x, linear.lora.left.weight).index_select(0, bk.token_long)
            u_exp = torch.zeros([u.size(0), nb, r], dtype=u.dtype, device=u.device)
            u_exp[rows, bk.block] = u
            rcat = linear.lora.right.weight.view([nb, bs, r]).permute(1, 0, 2).reshape(bs, nb * r)
            return base + torch.matmul(u_exp.view(-1, nb * r), rcat.T)

        h = self.activation(up(self.gate)) * up(self.side)
        y2 = grouped.grouped_linear(
            h, self.down.weight, bk.offsets, (d, bs, bs, self.d_feedforward, 1), rowscale=coeff)
        ldcat = self.down.lora.left.weight.view([nb, bs, r]).permute(1, 0, 2).reshape(bs, nb * r)
        z = torch.matmul(h, ldcat).view(-1, nb, r)[rows, bk.block]
        ys = y2 + torch.matmul(z, self.down.lora.right.weight.T)
        return torch.zeros_like(x).index_add_(0, bk.token_long, ys)

    def forward(self, x: torch.Tensor):
        x_size = x.size()
        origin = recompute.origin_of(x) if x.is_contiguous() else None
        x = x.view([-1, self.d_model])
        if grouped.usable(x, self.gate.weight, self.side.weight, self.down.weight) \
                and self.d_model % 4 == 0 and self.block_size % 4 == 0:
            return self._forward_grouped(x.contiguous(), origin).view(x_size)
        ext.note_path('routed_ffn', 'torch_loop', fallback=x.is_cuda,
                      why=lambda: 'x {} {}: the grouped GEMMs take fp32 CUDA activations and frozen base weights'.format(
                          tuple(x.shape), x.dtype))
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
