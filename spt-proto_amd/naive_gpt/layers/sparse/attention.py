"""PQ-driven sparse attention (reference: ``naive_gpt/layers/sparse/attention.py``).

V1 layers = dense attention + the PQ training loss (they exist so that the
upgrader can learn a codebook before switching to V2).  V2 layers are the hot
path: PQ-encode q and k, pick S/8 causal keys per query from code matches
(``kernels.lookup``), and run attention on that CSR pattern:
``sddmm -> scale, clamp(+-10) -> softmax -> spmm``.

Differences from the reference's orchestration, none of them observable:
* the scale and clamp of ``attention.py:125-127`` are fused into the SDDMM store
  (``spt_sddmm_forward(scale, clamp)``); its gradient mask is applied in the
  backward of the same autograd node;
* the uniform ``indptr`` (``arange * S/8``) is cached per (S, device);
* PQ 'encode' does not materialise the distance tensor.
"""
import torch

from naive_gpt import ext, kernels, layers

SPARSE_COEFF = 8     # Z = S / 8 keys per query (reference: attention.py:109-117)
CLAMP = 10.0         # reference: attention.py:125-127


class _ScaledClampedSDDMM(torch.autograd.Function):
    """values = clamp(scale * sddmm(q, k), -CLAMP, CLAMP), as one kernel; the saved
    output tells the backward which entries were clipped."""

    @staticmethod
    def forward(ctx, indptr, indices, query, key, scale: float):
        out = ext.sddmm_forward_cuda(False, True, indptr, indices, query, key,
                                     scale=scale, clamp=CLAMP)
        ctx.scale = scale
        ctx.save_for_backward(indptr, indices, query, key, out)
        return out

    @staticmethod
    def backward(ctx, grad_output: torch.Tensor):
        indptr, indices, query, key, out = ctx.saved_tensors
        # entries at the rail: scale*raw was >= CLAMP or <= -CLAMP.  torch.clamp's
        # backward keeps the gradient when scale*raw == +-CLAMP exactly; the saved
        # output cannot tell "== rail" from "beyond rail", a measure-zero difference
        # for real inputs (documented in DESIGN.md).
        inside = out.abs() < CLAMP
        grad_raw = torch.where(inside, grad_output * ctx.scale, torch.zeros_like(grad_output))
        grad_raw = grad_raw.contiguous()
        grad_query = grad_key = None
        if ctx.needs_input_grad[2]:
            grad_query = ext.spmm_forward_cuda(False, False, indptr, indices, grad_raw, key)
        if ctx.needs_input_grad[3]:
            grad_key = ext.spmm_transposed(ext.transposed_for(indptr, indices),
                                           indptr, indices, grad_raw, query)
        return None, None, grad_query, grad_key, None


class _SparseCore:
    """Mixin with the CSR attention shared by the Vanilla and Rotary V2 layers."""

    _indptr_cache = {}

    @staticmethod
    def _uniform_indptr(seq_length: int, device: torch.device) -> torch.Tensor:
        key = (seq_length, str(device))
        cached = _SparseCore._indptr_cache.get(key)
        if cached is None:
            top_k = seq_length // SPARSE_COEFF
            cached = torch.arange(0, top_k * seq_length + 1, step=top_k,
                                  dtype=torch.int, device=device)
            _SparseCore._indptr_cache[key] = cached
        return cached

    def _pq_loss(self, q: torch.Tensor, k: torch.Tensor):
        # one-shot: the training loop arms `trigger`, the layer disarms it
        # (reference: attention.py:98-104)
        if self.trigger.is_nonzero():
            self.trigger.logical_not_()
            loss_q = self.quantizer('train', z=q)[-1]
            loss_k = self.quantizer('train', z=k)[-1]
            self.register_buffer('loss', loss_q + loss_k, persistent=False)

    def _sparse_attn(self, q: torch.Tensor, k: torch.Tensor):
        # q, k: [N, S, H, E] -> [N*H, S, E]
        assert q.size() == k.size()
        seq_length = q.size(1)
        q = q.transpose(1, 2).contiguous()
        k = k.transpose(1, 2).contiguous()
        q = q.view([-1, q.size(-2), q.size(-1)])
        k = k.view([-1, k.size(-2), k.size(-1)])

        self._pq_loss(q, k)
        q_c = self.quantizer('encode', z=q)
        k_c = self.quantizer('encode', z=k)

        topk_indices = kernels.lookup(q_c, k_c, sparse_coeff=SPARSE_COEFF)
        csr_indices = topk_indices.flatten(start_dim=1)
        indptr = self._uniform_indptr(seq_length, q.device)

        values = _ScaledClampedSDDMM.apply(indptr, csr_indices, q, k, self.scaling)
        values = kernels.softmax(indptr, csr_indices, values=values)
        return indptr, csr_indices, values

    def _sparse_apply(self, attn, v: torch.Tensor):
        v_size = v.size()
        indptr, indices, values = attn
        v = v.transpose(1, 2).contiguous()
        v = v.view([-1, v.size(-2), v.size(-1)])
        y = kernels.spmm(indptr, indices, values, v)
        # REFERENCE QUIRK, kept for parity (attention.py:140-142): y is [N*H, S, E]
        # here, and the reference "transposes back" with transpose(1, 2) on this 3-D
        # tensor, i.e. it lays the result out as [N*H, E, S] and then *reinterprets*
        # that memory as [N, S, H, E].  Its only layer test feeds all-ones and cannot
        # see it; checkpoints tuned with the reference have linear_o trained on this
        # layout, so a drop-in must reproduce it.  See DESIGN.md "Reference quirks".
        y = y.transpose(1, 2).contiguous()
        return y.view(v_size)


def _make_trigger(module):
    module.trigger: torch.Tensor
    module.register_buffer('trigger', torch.scalar_tensor(False, dtype=torch.bool))


class SparseVanillaAttentionV1(layers.VanillaAttention):
    def __init__(self, d_head: int, p_dropout: float, d_codeword: int,
                 n_codewords: int, n_subspaces: int):
        super().__init__(d_head=d_head, p_dropout=p_dropout)
        self.d_codeword = d_codeword
        self.n_codewords = n_codewords
        self.n_subspaces = n_subspaces
        self.quantizer = layers.PQV1(d_codeword=d_codeword, n_codewords=n_codewords,
                                     n_subspaces=n_subspaces)
        _make_trigger(self)

    def _get_attn(self, q, k, attn_mask):
        loss_q = self.quantizer('train', z=q)[-1]
        loss_k = self.quantizer('train', z=k)[-1]
        self.register_buffer('loss', loss_q + loss_k, persistent=False)
        return layers.VanillaAttention._get_attn(self, q, k, attn_mask=attn_mask)


class SparseVanillaAttentionV2(layers.VanillaAttention, _SparseCore):
    def __init__(self, d_head: int, d_codeword: int, n_codewords: int, p_dropout: float):
        layers.VanillaAttention.__init__(self, d_head=d_head, p_dropout=p_dropout)
        self.quantizer = layers.PQV2(d_codeword=d_codeword, n_codewords=n_codewords,
                                     n_subspaces=d_head // d_codeword)
        _make_trigger(self)

    @staticmethod
    def from_pretrained(source: SparseVanillaAttentionV1):
        assert isinstance(source, SparseVanillaAttentionV1)
        model = layers.SparseVanillaAttentionV2(
            d_head=source.d_head, d_codeword=source.d_codeword,
            n_codewords=source.n_codewords, p_dropout=0.0
        )
        report = model.load_state_dict(source.state_dict(), strict=False)
        if len(report.missing_keys) != 0:
            raise RuntimeError
        return model

    def _get_attn(self, q, k, attn_mask):
        # attn_mask is ignored: causality is structural in lookup (lookup.cu:36-38,51-53)
        return self._sparse_attn(q, k)

    def _apply_attn(self, attn, v):
        return self._sparse_apply(attn, v)


class SparseRotaryAttentionV1(layers.RotaryAttention):
    def __init__(self, d_head: int, p_dropout: float, d_codeword: int,
                 n_codewords: int, n_subspaces: int):
        super().__init__(d_head=d_head, p_dropout=p_dropout)
        self.d_codeword = d_codeword
        self.n_codewords = n_codewords
        self.n_subspaces = n_subspaces
        # the reference ignores `n_subspaces` here (attention.py:161-165)
        self.quantizer = layers.PQV1(d_codeword=d_codeword, n_codewords=n_codewords,
                                     n_subspaces=d_head // d_codeword)
        _make_trigger(self)

    def _get_attn(self, q, k, attn_mask):
        q, k = self._rotate(q), self._rotate(k)
        loss_q = self.quantizer('train', z=q)[-1]
        loss_k = self.quantizer('train', z=k)[-1]
        self.register_buffer('loss', loss_q + loss_k, persistent=False)
        return layers.VanillaAttention._get_attn(self, q, k, attn_mask=attn_mask)


class SparseRotaryAttentionV2(layers.RotaryAttention, _SparseCore):
    def __init__(self, d_head: int, p_dropout: float, d_codeword: int, n_codewords: int):
        layers.RotaryAttention.__init__(self, d_head=d_head, p_dropout=p_dropout)
        self.quantizer = layers.PQV2(d_codeword=d_codeword, n_codewords=n_codewords,
                                     n_subspaces=d_head // d_codeword)
        _make_trigger(self)

    @staticmethod
    def from_pretrained(source: SparseRotaryAttentionV1):
        assert isinstance(source, SparseRotaryAttentionV1)
        model = layers.SparseRotaryAttentionV2(
            d_head=source.d_head, d_codeword=source.d_codeword,
            n_codewords=source.n_codewords, p_dropout=0.0
        )
        report = model.load_state_dict(source.state_dict(), strict=False)
        if len(report.missing_keys) != 0:
            raise RuntimeError
        return model

    def _get_attn(self, q, k, attn_mask):
        return self._sparse_attn(self._rotate(q), self._rotate(k))

    def _apply_attn(self, attn, v):
        return self._sparse_apply(attn, v)
