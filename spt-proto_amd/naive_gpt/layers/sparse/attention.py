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
* PQ 'encode' does not materialise the distance tensor;
* on the MI355X fast path (d_head 64 / 128, S * d_head * 4 <= 128 KiB, fp32 CUDA) the
  kernels read q, k, v and write their gradients in the layers' own ``[N, S, H, E]``
  layout (``*_heads`` arguments of the C ABI), so the reference's seven
  ``transpose(1, 2).contiguous()`` copies per call shrink to the one that produces the
  reference's output layout, and sddmm -> clamp -> softmax is one autograd node whose
  backward is a single fused kernel.
"""
import os

import torch

from naive_gpt import ext, kernels, layers

SPARSE_COEFF = 8     # Z = S / 8 keys per query (reference: attention.py:109-117)
CLAMP = 10.0         # reference: attention.py:125-127
# the armed step's PQ loss as a second output of the attention's autograd node (_MfmaAttention);
# False: two separate autograd functions whose q / k gradients autograd sums (tests compare both)
FOLD_PQ_LOSS = True
# q and k that lie back to back in one buffer: their two PQ losses in one pass (spt_pq_loss_*_parts)
PAIR_PQ_LOSS = os.environ.get('SPT_PQ_PAIR', '1') != '0'
# SparseRotaryAttentionV2: the rotary embedding of q and k as one launch (spt_rotary)
ROTARY_KERNEL = os.environ.get('SPT_ROTARY_KERNEL', '1') != '0'


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
            grad_key = ext.spmm_transposed(ext.transposed_for(indptr, indices, query.size(-1)),
                                           indptr, indices, grad_raw, query)
        return None, None, grad_query, grad_key, None


class _HeadScores(torch.autograd.Function):
    """attn = softmax(clamp(scale * sddmm(q, k))) with q, k in [N, S, H, E] layout.
    One autograd node: the backward is spt_softmax_backward_clamped (softmax VJP chained
    through the clamp and the scale) followed by the two gather products."""

    @staticmethod
    def forward(ctx, indptr, indices, q, k, scale: float, heads: int):
        scores = ext.sddmm_forward_cuda(False, True, indptr, indices, q, k, scale=scale,
                                        clamp=CLAMP, query_heads=heads, key_heads=heads)
        attn = ext.softmax_forward_cuda(indptr, indices, scores)
        ctx.scale, ctx.heads = scale, heads
        ctx.save_for_backward(indptr, indices, q, k, scores, attn)
        return attn

    @staticmethod
    def backward(ctx, grad_attn: torch.Tensor):
        indptr, indices, q, k, scores, attn = ctx.saved_tensors
        h = ctx.heads
        grad_raw = ext.softmax_backward_clamped(indptr, indices, attn, grad_attn.contiguous(),
                                                scores, ctx.scale, CLAMP)
        grad_q = grad_k = None
        if ctx.needs_input_grad[2]:
            grad_q = ext.spmm_forward_cuda(False, False, indptr, indices, grad_raw, k,
                                           x_heads=h, y_heads=h)
        if ctx.needs_input_grad[3]:
            grad_k = ext.spmm_transposed(ext.transposed_for(indptr, indices, q.size(-1)), indptr, indices,
                                         grad_raw, q, x_heads=h, y_heads=h)
        return None, None, grad_q, grad_k, None, None


class _HeadSPMM(torch.autograd.Function):
    """y [N*H, S, E] = A . v with v in [N, S, H, E] layout; grad_v comes back in that layout."""

    @staticmethod
    def forward(ctx, indptr, indices, values, v, heads: int):
        ctx.heads = heads
        ctx.save_for_backward(indptr, indices, values, v)
        return ext.spmm_forward_cuda(False, False, indptr, indices, values, v,
                                     x_heads=heads, y_heads=0)

    @staticmethod
    def backward(ctx, grad_y: torch.Tensor):
        indptr, indices, values, v = ctx.saved_tensors
        h = ctx.heads
        grad_y = grad_y.contiguous()
        grad_values = grad_v = None
        if ctx.needs_input_grad[2]:
            grad_values = ext.sddmm_forward_cuda(False, True, indptr, indices, grad_y, v,
                                                 query_heads=0, key_heads=h)
        if ctx.needs_input_grad[3]:
            grad_v = ext.spmm_transposed(ext.transposed_for(indptr, indices, v.size(-1)), indptr, indices,
                                         values, grad_y, x_heads=0, y_heads=h)
        return None, None, grad_values, grad_v, None


class _FusedAttention(torch.autograd.Function):
    """y = spmm(softmax(clamp(scale * sddmm(q, k))), v) in ONE forward launch
    (``spt_sparse_attention_forward``, fused_attention.hip), written straight into the
    reference's output layout (the ``[N*H, E, S]`` memory its ``transpose(1, 2).contiguous()``
    produces, see _sparse_apply).  The backward is ``spt_sparse_attention_backward_rows``
    (grad_q and the score gradients) plus the two transposed products (grad_k, grad_v)."""

    @staticmethod
    def forward(ctx, indptr, indices, q, k, v, scale: float):
        # causal: lookup only emits columns <= row (and column 0 as padding)
        scores, attn, y = ext.sparse_attention_forward(indices, q, k, v, scale, CLAMP,
                                                       y_transposed=True, causal=True)
        ctx.scale = scale
        ctx.save_for_backward(indptr, indices, q, k, v, scores, attn)
        return y.view(q.shape)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor):
        indptr, indices, q, k, v, scores, attn = ctx.saved_tensors
        h = q.size(2)
        # one launch: dP = dY V^T, softmax VJP through the clamp, grad_q = dS K; it reads
        # grad_out in the [N*H, E, S] layout it arrives in and leaves the [N*H, S, E] rows
        # behind for the grad_v product
        grad_raw, grad_q, grad_y = ext.sparse_attention_backward_rows(
            indices, grad_out.contiguous(), v, k, scores, attn, ctx.scale, CLAMP,
            grad_y_transposed=True, causal=True)
        transposed = ext.transposed_for(indptr, indices, q.size(-1))
        grad_v = ext.spmm_transposed(transposed, indptr, indices, attn, grad_y,
                                     x_heads=0, y_heads=h)
        grad_k = ext.spmm_transposed(transposed, indptr, indices, grad_raw, q,
                                     x_heads=h, y_heads=h)
        return None, None, grad_q, grad_k, grad_v, None


class _MfmaAttention(torch.autograd.Function):
    """The same function as _FusedAttention on the matrix cores (mfma_attention.hip): dense
    32 x 32 score tiles from split-bf16 MFMAs, the CSR only selecting the live cells.  Saves
    q, k, v, y and the [N*H, S] row sums -- nothing of size nnz; the backward recomputes the
    tiles in two launches (grad_q; grad_k and grad_v).

    Armed form (`table`, `loss_q`, `loss_k` given: the step's PQ codebook loss,
    attention.py:98-104): the loss of the same q and k is a second OUTPUT of this node -- its
    values were formed with the PQ codes in front of the lookup (ext.pq_loss_forward(want_codes)) --
    so that q and k have ONE consumer and the loss's input gradient is added into the attention's
    grad_q / grad_k by the loss's own backward kernel, inside this node (autograd would add the two
    with an elementwise pass per tensor: 2 x 33 MB read, 33 MB written per q and k and layer).
    Whatever else reads q or k is summed by autograd as usual."""

    @staticmethod
    def forward(ctx, indices, q, k, v, scale: float, table=None, loss_q=None, loss_k=None):
        ctx.armed = table is not None
        # `indices` come straight from kernels.lookup: only its padding column 0 repeats in a
        # row, which is what the compact tile layout needs (70 -> 17 MB per BERT-large layer)
        tiles = ext.attention_mfma_prepare(indices, q.size(1), lookup_pattern=True)
        y, row_sum = ext.attention_mfma_forward(tiles, q, k, v, scale, CLAMP, y_transposed=True)
        ctx.scale, ctx.tiles = scale, tiles
        if ctx.armed:
            ctx.save_for_backward(q, k, v, y, row_sum, table)
            # (loss_k None: loss_q is already the sum of both, from one pass over q and k)
            return y.view(q.shape), (loss_q.clone() if loss_k is None else loss_q + loss_k)
        ctx.save_for_backward(q, k, v, y, row_sum)
        return y.view(q.shape)

    @staticmethod
    def backward(ctx, grad_out: torch.Tensor, grad_loss: torch.Tensor = None):
        q, k, v, y, row_sum = ctx.saved_tensors[:5]
        grad_q, grad_k, grad_v = ext.attention_mfma_backward(
            ctx.tiles, q, k, v, y, grad_out.contiguous(), row_sum, ctx.scale, CLAMP,
            transposed=True)
        if not ctx.armed:
            return None, grad_q, grad_k, grad_v, None
        table = ctx.saved_tensors[5]
        grad_table = None
        if grad_loss is not None:
            grad_loss = grad_loss.contiguous()
            if PAIR_PQ_LOSS and q.dtype == torch.float32 and ext.back_to_back(q, k) \
                    and ext.back_to_back(grad_q, grad_k):
                # (q and k from the joint projection, their gradients from one buffer: one pass)
                ext.note_path('pq_loss_backward', 'pair')
                grad_table = ext.pq_loss_backward_pair(q, k, table, grad_loss, grad_q, grad_k)
                return None, grad_q, grad_k, grad_v, None, grad_table, None, None
            if q.dtype == torch.float32:
                ext.note_path('pq_loss_backward', 'each')
                _, gt_q = ext.pq_loss_backward(q, table, grad_loss, accumulate_into=grad_q)
                _, gt_k = ext.pq_loss_backward(k, table, grad_loss, accumulate_into=grad_k)
            else:
                # bf16 storage: the loss was formed in fp32 from the widened values
                gz_q, gt_q = ext.pq_loss_backward(q.float(), table, grad_loss)
                gz_k, gt_k = ext.pq_loss_backward(k.float(), table, grad_loss)
                grad_q = (grad_q.float() + gz_q).to(q.dtype)
                grad_k = (grad_k.float() + gz_k).to(k.dtype)
            grad_table = gt_q + gt_k
        return None, grad_q, grad_k, grad_v, None, grad_table, None, None


class _SparseCore:
    """Mixin with the CSR attention shared by the Vanilla and Rotary V2 layers."""

    _indptr_cache = {}

    @staticmethod
    def _uniform_indptr(seq_length: int, device: torch.device) -> torch.Tensor:
        key = (seq_length, str(device))
        cached = _SparseCore._indptr_cache.get(key)
        if cached is None:
            top_k = seq_length // SPARSE_COEFF
            # (a tensor created under torch.inference_mode() could not be saved for backward
            # by a later training step: the process-wide cache holds a normal tensor)
            with torch.inference_mode(False):
                cached = torch.arange(0, top_k * seq_length + 1, step=top_k,
                                      dtype=torch.int, device=device)
            _SparseCore._indptr_cache[key] = cached
        return cached

    def arm(self) -> None:
        """Arm the one-shot PQ-loss trigger from the host.  Same effect on the next forward as the
        reference's ``module.trigger.fill_(True)``, as a host-side note only: the forward neither ASKS
        the device (a blocking read: one pipeline drain per layer and step in the tuning recipe, which
        arms every step) nor writes the buffer twice (arm, disarm: two launches per layer and step) --
        the device buffer simply stays disarmed.  The reference's protocol (a write to `trigger`) still
        works: `_take_trigger` sees the buffer's version move."""
        self.__dict__['_armed_hint'] = True

    @property
    def armed(self) -> bool:
        """Will the next forward form the PQ loss?  True after `arm()` OR after a write of True to
        the `trigger` buffer (the reference's protocol, attention.py:98-104).  Reading it costs a
        device read only when the buffer was written since the layer last looked.  `arm()` does NOT
        show in `module.trigger` (or in a state_dict / a broadcast of it): code that inspects the
        buffer after arming must ask here ("the one-shot trigger")."""
        if self.__dict__.get('_armed_hint', False):
            return True
        t = self.trigger
        seen = self.__dict__.get('_trigger_seen')
        if not t.is_inference() and seen is not None and seen[0] == (t.data_ptr(), t._version):
            return bool(seen[1])
        return bool(t.is_nonzero())

    def __deepcopy__(self, memo):
        # `arm()` is an intent of the training loop for THIS module's next forward: a copy (a
        # checkpointed or cloned model) starts disarmed, as a copy of the reference's module whose
        # buffer was armed-and-consumed would; the host-side reading of the buffer is dropped too
        # (the copy's buffer is another tensor)
        import copy
        cls = self.__class__
        new = cls.__new__(cls)
        memo[id(self)] = new
        for key, value in self.__dict__.items():
            if key in ('_armed_hint', '_trigger_seen'):
                continue
            new.__dict__[key] = copy.deepcopy(value, memo)
        return new

    def _take_trigger(self) -> bool:
        """One-shot flag: the training loop arms the device buffer `trigger`, the layer
        disarms it (reference: attention.py:98-104, which reads it with a blocking
        ``.item()`` on every forward).  The buffer's autograd version counter tells whether
        anything wrote to it since the last look, so the device is only asked (one host
        sync) after a write: unarmed steps never synchronise."""
        t = self.trigger
        if self.__dict__.pop('_armed_hint', False):
            return True                           # armed by `arm()`: the device buffer was never set
        if t.is_inference():                      # no version counter to consult
            armed = bool(t.is_nonzero())
            if armed:
                t.logical_not_()
            return armed
        seen = self.__dict__.get('_trigger_seen')
        if seen is not None and seen[0] == (t.data_ptr(), t._version):
            armed = seen[1]
        else:
            armed = bool(t.is_nonzero())
        if armed:
            t.logical_not_()
        self.__dict__['_trigger_seen'] = ((t.data_ptr(), t._version), False)
        return armed

    def _pq_loss(self, q: torch.Tensor, k: torch.Tensor):
        if self._take_trigger():
            loss_q = self.quantizer.train_loss(q)
            loss_k = self.quantizer.train_loss(k)
            self.register_buffer('loss', loss_q + loss_k, persistent=False)

    @staticmethod
    def _mfma_ok(t: torch.Tensor) -> bool:
        n, s, h, e = t.shape
        # (n * h >= 32 as for the other head-layout kernels: below that the launch cannot fill
        # the GPU, and small direct calls of _get_attn keep returning the reference's triple)
        # bf16: the bf16-storage build of the same kernels (BASELINE configs[1] "bf16"); it is the
        # only path that takes bf16 tensors, so there the shape must fit or the layer raises
        return (t.is_cuda and t.dtype in (torch.float32, torch.bfloat16) and s % 4 == 0
                and s % SPARSE_COEFF == 0 and (n * h >= 32 or t.dtype == torch.bfloat16)
                and ext.attention_mfma_supported(s, e, s * (s // SPARSE_COEFF)))

    @staticmethod
    def _head_layout_ok(t: torch.Tensor) -> bool:
        n, s, h, e = t.shape
        if _SparseCore._mfma_ok(t):        # the matrix-core kernels take [N, S, H, E] at any S
            return True
        return (t.is_cuda and t.dtype == torch.float32
                and ext.head_layout_supported(s, e, n * h))

    def _sparse_attn_heads(self, q: torch.Tensor, k: torch.Tensor):
        """MI355X fast path: no layout copies (see module docstring)."""
        seq_length, heads = q.size(1), q.size(2)
        q, k = q.contiguous(), k.contiguous()
        q_c = k_c = None
        armed = self._take_trigger()
        self.__dict__.pop('_armed_losses', None)
        mfma = seq_length % 4 == 0 and ext.attention_mfma_supported(
            seq_length, q.size(-1), seq_length * (seq_length // SPARSE_COEFF))
        if armed:
            # the PQ loss is a mean over all sub-vectors: the head layout holds the same set
            # (bf16 storage: the loss is formed in fp32 from the widened values); its argmin is
            # the PQ code, so an armed step needs no encode pass
            qf, kf = q.float(), k.float()
            if FOLD_PQ_LOSS and mfma and torch.is_grad_enabled() \
                    and ext.pq_loss_supported(qf, self.quantizer.weight):
                # values and codes now; the loss becomes a differentiable output of the attention's
                # own autograd node in _sparse_apply (_MfmaAttention, armed form)
                with torch.no_grad():
                    if PAIR_PQ_LOSS and ext.back_to_back(qf, kf):
                        # q and k of the joint projection lie in one buffer: one pass over both
                        ext.note_path('pq_loss_forward', 'pair')
                        loss_q, q_c, k_c = ext.pq_loss_forward_pair(qf, kf, self.quantizer.weight)
                        loss_k = None
                        self.register_buffer('loss', loss_q, persistent=False)          # (until then)
                    else:
                        ext.note_path('pq_loss_forward', 'each')
                        loss_q, q_c = ext.pq_loss_forward(qf.contiguous(), self.quantizer.weight, want_codes=True)
                        loss_k, k_c = ext.pq_loss_forward(kf.contiguous(), self.quantizer.weight, want_codes=True)
                        self.register_buffer('loss', loss_q + loss_k, persistent=False)
                self.__dict__['_armed_losses'] = (loss_q, loss_k)
            else:
                loss_q, q_c = self.quantizer.train_loss_and_codes(qf)
                loss_k, k_c = self.quantizer.train_loss_and_codes(kf)
                self.register_buffer('loss', loss_q + loss_k, persistent=False)
        table = self.quantizer.weight.detach()
        if q_c is None:
            q_c = ext.pq_encode_heads(q.detach(), table)
        if k_c is None:
            k_c = ext.pq_encode_heads(k.detach(), table)
        topk_indices = kernels.lookup(q_c, k_c, sparse_coeff=SPARSE_COEFF)
        csr_indices = topk_indices.flatten(start_dim=1)
        indptr = self._uniform_indptr(seq_length, q.device)
        if mfma:
            ext.note_path('attention', 'mfma')
            return 'mfma', indptr, csr_indices, q, k
        if ext.fused_attention_supported(seq_length, q.size(-1), q.size(0) * heads,
                                         csr_indices.size(-1)):
            # scores, softmax and the product with v run as one launch in _sparse_apply
            ext.note_path('attention', 'fused_gather', fallback=True,
                          why=lambda: 'q {}: the matrix-core kernels take d_head 64 / 128, S % 32 == 0, S <= 2048'.format(
                              tuple(q.shape)))
            return 'fused', indptr, csr_indices, q, k
        ext.note_path('attention', 'operators_head_layout', fallback=True, why=lambda: 'q {}'.format(tuple(q.shape)))
        values = _HeadScores.apply(indptr, csr_indices, q, k, self.scaling, heads)
        return indptr, csr_indices, values, heads

    def _sparse_attn(self, q: torch.Tensor, k: torch.Tensor):
        # q, k: [N, S, H, E] -> [N*H, S, E]
        assert q.size() == k.size()
        if self._head_layout_ok(q) and self.quantizer.method == 'v2':
            return self._sparse_attn_heads(q, k)
        if q.dtype != torch.float32:
            raise RuntimeError('sparse attention: {} tensors only on the matrix-core path (cuda, '
                               'PQ v2, d_head 64 / 128, S <= 2048, S % 32 == 0)'.format(q.dtype))
        seq_length = q.size(1)
        ext.note_path('attention', 'operators', fallback=q.is_cuda,
                      why=lambda: 'q {} {}, PQ {}: the fast paths take n * h >= 32, d_head 64 / 128, PQ v2'.format(
                          tuple(q.shape), q.dtype, self.quantizer.method))
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
        if len(attn) == 5:
            kind, indptr, indices, q, k = attn
            if kind == 'mfma':
                losses = self.__dict__.pop('_armed_losses', None)
                if losses is None:
                    return _MfmaAttention.apply(indices, q, k, v.contiguous(), self.scaling)
                y, loss = _MfmaAttention.apply(indices, q, k, v.contiguous(), self.scaling,
                                               self.quantizer.weight, losses[0], losses[1])
                self.register_buffer('loss', loss, persistent=False)
                return y
            return _FusedAttention.apply(indptr, indices, q, k, v.contiguous(), self.scaling)
        if len(attn) == 4:
            indptr, indices, values, heads = attn
            y = _HeadSPMM.apply(indptr, indices, values, v.contiguous(), heads)
            # REFERENCE QUIRK, see below
            return y.transpose(1, 2).contiguous().view(v_size)
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

    def forward(self, q, k, v, attn_mask=None):
        cos, sin = self.embedding.cos_cached, self.embedding.sin_cached
        if ROTARY_KERNEL and q.shape == k.shape == v.shape and ext.rotary_supported(q, cos, sin) \
                and ext.rotary_supported(k, cos, sin) and ext.rotary_supported(v, cos, sin):
            # q and k rotated by ONE launch into one buffer (the PQ loss, the lookup and the attention
            # read them as a pair), and in the backward dq, dk, dv leave in one buffer of three
            # equally spaced gradients (the joint projection contracts them as one product)
            ext.note_path('rotary', 'kernel')
            q, k, v = _Rotary3.apply(q, k, v, cos, sin)
            return self._apply_attn(self._sparse_attn(q, k), v)
        ext.note_path('rotary', 'torch', fallback=q.is_cuda,
                      why=lambda: 'q {} {}'.format(tuple(q.shape), q.dtype))
        return super().forward(q, k, v, attn_mask=attn_mask)


class _Rotary3(torch.autograd.Function):
    """(rotary(q), rotary(k), v): position.py:24-34 twice, as one launch each way (spt_rotary)."""

    @staticmethod
    def forward(ctx, q, k, v, cos, sin):
        ctx.save_for_backward(cos, sin)
        out = ext.rotary([q, k], 2, cos, sin)
        return out[0], out[1], v.view_as(v)

    @staticmethod
    def backward(ctx, dq, dk, dv):
        cos, sin = ctx.saved_tensors
        if dq is None or dk is None or dv is None:
            like = next(t for t in (dq, dk, dv) if t is not None)
            dq = torch.zeros_like(like) if dq is None else dq
            dk = torch.zeros_like(like) if dk is None else dk
            dv = torch.zeros_like(like) if dv is None else dv
        out = ext.rotary([dq.contiguous(), dk.contiguous(), dv.contiguous()], 2, cos, sin, transpose=True)
        return out[0], out[1], out[2], None, None
