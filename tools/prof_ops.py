"""Run each HIP operator a few times on config-2 shaped inputs (for rocprofv3)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch  # noqa: E402
from naive_gpt import ext  # noqa: E402

ops = sys.argv[1].split(',') if len(sys.argv) > 1 else ['sddmm', 'spmm_n', 'spmm_t', 'lookup', 'cdist', 'softmax']
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, S, H, E = 16, 512, 16, 64
B, M, Z = N * H, E // 8, S // 8
torch.manual_seed(0)
dev = 'cuda'
q = torch.randn([B, S, E], device=dev)
k = torch.randn([B, S, E], device=dev)
table = torch.randn([M, 16, 8], device=dev)


def codes(z):
    zf = z.reshape(B * S, M, 8).transpose(0, 1).contiguous()
    return ext.cdist_encode(zf, table).t().contiguous().view(B, S, M)


qc, kc = codes(q), codes(k)
idx = ext.lookup_forward_cuda(torch.empty([8]), qc, kc).flatten(1)
indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32, device=dev)
vals = torch.rand([B, S * Z], device=dev)
torch.cuda.synchronize()
for _ in range(reps):
    if 'sddmm' in ops:
        ext.sddmm_forward_cuda(False, True, indptr, idx, q, k)
    if 'spmm_n' in ops:
        ext.spmm_forward_cuda(False, False, indptr, idx, vals, k)
    if 'spmm_t' in ops:
        ext.spmm_forward_cuda(True, False, indptr, idx, vals, k)
    if 'transpose' in ops:
        t = ext.csr_transpose(indptr, idx)
        ext.spmm_transposed(t, indptr, idx, vals, k)
    if 'lookup' in ops:
        ext.lookup_forward_cuda(torch.empty([8]), qc, kc)
    if 'cdist' in ops:
        codes(q)
    if 'softmax' in ops:
        y = ext.softmax_forward_cuda(indptr, idx, vals)
        ext.softmax_backward_cuda(indptr, idx, y, vals)
        ext.softmax_backward_clamped(indptr, idx, y, vals, vals, 0.125, 10.0)
    if 'fused' in ops:
        q4 = q.view(N, H, S, E).transpose(1, 2).contiguous()
        ext.sparse_attention_forward(idx, q4, q4, q4, 0.125, 10.0, y_transposed=True, causal=True)
        ext.sparse_attention_forward(idx, q4, q4, q4, 0.125, 10.0, y_transposed=False, causal=True)
        sc, at, yt = ext.sparse_attention_forward(idx, q4, q4, q4, 0.125, 10.0, y_transposed=True, causal=True)
        ext.sparse_attention_backward_rows(idx, yt, q4, q4, sc, at, 0.125, 10.0, grad_y_transposed=True, causal=True)
        ext.sparse_attention_backward_rows(idx, yt.view(B, S, E), q4, q4, sc, at, 0.125, 10.0, grad_y_transposed=False, causal=True)
    if 'mfma' in ops:
        q4 = q.view(N, H, S, E).transpose(1, 2).contiguous()
        tiles = ext.attention_mfma_prepare(idx, S, lookup_pattern=True)      # (the layers' compact tiles)
        yt, rs = ext.attention_mfma_forward(tiles, q4, q4, q4, 0.125, 10.0, y_transposed=True)
        ext.attention_mfma_backward(tiles, q4, q4, q4, yt, yt, rs, 0.125, 10.0, transposed=True)
        qb = q4.bfloat16()                                                    # bf16 storage build
        yb, rsb = ext.attention_mfma_forward(tiles, qb, qb, qb, 0.125, 10.0, y_transposed=True)
        ext.attention_mfma_backward(tiles, qb, qb, qb, yb, yb, rsb, 0.125, 10.0, transposed=True)
    if 'pq_loss' in ops:
        zh = q.view(N, H, S, E)
        ext.pq_encode_heads(zh, table)
        ext.pq_loss_forward(q, table)
        ext.pq_loss_backward(q, table, torch.ones([], device=dev))
    if 'norm' in ops:          # LayerNorm + residual additions (BERT-large stream: [8192, 1024])
        x2, r2 = torch.randn([8192, 1024], device=dev), torch.randn([8192, 1024], device=dev)
        gam, bet = torch.ones([1024], device=dev), torch.zeros([1024], device=dev)
        s2, y2, mu, rs2 = ext.add_layernorm_forward(x2, r2, gam, bet, 1e-5)
        ext.layernorm_backward(s2, y2, gam, mu, rs2, dskip=x2)
    if 'lora' in ops:          # LoRA down product + operand image (+ row norms)
        x2 = torch.randn([8192, 1024], device=dev)
        ext.lora_down(x2, torch.randn([1024, 16], device=dev), want_image=True)
        ext.lora_down(x2, torch.randn([1024, 48], device=dev), want_image=True)
        ext.lora_down(x2, torch.randn([1024, 16], device=dev), want_image=True, want_norms=True)
    if 'ce' in ops:            # LM head loss in place on padded logits
        z2 = torch.randn([8192, 30528], device=dev)
        ext.cross_entropy_grad_(z2, 30522, torch.randint(0, 30522, [8192], device=dev),
                                torch.full([1], 1.0 / 8192, device=dev))
torch.cuda.synchronize()
print('done')
