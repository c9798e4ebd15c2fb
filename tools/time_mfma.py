"""Time the matrix-core attention kernels against the gather kernels at the bench shape."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch  # noqa: E402
from naive_gpt import ext  # noqa: E402

N, S, H, E = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 512, 16, 64)))
B, M, Z = N * H, E // 8, min(64, S // 8)
torch.manual_seed(0)
dev = 'cuda'
q, k, v = [torch.randn([N, S, H, E], device=dev) for _ in range(3)]
table = torch.randn([M, 16, 8], device=dev)
qc, kc = ext.pq_encode_heads(q, table), ext.pq_encode_heads(k, table)
idx = ext.lookup_forward_cuda(torch.empty([S // Z]), qc, kc).flatten(1)


def timeit(fn, reps=20):
    for _ in range(3):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


res = {}
res['mfma_prepare_us'] = timeit(lambda: ext.attention_mfma_prepare(idx, S))
tiles = ext.attention_mfma_prepare(idx, S)
res['mfma_fwd_us'] = timeit(lambda: ext.attention_mfma_forward(tiles, q, k, v, 0.125, 10.0, y_transposed=True))
if ext.fused_attention_supported(B, S, E, idx.size(-1)) and "SPT_HIP_LIBRARY" not in os.environ:
    res['gather_fwd_us'] = timeit(lambda: ext.sparse_attention_forward(idx, q, k, v, 0.125, 10.0, y_transposed=True, causal=True))
    sc, at, y0 = ext.sparse_attention_forward(idx, q, k, v, 0.125, 10.0, y_transposed=True, causal=True)
    y1, rs = ext.attention_mfma_forward(tiles, q, k, v, 0.125, 10.0, y_transposed=True)
    res['max_rel_diff_y'] = float((y1 - y0).abs().max() / y0.abs().max())
if hasattr(ext, 'attention_mfma_backward'):
    y1, rs = ext.attention_mfma_forward(tiles, q, k, v, 0.125, 10.0, y_transposed=True)
    gy = torch.randn_like(y1)
    res['mfma_bwd_us'] = timeit(lambda: ext.attention_mfma_backward(tiles, q, k, v, y1, gy, rs, 0.125, 10.0, transposed=True))
if os.environ.get('MA_STAMPS') == '1':
    _, rs = ext.attention_mfma_forward(tiles, q, k, v, 0.125, 10.0, y_transposed=True)
    st = rs.flatten()[:320].cpu().view(2, 8, 20)
    for wg in range(2):
        for w in range(8):
            row = st[wg, w]
            print('wg', wg, 'wave', w, 'rt', int(row[19]), 'T', int(row[18]), 'end', int(row[17]),
                  'iter ends', [int(x) for x in row[:17]])
if os.environ.get('MA_STAMPS') == '1' and hasattr(ext, 'attention_mfma_backward'):
    # row-owned backward, workgroup 0: marks = per pass [prologue, iteration ends..., loop, pass]
    gq = ext.attention_mfma_backward(tiles, q, k, v, y1, gy, rs, 0.125, 10.0, transposed=True)[0]
    st = gq[0, :4, 0, :].flatten().cpu().view(8, 32)
    for w in range(8):
        print('bwd rows wave', w, [int(x) for x in st[w] if x > 0])
print(res)
