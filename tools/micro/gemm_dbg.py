"""One small grouped GEMM through the shipped path: where (which 32 x 32 blocks) the result is wrong."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
dev = 'cuda'
torch.manual_seed(0)
rows, kk, n = (int(x) for x in (sys.argv[1:4] if len(sys.argv) > 3 else (256, 64, 256)))
cuts = torch.tensor([0, rows], dtype=torch.int32, device=dev)
w = torch.randn([n, kk], device=dev)
a = torch.randn([rows, kk], device=dev)
out = torch.full([rows, n], -7.0, device=dev)
ext.grouped_gemm_fused(a, w, cuts, 1, n, kk, n * kk, kk, 1, rows, a_image=ext.split_bf16(a),
                       w_image=ext.split_bf16(w), out=out)
torch.cuda.synchronize()
want = a.double() @ w.double().T
err = (out.double() - want).abs()
print('path', ext.LAST_GEMM_PATH, 'untouched', int((out == -7.0).sum()), 'of', out.numel(), 'max err', float(err.max()))
blk = err.view(rows // 32, 32, n // 32, 32).amax((1, 3))
for r in range(min(rows // 32, 8)):
    print(' '.join('%8.1e' % float(v) for v in blk[r][:8]))
if err.max() > 1e-2:
    # does the wrong block hold another block's right values?
    bad = (blk > 1e-2).nonzero()[0].tolist()
    r0, c0 = bad
    got = out[32 * r0:32 * r0 + 32, 32 * c0:32 * c0 + 32].double()
    best = None
    for r in range(rows // 32):
        for c in range(n // 32):
            d = float((got - want[32 * r:32 * r + 32, 32 * c:32 * c + 32]).abs().max())
            if best is None or d < best[0]:
                best = (d, r, c)
    print('block', (r0, c0), 'best matches the true block', best)
    print('got row0', got[0, :6].tolist(), 'want', want[32 * r0, 32 * c0:32 * c0 + 6].tolist())
