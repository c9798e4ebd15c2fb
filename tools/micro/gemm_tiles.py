"""Grouped GEMM (image path, K = N = 1024, one bucket) against the number of 128 x 128 tiles:
where the time goes between the k-loop, the per-tile fixed cost and the last partial round
(768 workgroup slots = 3 per CU)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d = 1024
dev = 'cuda'
torch.manual_seed(0)
out = {}
ks = [d] if len(sys.argv) < 2 else [int(v) for v in sys.argv[1:]]
for mt in [32, 64, 128, 130, 192, 256]:
    rows = 128 * mt
    one = torch.tensor([0, rows], dtype=torch.int32, device=dev)
    for kk in ks:
        # (K = 1056: image rows of 33 blocks = 4224 bytes, not a power of two)
        w = torch.randn([d, kk], device=dev)
        wi = ext.split_bf16(w)
        a = torch.randn([rows, kk], device=dev)
        ai = ext.split_bf16(a)
        def call():
            return ext.grouped_gemm_fused(a, w, one, 1, d, kk, 0, kk, 1, rows, a_image=ai, w_image=wi)
        for _ in range(5):
            call()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            call()
        ev[1].record()
        torch.cuda.synchronize()
        us = ev[0].elapsed_time(ev[1]) * 1e3 / 20
        tiles = mt * 8
        out['tiles%d_k%d' % (tiles, kk)] = {'us': round(us, 1), 'rounds': round(tiles / 768, 2),
                                            'TF': round(6.0 * rows * d * kk / us / 1e6)}
for k, v in out.items():
    print(k, v)
