"""-DGR_STAMP build of the ring GEMM (tools/variant.sh rstamp grouped_gemm -DGR_STAMP): per k-step, cycles
a loader wave waits for FREE / issues its 12 LDS-DMAs / waits for the previous step to land, and a consumer
waits for FULL (at a tile's first step / at later steps)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, kk, dev = 1024, 1024, 'cuda'
torch.manual_seed(0)
for mt in [64, 512]:
    rows = 128 * mt
    extra = 256 * 8 * 6 * 8 // (d * 4) + 8
    one = torch.tensor([0, rows], dtype=torch.int32, device=dev)
    w = torch.randn([d, kk], device=dev); wi = ext.split_bf16(w)
    a = torch.randn([rows, kk], device=dev); ai = ext.split_bf16(a)
    big = torch.zeros([rows + extra, d], device=dev)
    call = lambda: ext.grouped_gemm_fused(a, w, one, 1, d, kk, 0, kk, 1, rows, a_image=ai, w_image=wi, out=big[:rows])
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    st = big[rows:].flatten().view(torch.int64)[:256 * 8 * 6].view(256, 8, 6).cpu().double()
    cons, load = st[:, :4].reshape(-1, 6), st[:, 4:].reshape(-1, 6)
    steps = cons[:, 5].median()
    c, l = cons.median(0).values, load.median(0).values
    print('%d tiles (128 x 128): %d k-steps per wave, kernel %.0f cycles; per step: total %.0f' % (mt * 8, steps, c[4], c[4] / steps))
    print('   consumer: FULL wait at tile start %.0f, at later steps %.0f' % (c[0] / steps, c[1] / steps))
    print('   loader:   FREE wait %.0f, 12 ds_write + publish %.0f, 12 global loads issued %.0f' % (l[0] / steps, l[1] / steps, l[2] / steps))
