"""What the partial last round of a ragged grouped GEMM costs: P = 16384 rows in 4 buckets,
K = N = 1024, image path -- balanced buckets (1024 tiles = two full rounds of 512 slots) against
ragged ones (row tiles 129 .. 132: 8 .. 32 tiles in a third, halved round)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, kk, dev = 1024, 1024, 'cuda'
torch.manual_seed(0)
rows = 16384
w = torch.randn([4, d, kk], device=dev); wi = ext.split_bf16(w.view(4 * d, kk))
a = torch.randn([rows, kk], device=dev); ai = ext.split_bf16(a)
CASES = [('balanced', [4096] * 4), ('ragged +1', [4096 + 64, 4096 - 64, 4096, 4096]),
                    ('ragged +2', [4000, 4200, 4100, 4084]), ('ragged +4', [3900, 4300, 4150, 4034]),
                    ('ragged +3 (one short)', [4097, 4097, 4097, 4093])]
for name, sizes in CASES + CASES:     # (twice: the first pass also warms the clocks)
    off = torch.tensor([0] + list(torch.tensor(sizes).cumsum(0)), dtype=torch.int32, device=dev)
    tiles = sum((s + 127) // 128 for s in sizes) * 8
    call = lambda: ext.grouped_gemm_fused(a, w, off, 4, d, kk, d * kk, kk, 1, rows, a_image=ai, w_image=wi)
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(50):
        call()
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / 50
    print('%-22s %4d tiles (%d beyond two rounds)  %.1f us  %d TF executed' % (
        name, tiles, tiles - 1024, us, round(6.0 * rows * d * kk / us / 1e6)))
