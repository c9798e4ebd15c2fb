"""Grouped GEMM, image path, K = N = 1024, one bucket, at 512 / 1024 / 1536 / 4096 tiles: run under
SPT_HIP_LIBRARY=<ablation variant> (tools/variant.sh ... grouped_gemm -DGG_ABL_*) to see which stream
the k-loop waits for."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, kk, dev = 1024, int(os.environ.get('K', 1024)), 'cuda'
torch.manual_seed(0)
res = []
for mt in [64, 128, 192, 512]:
    rows = 128 * mt
    one = torch.tensor([0, rows], dtype=torch.int32, device=dev)
    w = torch.randn([d, kk], device=dev); wi = ext.split_bf16(w)
    a = torch.randn([rows, kk], device=dev); ai = ext.split_bf16(a)
    call = lambda: ext.grouped_gemm_fused(a, w, one, 1, d, kk, 0, kk, 1, rows, a_image=ai, w_image=wi)
    for _ in range(10):
        call()
    torch.cuda.synchronize()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    ev[0].record()
    for _ in range(30):
        call()
    ev[1].record()
    torch.cuda.synchronize()
    us = ev[0].elapsed_time(ev[1]) * 1e3 / 30
    res.append('%d tiles %.1f us %d TF' % (mt * 8, us, round(6.0 * rows * d * kk / us / 1e6)))
print(os.environ.get('SPT_HIP_LIBRARY', 'shipped').split('_')[-1], ' | '.join(res))
