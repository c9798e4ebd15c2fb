"""Grouped GEMM micro-benchmark at the routed-FFN shapes (P = 16384 rows, 4 buckets)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, dff, T = 1024, 4096, 8192
P, bs = 2 * T, dff // 4
dev = 'cuda'
torch.manual_seed(0)
def timeit(fn, n=20, w=5):
    for _ in range(w): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
# PAD=<floats>: leading dimension d + PAD for A and W (rows no longer 4 KiB apart)
pad = int(os.environ.get('PAD', 0))
a = torch.randn([T, d + pad], device=dev)[:, :d]
gather = (torch.randperm(P, device=dev) % T).int()
even = os.environ.get('EVEN', '0') == '1'
cuts = [0, P // 4, P // 2, 3 * P // 4, P] if even else [0, 4000, 8300, 12100, P]
offsets = torch.tensor(cuts, dtype=torch.int32, device=dev)
w1 = torch.randn([dff, d], device=dev)
w1p = torch.randn([dff, d + pad], device=dev)
h = torch.randn([P, bs], device=dev)
flops = 2.0 * P * bs * d
res = {}
ms = timeit(lambda: ext.grouped_gemm(a, w1p, offsets, 4, bs, d, bs * (d + pad), d + pad, 1, gather=gather))
res['bt'] = (round(ms * 1e3, 1), round(flops / ms / 1e9, 1))
ms = timeit(lambda: ext.grouped_gemm(h, w1, offsets, 4, d, bs, bs * d, 1, d))
res['bn'] = (round(ms * 1e3, 1), round(flops / ms / 1e9, 1))
b = torch.randn([P, d], device=dev); wd = torch.randn([bs, d], device=dev)
ms = timeit(lambda: torch.matmul(b, wd.T))
res['torch'] = (round(ms * 1e3, 1), round(flops / ms / 1e9, 1))
print(os.environ.get('SPT_HIP_LIBRARY', 'default').split('/')[-1], 'pad', pad, json.dumps(res))
