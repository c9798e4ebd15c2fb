"""spt_lora_down alone (u = x @ table, optionally with x's split image and row norms from the same
pass) against torch.matmul and spt_split_bf16, at the BERT-large step's shapes (read the kernel
times from `tools/gpu.sh stats`: the Python loop here is launch-bound)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext


def timeit(fn, n=50, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return round((time.perf_counter() - t0) / n * 1e6, 1)


res = {}
for rows, K, n in ((8192, 1024, 16), (8192, 1024, 48), (16384, 1024, 16), (16384, 1024, 64), (8192, 4096, 16)):
    x = torch.randn([rows, K], device='cuda')
    t = torch.randn([K, n], device='cuda')
    d = torch.randn([rows, n], device='cuda')
    tag = '{}x{}x{}'.format(rows, K, n)
    res[tag] = {
        'down_torch': timeit(lambda: torch.matmul(x, t)),
        'down': timeit(lambda: ext.lora_down(x, t)),
        'down+image+norms': timeit(lambda: ext.lora_down(x, t, want_image=True, want_norms=True)),
        'split_alone': timeit(lambda: ext.split_bf16(x)),
        'GBps_down': None}
    res[tag]['GBps_down'] = round(rows * K * 4 / res[tag]['down'] / 1e3)
print(json.dumps(res, indent=1))
