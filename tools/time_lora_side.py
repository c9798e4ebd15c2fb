"""spt_lora_down alone (u = x @ table, optionally with x's split image and row norms from the same
pass) against torch.matmul and spt_split_bf16, at the BERT-large step's shapes (read the kernel
times from `tools/gpu.sh stats`: the Python loop here is launch-bound)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext


def timeit(fn, n=50, w=10):
    """GPU time per call from HIP events around n back-to-back calls (the kernels here take
    10-40 us, the launches 3-5: the stream never runs dry)."""
    for _ in range(w):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


res = {}
for rows, K, n in ((8192, 1024, 16), (8192, 1024, 48), (16384, 1024, 16), (16384, 1024, 64), (8192, 4096, 16),
                   (4096, 2752, 64), (2048, 4096, 16)):
    x = torch.randn([rows, K], device='cuda')
    t = torch.randn([K, n], device='cuda')
    d = torch.randn([rows, n], device='cuda')
    tag = '{}x{}x{}'.format(rows, K, n)
    res[tag] = {
        'down_torch': timeit(lambda: torch.matmul(x, t)),
        'down': timeit(lambda: ext.lora_down(x, t)),
        'down_exact': timeit(lambda: ext.lora_down(x, t, exact=True)),
        'down+image': timeit(lambda: ext.lora_down(x, t, want_image=True)),
        'down+image+norms': timeit(lambda: ext.lora_down(x, t, want_image=True, want_norms=True)),
        'split_alone': timeit(lambda: ext.split_bf16(x)),
        'GBps_down': None}
    res[tag]['GBps_down'] = round(rows * K * 4 / res[tag]['down'] / 1e3)
print(json.dumps(res, indent=1))
