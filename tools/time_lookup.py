"""lookup alone at the benchmark shapes: configs[2] (S = 512), configs[3] (S = 2048, M = 8) and
configs[4] (S = 2048, M = 16); codes from the reference's test distribution (test_lookup.py:36-43)."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext


def timeit(fn, n=20, w=5):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


res = {}
cfg = torch.empty([8])
for tag, B, S, M, hi in (('S512_B256_M8', 256, 512, 8, 16), ('S1024_B128_M8', 128, 1024, 8, 16),
                         ('S2048_B64_M8', 64, 2048, 8, 16), ('S2048_B32_M16', 32, 2048, 16, 16),
                         ('S2048_B64_M8_dense', 64, 2048, 8, 2)):
    g = torch.Generator(device='cuda').manual_seed(0)
    q = torch.randint(0, hi, [B, S, M], device='cuda', dtype=torch.int32, generator=g)
    k = torch.randint(0, hi, [B, S, M], device='cuda', dtype=torch.int32, generator=g)
    res[tag] = round(timeit(lambda: ext.lookup_forward_cuda(cfg, q, k)), 1)
print(json.dumps(res))
