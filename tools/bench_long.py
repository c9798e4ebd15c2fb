"""Sparse vs dense attention core at long sequence (BASELINE configs[3]: OPT-1.3B dims, S = 2048)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import layers
N, S, H, E = (int(os.environ.get('BATCH', 2)), int(os.environ.get('SEQ', 2048)), 32,
              int(os.environ.get('DHEAD', 64)))
dev = 'cuda'
torch.manual_seed(0)
q, k, v = [torch.randn([N, S, H, E], device=dev, requires_grad=True) for _ in range(3)]
def timeit(fn, n=10, w=3):
    for _ in range(w): fn()
    torch.cuda.synchronize(); torch.cuda.reset_peak_memory_stats(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3, torch.cuda.max_memory_allocated() / 1e9
sparse = layers.SparseVanillaAttentionV2(d_head=E, d_codeword=8, n_codewords=16, p_dropout=0.0).to(dev)   # (M = E / 8 subspaces)
dense = layers.VanillaAttention(d_head=E, p_dropout=0.0).to(dev)
mask = torch.full([S, S], float('-inf'), device=dev).triu(1)
def step(m, mk):
    def f():
        for t in (q, k, v): t.grad = None
        m(q, k, v, attn_mask=mk).sum().backward()
    return f
ms_s, gb_s = timeit(step(sparse, None))
ms_d, gb_d = timeit(step(dense, mask)) if os.environ.get('SPARSE_ONLY') != '1' else (float('nan'), float('nan'))
print(json.dumps({'N': N, 'S': S, 'H': H, 'sparse_ms': ms_s, 'dense_ms': ms_d, 'speedup': ms_d / ms_s,
                  'sparse_gb': gb_s, 'dense_gb': gb_d, 'tokens_per_s_sparse': N * S / ms_s * 1e3}))
