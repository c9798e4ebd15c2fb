import sys, os, time, cProfile, pstats
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import layers
timer = bench.EventTimer() if os.environ.get('WRAP', '1') == '1' else None
attn = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0).cuda()
q, k, v = [torch.randn([16, 512, 16, 64], device='cuda', requires_grad=True) for _ in range(3)]
params = list(attn.parameters())
def step(trig):
    for t in (q, k, v): t.grad = None
    attn.zero_grad(set_to_none=True)
    bench.sparse_step(attn, q, k, v, trig)
    bench.allreduce_grads(params, 1)
for trig in (False, True, True, True, False, True):
    dt = bench.timed_loop(lambda: step(trig), 20, 5, 1)
    print('trig', trig, 'ms/step %.3f' % (dt / 20 * 1e3), 'reserved MB', torch.cuda.memory_reserved() >> 20, flush=True)
t0 = time.perf_counter()
for i in range(12):
    t1 = time.perf_counter(); step(True); torch.cuda.synchronize(); print('  single step ms %.3f' % ((time.perf_counter() - t1) * 1e3))
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step(True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(12)
