python bench.py --steps 20 --warmup 5 --trigger --no-cpu --no-dense 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('trigger', d['ms_per_step'])"
python bench.py --steps 20 --warmup 5 --no-cpu --no-dense 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('plain', d['ms_per_step'], d['with_pq_loss']['ms_per_step'])"
python - <<'PY'
import sys, time, os
sys.path.insert(0, 'spt-proto_amd')
import torch
from naive_gpt import layers
attn = layers.SparseVanillaAttentionV2(d_head=64, d_codeword=8, n_codewords=16, p_dropout=0.0).cuda()
q, k, v = [torch.randn([16, 512, 16, 64], device='cuda', requires_grad=True) for _ in range(3)]
def step(trig):
    for t in (q, k, v): t.grad = None
    attn.zero_grad(set_to_none=True)
    if trig: attn.trigger.fill_(True)
    y = attn(q, k, v, attn_mask=None)
    loss = y.sum()
    if trig: loss = loss + 1e-2 * attn.loss
    loss.backward()
for trig in (False, True):
    for _ in range(5): step(trig)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(20): step(trig)
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print('trig', trig, 'cpu enqueue ms/step %.3f  total ms/step %.3f' % ((t1 - t0) / 20 * 1e3, (t2 - t0) / 20 * 1e3))
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
for _ in range(20): step(True)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
PY
