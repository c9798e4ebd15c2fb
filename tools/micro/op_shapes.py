"""Which aten ops (by input shape) the configs[2] step spends its elementwise / reduction time in:
torch.profiler over a few steps of an N-layer sparse model (record_shapes), GPU time per (op, shapes)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import utils
from torch.profiler import profile, ProfilerActivity

dev = torch.device('cuda:0')
layers = int(os.environ.get('LAYERS', 4))
model = bench.build_model('sparse', dev, layers)
tuner = utils.SparseTuner(model)
gen = torch.Generator(device=dev).manual_seed(1)


def step():
    batch = torch.randint(3, bench.VOCAB, [16, bench.S + 2], device=dev, generator=gen)
    tuner.training_step(batch, pq_loss=True)


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True) as prof:
    for _ in range(2):
        step()
    torch.cuda.synchronize()
rows = []
for e in prof.key_averages(group_by_input_shape=True):
    dt = getattr(e, 'self_device_time_total', None)
    if dt is None:
        dt = e.self_cuda_time_total
    if dt > 0:
        rows.append((dt / 2, e.count / 2, e.key, str(e.input_shapes)[:110]))
rows.sort(reverse=True)
print('layers', layers, ' (per step)')
for dt, cnt, key, shapes in rows[:60]:
    print('{:9.1f} us {:7.1f} x  {:40s} {}'.format(dt, cnt, key[:40], shapes))
