"""rocprofv3 target: the configs[2] step replayed as one HIP graph (SparseTuner.capture), 10 replays
-- to compare its kernels' durations with the eager step's (tools/gpu.sh stats ... bench.py --only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import utils

dev = torch.device('cuda:0')
model = bench.build_model('sparse', dev, int(os.environ.get('LAYERS', bench.LAYERS)))
tuner = utils.SparseTuner(model)
gen = torch.Generator(device=dev).manual_seed(1)
example = torch.randint(3, bench.VOCAB, [16, bench.S + 2], device=dev, generator=gen)
tuner.capture(example.shape, pq_loss=True, example=example)
for _ in range(10):
    tuner.training_step(torch.randint(3, bench.VOCAB, [16, bench.S + 2], device=dev, generator=gen))
torch.cuda.synchronize()
