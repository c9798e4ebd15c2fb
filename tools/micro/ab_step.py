"""A/B of one switch on the configs[2] step inside ONE process (box-to-box and run-to-run variance
is +-2 %, more than most single changes): AB=<module>:<attribute> names a function that is replaced
by `lambda *a, **k: False` for the B legs (e.g. naive_gpt.layers.tuning.lora:qkv_usable), the legs
alternate, 10 timed steps each."""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import utils

dev = torch.device('cuda:0')
mod_name, attr = os.environ['AB'].split(':')
mod = importlib.import_module(mod_name)
orig = getattr(mod, attr)
model = bench.build_model('sparse', dev, int(os.environ.get('LAYERS', bench.LAYERS)))
tuner = utils.SparseTuner(model)
gen = torch.Generator(device=dev).manual_seed(1)


def step():
    batch = torch.randint(3, bench.VOCAB, [16, bench.S + 2], device=dev, generator=gen)
    tuner.training_step(batch, pq_loss=True)


def leg(on, n=10):
    setattr(mod, attr, orig if on else (lambda *a, **k: False))
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


for _ in range(5):
    step()
for r in range(4):
    print('on {:.2f} ms   off {:.2f} ms'.format(leg(True), leg(False)), flush=True)
