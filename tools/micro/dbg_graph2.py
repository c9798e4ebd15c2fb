"""debug: bench.py's own sequence -- a headline model stepped eagerly, released, then a second model
captured with SparseTuner.capture() and replayed -- with a marker after every phase"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import utils

layers = int(os.environ.get('LAYERS', 2))
opt = os.environ.get('DBG', '')
dev = torch.device('cuda', 0)
N, S, VOCAB = 16, bench.S, bench.VOCAB


def mark(msg):
    torch.cuda.synchronize()
    print(msg, flush=True)


if 'noprior' not in opt:
    model = bench.build_model('sparse', dev, layers)
    tuner = utils.SparseTuner(model)
    gen = torch.Generator(device=dev).manual_seed(1)
    for i in range(3):
        tuner.training_step(torch.randint(3, VOCAB, [N, S + 2], device=dev, generator=gen))
    mark('prior model stepped')
    del model, tuner
    bench._release()
    mark('prior model released')
torch.cuda.empty_cache()
model = bench.build_model('sparse', dev, layers)
tuner = utils.SparseTuner(model)
gen = torch.Generator(device=dev).manual_seed(1)
mark('second model built')
tuner.capture([N, S + 2], pq_loss=True, warmup=int(os.environ.get('WARM', 3)),
              example=torch.randint(3, VOCAB, [N, S + 2], device=dev, generator=gen))
mark('captured')
for i in range(int(os.environ.get('REPLAYS', 4))):
    out = tuner.training_step(torch.randint(3, VOCAB, [N, S + 2], device=dev, generator=gen))
    mark('replay {} {}'.format(i, float(out)))
del model, tuner
mark('deleted')
bench._release()
mark('released')
