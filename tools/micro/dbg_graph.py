"""debug: the capture leg of bench.py step by step (eager on a side stream, capture, replay)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import utils

layers = int(os.environ.get('LAYERS', 2))
opt = os.environ.get('DBG', '')
if 'nofold' in opt:
    from naive_gpt.layers.sparse import attention as A
    A.FOLD_PQ_LOSS = False
if 'norouter' in opt:
    from naive_gpt.layers.tuning import lora_ffn
    lora_ffn._plain_router = lambda r: False
if 'nohead' in opt:
    from naive_gpt.layers.tuning import head_loss
    head_loss.fused_usable = lambda *a: False
if 'nopq' in opt:
    PQ = False
else:
    PQ = True
if 'nodown' in opt:
    from naive_gpt import ext as _e
    _e.lora_down_supported = lambda *a: False
    from naive_gpt.layers.tuning import lora as _l
    _l.qkv_usable = lambda *a: False
if 'noflat' in opt:
    utils.SparseTuner.FLAT_LIMIT = 0
if 'nopqkernel' in opt:
    from naive_gpt import ext as _e2
    _e2.pq_loss_supported = lambda *a: False
if 'nocheck' in opt:
    utils.SparseTuner._check_flat_views = lambda self: None
print('variant', opt, flush=True)
dev = torch.device('cuda', 0)
model = bench.build_model('sparse', dev, layers)
tuner = utils.SparseTuner(model)
gen = torch.Generator(device=dev).manual_seed(1)
N, S, VOCAB = 16, bench.S, bench.VOCAB
batch = lambda: torch.randint(3, VOCAB, [N, S + 2], device=dev, generator=gen)
if 'nomain' not in opt:
    for i in range(2):
        tuner.training_step(batch(), pq_loss=PQ)
    torch.cuda.synchronize(); print('eager main stream ok', flush=True)
tuner.use_capturable_optimizer()
if 'nomain' not in opt:
    for i in range(2):
        tuner._eager_step(batch(), PQ)
    torch.cuda.synchronize(); print('capturable optimizer eager ok', flush=True)
side = torch.cuda.Stream(dev)
side.wait_stream(torch.cuda.current_stream(dev))
with torch.cuda.stream(side):
    for _ in range(int(os.environ.get('WARM', 2))):
        tuner._eager_step(batch(), PQ)
torch.cuda.current_stream(dev).wait_stream(side)
torch.cuda.synchronize(); print('eager side stream ok', flush=True)
if 'droploss' in opt:
    import gc
    for m in model.modules():
        m._buffers.pop('loss', None)
    gc.collect()
tuner._graph_batch = batch()
graph = torch.cuda.CUDAGraph()
with torch.cuda.graph(graph):
    loss = tuner._eager_step(tuner._graph_batch, PQ)
torch.cuda.synchronize(); print('capture ok', flush=True)
for i in range(int(os.environ.get('REPLAYS', 3))):
    tuner._graph_batch.copy_(batch())
    graph.replay()
    out = loss.clone()
    torch.cuda.synchronize(); print('replay', i, float(out), flush=True)
