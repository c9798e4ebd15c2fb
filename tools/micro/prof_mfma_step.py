"""The attention kernels on the operands of a REAL step (layer 0 of the BERT-large sparse model at
the bench's batch): the pattern of a real PQ lookup, not the uniform one of prof_mfma.py.  Meant
for rocprofv3 --kernel-trace --stats (tools/gpu.sh stats TAG tools/micro/prof_mfma_step.py)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import ext, utils

bench.set_config('bert-large')
dev = torch.device('cuda:0')
model = bench.build_model('sparse', dev, 2)
tuner = utils.SparseTuner(model)
fwd_args, bwd_args = [], []
real_f, real_b = ext.attention_mfma_forward, ext.attention_mfma_backward


def spy_f(*a, **kw):
    if not fwd_args:
        fwd_args.append((a, kw))
    return real_f(*a, **kw)


def spy_b(*a, **kw):
    if not bwd_args:
        bwd_args.append(([t.clone() if isinstance(t, torch.Tensor) else t for t in a], kw))
    return real_b(*a, **kw)


ext.attention_mfma_forward, ext.attention_mfma_backward = spy_f, spy_b
gen = torch.Generator(device=dev).manual_seed(1)
batch = torch.randint(3, bench.VOCAB, [16, bench.S + 2], device=dev, generator=gen)
tuner.training_step(batch, pq_loss=True)
torch.cuda.synchronize()
(fa, fkw), (ba, bkw) = fwd_args[0], bwd_args[0]
tiles, q, k, v, y, gy, row_sum = ba[:7]
print('q', tuple(q.shape), 'Z', tiles.nnz // q.size(1), 'kwargs', fkw, bkw, ba[7:])
for _ in range(int(os.environ.get('SPT_REPS', 20))):
    y2, rs2 = real_f(tiles, q, k, v, *fa[4:], **fkw)
    real_b(tiles, q, k, v, y2, gy, rs2, *ba[7:], **bkw)
torch.cuda.synchronize()
