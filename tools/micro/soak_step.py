"""N steps of the configs[2] tuner on a FIXED batch: the loss must stay finite and fall (a soak run for
rare paths: ReLU fix-up queue, saturated routers, PQ pair passes, graph replays)."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext, models, utils
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
torch.manual_seed(0)
model = models.OPTModel(d_model=1024, n_heads=16, n_layers=int(os.environ.get('LAYERS', 6)), max_length=512,
                        vocab_size=30522, d_feedforward=4096, p_dropout=0.0)
tuner = utils.SparseTuner(utils.upgrade_sparse(model, d_lora=16).cuda(), lr=1e-3)
batch = torch.randint(3, 30522, [16, 514]).cuda()
if os.environ.get('GRAPH') == '1':
    tuner.capture(batch.shape, pq_loss=True, warmup=3, example=batch)
losses = []
for i in range(steps):
    losses.append(float(tuner.training_step(batch, pq_loss=True)))
    assert math.isfinite(losses[-1]), (i, losses[-5:])
print('first', [round(v, 4) for v in losses[:3]], 'last', [round(v, 4) for v in losses[-3:]])
assert losses[-1] < losses[0], 'the loss did not fall on a fixed batch'
print('paths', {k: v for k, v in ext.paths_taken().items() if 'library' in k[1] or 'torch' in k[1]})
