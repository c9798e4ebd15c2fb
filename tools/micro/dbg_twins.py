"""Two identical tuners stepped on the same batches: do their parameters stay bit-equal?
(eager/eager shows run-to-run nondeterminism of a kernel; eager/graphed what capture changes)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import models, utils
config = dict(d_model=1024, n_heads=16, n_layers=2, max_length=256, vocab_size=512,
              d_feedforward=4096, p_dropout=0.0)
n_seq = 16


def build():
    torch.manual_seed(3)
    model = models.OPTModel(**config)
    model = utils.upgrade_sparse(model, d_lora=16)
    for name, p in model.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.normal_(0, 0.02)
    return utils.SparseTuner(model.cuda())


gen = torch.Generator().manual_seed(9)
batches = [torch.randint(3, 512, [n_seq, 258], generator=gen).cuda() for _ in range(6)]
if len(sys.argv) > 2:       # the data of tests/test_models.py::test_captured_step_replays_the_eager_step
    batches = batches[:3] + batches[:2] + [batches[3]]
a, b = build(), build()
a.use_capturable_optimizer()
if 'explicit' in sys.argv:
    b.use_capturable_optimizer()
mode = sys.argv[1] if len(sys.argv) > 1 else 'eager'
if mode == 'graph':
    b.capture(batches[0].shape, pq_loss=True, warmup=3, example=batches[5])
    for _ in range(3):
        a.training_step(batches[5], pq_loss=True)
for step, x in enumerate(batches[:3]):
    la = a.training_step(x, pq_loss=True)
    lb = b.training_step(x, pq_loss=True)
    torch.cuda.synchronize()
    worst = []
    for (n, pa), pb in zip(a.model.named_parameters(), b.model.parameters()):
        if pa.requires_grad and not torch.equal(pa, pb):
            worst.append((float((pa - pb).abs().max()), n))
    worst.sort(reverse=True)
    print('step', step, 'loss', float(la), float(lb), 'params differing', len(worst), worst[:3])
# back to back, no host synchronisation in between (bench.py's timed loop)
for rep in range(3):
    for x in batches[:3] + batches[:3]:
        a.training_step(x, pq_loss=True)
        b.training_step(x, pq_loss=True)
    torch.cuda.synchronize()
    worst = []
    for (n, pa), pb in zip(a.model.named_parameters(), b.model.parameters()):
        if pa.requires_grad and not torch.equal(pa, pb):
            worst.append((float((pa - pb).abs().max()), n))
    worst.sort(reverse=True)
    print('step back-to-back', rep, 'params differing', len(worst), worst[:3])
