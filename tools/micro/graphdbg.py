import sys
sys.path.insert(0, 'spt-proto_amd')
import torch
from naive_gpt import models, utils
config = dict(d_model=1024, n_heads=16, n_layers=2, max_length=256, vocab_size=512, d_feedforward=4096, p_dropout=0.0)
def build():
    torch.manual_seed(3)
    model = models.OPTModel(**config)
    model = utils.upgrade_sparse(model, d_lora=16)
    for name, p in model.named_parameters():
        if name.endswith('lora.right.weight'):
            p.data.normal_(0, 0.02)
    return utils.SparseTuner(model.cuda())
gen = torch.Generator().manual_seed(9)
batches = [torch.randint(3, 512, [4, 258], generator=gen).cuda() for _ in range(3)]
def pdiff(a, b):
    return max(float((x - y).abs().max()) for (n, x), y in zip(a.model.named_parameters(), b.model.parameters()) if x.requires_grad)
e1, e2, g = build(), build(), build()
print('init diff', pdiff(e1, e2), pdiff(e1, g))
g.capture(batches[0].shape, pq_loss=True, warmup=3)
for _ in range(3):
    e1.training_step(torch.zeros_like(batches[0]), pq_loss=True)
    e2.training_step(torch.zeros_like(batches[0]), pq_loss=True)
print('after warmup: eager-eager', pdiff(e1, e2), 'eager-graphed', pdiff(e1, g))
for b in batches:
    l1, l2, lg = e1.training_step(b), e2.training_step(b), g.training_step(b)
    print('loss', float(l1), float(l2), float(lg), 'param diff e-e', pdiff(e1, e2), 'e-g', pdiff(e1, g))
