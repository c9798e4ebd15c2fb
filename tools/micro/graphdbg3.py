import sys
sys.path.insert(0, 'spt-proto_amd')
import torch
from torch import optim
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
def pdiff(a, b):
    return max(float((x.detach() - y.detach()).abs().max()) for (n, x), y in zip(a.model.named_parameters(), b.model.parameters()) if x.requires_grad)
z = torch.zeros([4, 258], dtype=torch.long, device='cuda')
e1, e3, e4 = build(), build(), build()
g = e3.optimizer.param_groups[0]
e3.optimizer = optim.AdamW(e3.params, lr=torch.tensor(1e-4, device='cuda'), betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'], capturable=True)
for _ in range(3):
    e1.training_step(z); e3.training_step(z)
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(side):
    for _ in range(3):
        e4.training_step(z)
torch.cuda.current_stream().wait_stream(side)
torch.cuda.synchronize()
print('capturable optimizer:', pdiff(e1, e3), ' side stream:', pdiff(e1, e4))
rows = []
for (n, x), y in zip(e1.model.named_parameters(), e3.model.parameters()):
    if x.requires_grad:
        rows.append((float((x.detach() - y.detach()).abs().max()), n, tuple(x.shape)))
rows.sort(reverse=True)
for r in rows[:8]:
    print(r)
print('steps', e1.optimizer.state[e1.params[0]]['step'], e3.optimizer.state[e3.params[0]]['step'])
