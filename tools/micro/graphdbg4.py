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
gen = torch.Generator().manual_seed(9)
z = torch.randint(3, 512, [4, 258], generator=gen).cuda()
e1, e3 = build(), build()
g = e3.optimizer.param_groups[0]
e3.optimizer = optim.AdamW(e3.params, lr=torch.tensor(1e-4, device='cuda'), betas=g['betas'], eps=g['eps'], weight_decay=g['weight_decay'], capturable=True)
for step in range(3):
    l1, l3 = e1.training_step(z), e3.training_step(z)
    print(step, 'loss', float(l1), float(l3), 'norm', float(e1.last_grad_norm), float(e3.last_grad_norm), 'pdiff', pdiff(e1, e3))
