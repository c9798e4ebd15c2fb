import sys
sys.path.insert(0, 'spt-proto_amd')
import torch
from torch import optim
torch.manual_seed(0)
p1 = [torch.randn(1000, device='cuda', requires_grad=True) for _ in range(3)]
p2 = [p.detach().clone().requires_grad_(True) for p in p1]
o1 = optim.AdamW(p1, lr=1e-4, weight_decay=0.1)
o2 = optim.AdamW(p2, lr=torch.tensor(1e-4, device='cuda'), weight_decay=0.1, capturable=True)
for step in range(3):
    gs = [torch.randn(1000, device='cuda') for _ in range(3)]
    for a, b, g in zip(p1, p2, gs):
        a.grad = g.clone(); b.grad = g.clone()
    o1.step(); o2.step()
    print(step, max(float((a - b).abs().max()) for a, b in zip(p1, p2)))
