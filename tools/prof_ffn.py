"""rocprofv3 target: the fused LoRA routed FFN alone (fwd + bwd), BERT-large dims."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from torch import nn
from naive_gpt import layers
torch.manual_seed(0)
d, dff, N, S, r = 1024, 4096, 16, 512, 16
routed = layers.LoRARoutedFFN(d_lora=r, block_size=dff // 4, d_model=d, d_feedforward=dff,
                              activation=nn.ReLU()).cuda()
for n_, p in routed.named_parameters():
    if n_.endswith('lora.right.weight'):
        p.data.normal_(0, 0.02)
x = torch.randn([N, S, d], device='cuda', requires_grad=True)
for _ in range(13):
    x.grad = None
    routed.zero_grad(set_to_none=True)
    routed(x).sum().backward()
torch.cuda.synchronize()
