import os, sys
sys.path.insert(0, 'spt-proto_amd')
import torch
from naive_gpt import ext
d, dff, T = 1024, 4096, 8192
P, bs = 2 * T, dff // 4
dev='cuda'
torch.manual_seed(0)
a = torch.randn([T, d], device=dev)
gather = (torch.randperm(P, device=dev) % T).int()
offsets = torch.tensor([0, 4000, 8300, 12100, P], dtype=torch.int32, device=dev)
w1 = torch.randn([dff, d], device=dev)
coeff = (torch.rand([P], device=dev) + 0.5)
b1 = torch.randn([4, bs], device=dev)
u = torch.randn([T, 16], device=dev)
r1 = 0.05 * torch.randn([dff, 16], device=dev)
h, _ = ext.grouped_gemm_fused(a, w1, offsets, 4, bs, d, bs * d, d, 1, P, gather=gather, bias=b1, rowscale=coeff, a2=u, gather2=gather, b2=r1, b2_group_stride=bs*16, epilogue=ext.EPI_ACT, activation=ext.ACT_RELU, a_image=ext.split_bf16(a), w_image=ext.split_bf16(w1))
torch.cuda.synchronize()
q = ext.LAST_RELU_QUEUE
cnt = q[:256*64].view(torch.int32)[::16]
print('queued', int(cnt.sum()), 'max per segment', int(cnt.max()), 'of', P*bs, 'elements =', float(cnt.sum())/(P*bs))
