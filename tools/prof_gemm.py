"""rocprofv3 target: the grouped GEMM at the routed-FFN and LoRA-linear shapes, image path
(pre-split operands) and register path, 5 launches each -- for the PMC traffic passes."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, dff, T = 1024, 4096, 8192
P, bs = 2 * T, dff // 4
dev = 'cuda'
torch.manual_seed(0)
a = torch.randn([T, d], device=dev)
gather = (torch.randperm(P, device=dev) % T).int()
offsets = torch.tensor([0, 4000, 8300, 12100, P], dtype=torch.int32, device=dev)
one = torch.tensor([0, T], dtype=torch.int32, device=dev)
w1 = torch.randn([dff, d], device=dev)
wq = torch.randn([d, d], device=dev)
h = torch.randn([P, bs], device=dev)
ai, hi, w1i, wqi = [ext.split_bf16(t) for t in (a, h, w1, wq)]
for _ in range(5):
    ext.grouped_gemm_fused(a, w1, offsets, 4, bs, d, bs * d, d, 1, P, gather=gather, a_image=ai, w_image=w1i)
    ext.grouped_gemm_fused(h, w1, offsets, 4, d, bs, bs * d, 1, d, P, a_image=hi, w_image=w1i)
    ext.grouped_gemm_fused(a, wq, one, 1, d, d, 0, d, 1, T, a_image=ai, w_image=wqi)
    ext.grouped_gemm_fused(a, w1, offsets, 4, bs, d, bs * d, d, 1, P, gather=gather)
    ext.grouped_gemm_fused(h, w1, offsets, 4, d, bs, bs * d, 1, d, P)
torch.cuda.synchronize()
