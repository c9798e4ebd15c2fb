"""rocprofv3 target: spt_tall_tn against the batched-product + sum it replaces."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
for rows, width, n in ((8192, 1024, 16), (8192, 1024, 48), (16384, 1024, 16)):
    wide = torch.randn([rows, width], device='cuda'); narrow = torch.randn([rows, n], device='cuda')
    for _ in range(20):
        ext.tall_tn(wide, narrow)
        parts = rows // 256
        torch.bmm(wide.reshape(parts, 256, width).transpose(1, 2), narrow.reshape(parts, 256, n)).sum(dim=0)
torch.cuda.synchronize()
