"""rocprofv3 target: spt_lora_down at the step's shapes (rows x 1024): one table / three tables with the
image by-product / plain, cold (a 512 MB buffer is swept between calls) and warm."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
dev = 'cuda'
torch.manual_seed(0)
sweep = torch.empty([128 * 1024 * 1024], device=dev)
for rows in (8192, 16384):
    x = torch.randn([rows, 1024], device=dev)
    t1 = torch.randn([1024, 16], device=dev)
    t3 = torch.randn([1024, 48], device=dev)
    for cold in (False, True):
        for _ in range(6):
            if cold:
                sweep.add_(1.0)
            ext.lora_down(x, t1)
            if cold:
                sweep.add_(1.0)
            ext.lora_down(x, t3, want_image=True, block_major=True)
torch.cuda.synchronize()
