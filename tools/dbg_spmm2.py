import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
B, S, E, Z = 4, 64, 64, 64
dev = 'cuda'
indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32, device=dev)
idx = torch.arange(Z, dtype=torch.int32, device=dev).repeat(B, S)
x1 = torch.ones([B, S, E], device=dev)
res = []
for e in range(64):
    v = torch.zeros([B, S * Z], device=dev)
    v[:, e] = 1.0          # row 0, entry e
    v[:, 5 * Z + e] = 1.0  # row 5
    y = ext.spmm_forward_cuda(False, False, indptr, idx, v, x1)
    res.append((y[0, 0, 0].item(), y[0, 5, 0].item(), y[0, 1, 0].item()))
print('row0:', [int(r[0]) for r in res])
print('row5:', [int(r[1]) for r in res])
print('row1:', [int(r[2]) for r in res])
