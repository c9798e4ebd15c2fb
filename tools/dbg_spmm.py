import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
B, S, E, Z = 32, 64, 64, 64
dev = 'cuda'
indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32, device=dev)
idx = torch.arange(Z, dtype=torch.int32, device=dev).repeat(B, S)          # every row: cols 0..63
vals1 = torch.ones([B, S * Z], device=dev)
xe = torch.arange(E, dtype=torch.float32, device=dev).repeat(B, S, 1)     # x[b,c,e] = e
y = ext.spmm_forward_cuda(False, False, indptr, idx, vals1, xe)
print('chunk placement: y[0,0,:8]', y[0, 0, :8].tolist(), 'expect', [Z * e for e in range(8)])
print('rows equal?', torch.equal(y[0, 0], y[0, 5]), 'max dev', (y - Z * xe).abs().max().item())
xc = torch.arange(S, dtype=torch.float32, device=dev).view(1, S, 1).repeat(B, 1, E)  # x[b,c,e] = c
y = ext.spmm_forward_cuda(False, False, indptr, idx, vals1, xc)
print('col sum: y[0,0,0]', y[0, 0, 0].item(), 'expect', sum(range(Z)), 'uniform', (y - sum(range(Z))).abs().max().item())
vr = torch.arange(S * Z, dtype=torch.float32, device=dev).repeat(B, 1) % 7
x1 = torch.ones([B, S, E], device=dev)
y = ext.spmm_forward_cuda(False, False, indptr, idx, vr, x1)
ref = vr.view(B, S, Z).sum(-1)
print('val sum err', (y[:, :, 0] - ref).abs().max().item(), y[0, :4, 0].tolist(), ref[0, :4].tolist())
print('--- B=4 (global gather variant)')
B = 4
idx4 = idx[:B].contiguous(); vr4 = vr[:B].contiguous(); x14 = x1[:B].contiguous()
y = ext.spmm_forward_cuda(False, False, indptr, idx4, vr4, x14)
ref = vr4.view(B, S, Z).sum(-1)
print('val sum err', (y[:, :, 0] - ref).abs().max().item(), y[0, :4, 0].tolist(), ref[0, :4].tolist())
# sddmm with q = 1, k[c,:] = c/64 -> out[r,e] = idx
kc = (torch.arange(S, dtype=torch.float32, device=dev) / 64).view(1, S, 1).repeat(32, 1, E).contiguous()
q1 = torch.ones([32, S, E], device=dev)
o = ext.sddmm_forward_cuda(False, True, indptr, idx, q1, kc)
print('sddmm row0 first/last 4', o[0, :4].tolist(), o[0, 60:64].tolist())
