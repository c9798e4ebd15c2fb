import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
N, S, H, E = 16, 512, 16, 64
B, M, Z = N * H, E // 8, S // 8
torch.manual_seed(0)
dev = 'cuda'
q = torch.randn([B, S, E], device=dev); k = torch.randn([B, S, E], device=dev)
table = torch.randn([M, 16, 8], device=dev)
def codes(z):
    zf = z.reshape(B * S, M, 8).transpose(0, 1).contiguous()
    return ext.cdist_encode(zf, table).t().contiguous().view(B, S, M)
idx = ext.lookup_forward_cuda(torch.empty([8]), codes(q), codes(k)).flatten(1)
indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32, device=dev)
vals = torch.rand([B, S * Z], device=dev)
empty_ptr = torch.zeros([S + 1], dtype=torch.int32, device=dev)
# rows with only the first 16 entries
short_ptr = indptr.clone()
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
print('sddmm full      %.1f us' % t(lambda: ext.sddmm_forward_cuda(False, True, indptr, idx, q, k)))
print('sddmm empty csr %.1f us' % t(lambda: ext.sddmm_forward_cuda(False, True, empty_ptr, idx, q, k)))
print('spmm  full      %.1f us' % t(lambda: ext.spmm_forward_cuda(False, False, indptr, idx, vals, k)))
print('spmm  empty csr %.1f us' % t(lambda: ext.spmm_forward_cuda(False, False, empty_ptr, idx, vals, k)))
tr = ext.csr_transpose(indptr, idx)
print('spmmT full      %.1f us' % t(lambda: ext.spmm_transposed(tr, indptr, idx, vals, k)))
idx_sorted = torch.sort(idx.view(B, S, Z), dim=-1).values.reshape(B, -1).contiguous()
print('sddmm sorted idx %.1f us' % t(lambda: ext.sddmm_forward_cuda(False, True, indptr, idx_sorted, q, k)))
x = torch.empty(256 * 1024 * 1024 // 4, device=dev); y = torch.empty_like(x)
print('copy 256MB       %.1f us -> %.0f GB/s' % ((lambda u: (u, 2 * 256 * 1.048576e3 / u * 1e3))(t(lambda: y.copy_(x), 10))))
