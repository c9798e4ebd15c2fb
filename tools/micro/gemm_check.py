"""Grouped GEMM (plain epilogue) against torch.matmul in fp64 over a few shapes / operand forms:
quick parity probe for kernel variants (SPT_GEMM_PERSISTENT=0/1, SPT_HIP_LIBRARY=...)."""
import os, sys, itertools
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
dev = 'cuda'
torch.manual_seed(0)
bad = 0
for rows, kk, n, groups, form, ext_r in itertools.product([4096, 8192 + 300], [1024, 2048], [1024, 2048], [1, 4], ['image', 'a32'], [0, 16]):
    if groups == 4:
        cuts = torch.tensor([0, rows // 5, rows // 2 + 37, rows - 200, rows], dtype=torch.int32, device=dev)
    else:
        cuts = torch.tensor([0, rows], dtype=torch.int32, device=dev)
    w = torch.randn([groups, n, kk], device=dev) / kk ** 0.5
    a = torch.randn([rows, kk], device=dev)
    bias = torch.randn([groups, n], device=dev)
    kw = {}
    if form == 'image':
        kw['a_image'] = ext.split_bf16(a)
    kw['w_image'] = ext.split_bf16(w.view(groups * n, kk))
    a2 = b2 = None
    if ext_r:
        a2 = torch.randn([rows, ext_r], device=dev)
        b2 = torch.randn([groups, n, ext_r], device=dev) * 0.1
        kw.update(a2=a2, b2=b2, b2_group_stride=n * ext_r)
    out = ext.grouped_gemm_fused(a, w.view(groups * n, kk), cuts, groups, n, kk, n * kk, kk, 1, rows,
                                 bias=bias, **kw)
    path = ext.LAST_GEMM_PATH
    want = torch.empty([rows, n], dtype=torch.float64, device=dev)
    c = cuts.tolist()
    for g in range(groups):
        sl = slice(c[g], c[g + 1])
        want[sl] = a[sl].double() @ w[g].double().T + bias[g].double()
        if ext_r:
            want[sl] += a2[sl].double() @ b2[g].double().T
    err = float((out.double() - want).abs().max() / want.abs().max())
    ok = err < 1e-4
    bad += not ok
    if not ok or os.environ.get('V'):
        print('rows %d K %d N %d groups %d %s (%s) ext %d: max rel err %.2e %s' % (rows, kk, n, groups, form, path, ext_r, err, '' if ok else '<-- BAD'))
print('cases bad:', bad)
