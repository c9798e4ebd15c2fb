"""Time kernels.sddmm's C entry at a lookup pattern (default: the configs[2] attention shape).
   python tools/micro/time_sddmm.py [N S H E]   -> one line: shape, us, GB/s of the algorithmic bytes
SPT_SDDMM_GATHER=1 forces the gather form (sddmm.hip); SPT_HIP_LIBRARY picks a variant build."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch  # noqa: E402
from naive_gpt import ext  # noqa: E402

N, S, H, E = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 512, 16, 64)))
B, M, Z = N * H, E // 8, int(os.environ.get('Z', S // 8))
torch.manual_seed(0)
dev = 'cuda'
q, k = [torch.randn([B, S, E], device=dev) for _ in range(2)]
table = torch.randn([M, 16, 8], device=dev)


def codes(z):
    zf = z.reshape(B * S, M, 8).transpose(0, 1).contiguous()
    return ext.cdist_encode(zf, table).t().contiguous().view(B, S, M)


idx = ext.lookup_forward_cuda(torch.empty([S // Z]), codes(q), codes(k)).flatten(1)
indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32, device=dev)
vals = torch.rand([B, S * Z], device=dev)


def timeit(fn, reps=30):
    for _ in range(5):
        fn()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3


res = {'shape': [N, S, H, E, Z], 'lib': os.environ.get('SPT_HIP_LIBRARY', 'in-tree')[-30:],
       'gather': bool(os.environ.get('SPT_SDDMM_GATHER'))}
byts = 2 * B * S * E * 4 + 2 * B * S * Z * 4
for name, fn in (('sddmm', lambda: ext.sddmm_forward_cuda(False, True, indptr, idx, q, k)),
                 ('spmm', lambda: ext.spmm_forward_cuda(False, False, indptr, idx, vals, k))):
    if name in os.environ.get('OPS', 'sddmm,spmm'):
        us = timeit(fn)
        res[name + '_us'] = round(us, 1)
        res[name + '_frac_of_8TBs'] = round(byts / us / 1e6 / 8.0, 3)
print(res)
if os.environ.get('ST_STAMPS') == '1':        # a -DST_STAMP build (tools/variant.sh): workgroup 0's timeline
    o = ext.sddmm_forward_cuda(False, True, indptr, idx, q, k)
    torch.cuda.synchronize()
    st = o.flatten()[:1024].view(torch.int32).cpu().view(8, 128).tolist()
    t00 = min(r[0] for r in st)
    for w, r in enumerate(st):
        marks = [(x - t00) & 0xffffffff for x in r[:2 + 4 * (S // 32)]]
        its = marks[2:]
        print('wave', w, 'start', marks[0], 'loop', marks[1], 'iterations [top, half, both, stored]:',
              [tuple(its[4 * i:4 * i + 4]) for i in range(0, S // 32, 3)], 'last', its[-1])
