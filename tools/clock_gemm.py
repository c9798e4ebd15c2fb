"""In-kernel clock of the grouped GEMM (diagnostic build -DGG_STAMP, SPT_HIP_LIBRARY):
d(s_memtime) / d(s_memrealtime) * 100 MHz per workgroup, median, after 2 s of warm launches."""
import os, sys, time, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, dff, T = 1024, 4096, 8192
P, bs = 2 * T, dff // 4
dev = 'cuda'
a = torch.randn([T, d], device=dev)
gather = (torch.randperm(P, device=dev) % T).int()
offsets = torch.tensor([0, 4000, 8300, 12100, P], dtype=torch.int32, device=dev)
w1 = torch.randn([dff, d], device=dev)
h = torch.randn([P, bs], device=dev)
lib = ext.load_library()
n_blocks = 2 * (P // 128 + 4) * 8
stamps = torch.zeros([12 * n_blocks], dtype=torch.int64, device=dev)
out = torch.empty([P, 1024], device=dev)

def launch(bn):
    if bn:
        desc = ext._GroupedDesc(a=h.data_ptr(), w=w1.data_ptr(), offsets=offsets.data_ptr(), out=out.data_ptr(),
                                n_rows=P, k=bs, n=d, n_groups=4, lda=bs, w_group_stride=bs * d, w_ldn=1, w_ldk=d,
                                epilogue=0, pdot_main=stamps.data_ptr())
    else:
        desc = ext._GroupedDesc(a=a.data_ptr(), gather=gather.data_ptr(), w=w1.data_ptr(), offsets=offsets.data_ptr(),
                                out=out.data_ptr(), n_rows=P, k=d, n=bs, n_groups=4, lda=d, w_group_stride=bs * d,
                                w_ldn=d, w_ldk=1, epilogue=0, pdot_main=stamps.data_ptr())
    rc = lib.spt_grouped_gemm_fused(ctypes.byref(desc), None)
    assert rc == 0, rc

for bn in (False, True):
    t0 = time.time()
    while time.time() - t0 < 2.0:
        for _ in range(50):
            launch(bn)
        torch.cuda.synchronize()
    stamps.zero_()
    launch(bn)
    torch.cuda.synchronize()
    st = stamps.view(-1, 12).cpu()
    st = st[st[:, 2] > 0]
    start = (st[:, 1] - st[:, 1].min()).double() / 100.0        # us
    end = (st[:, 2] - st[:, 1].min()).double() / 100.0
    dur = end - start
    clk = st[:, 0].double() / (st[:, 2] - st[:, 1]).double() * 0.1
    full, halves = st[:, 3] == 0, st[:, 3] > 0
    print('bn' if bn else 'bt', 'blocks', len(st), 'clock GHz median %.3f' % clk.median(), 'makespan us %.1f' % end.max())
    for name, m in (('full', full), ('half', halves)):
        if m.any():
            print('   %s: n %d  start %.1f..%.1f  end %.1f..%.1f  dur median %.1f min %.1f max %.1f' % (
                name, int(m.sum()), start[m].min(), start[m].max(), end[m].min(), end[m].max(),
                dur[m].median(), dur[m].min(), dur[m].max()))
    # wave 0's shader-clock time per phase of the k-loop, median over full tiles (cycles per tile)
    names = ['barrier1', 'wait_loads', 'split+lds_store', 'barrier2', 'reads+mfma']
    for name, m in (('full', full), ('half', halves)):
        if m.any():
            print('   %s phases (cycles per tile, median):' % name,
                  {n: int(st[m][:, 4 + i].median()) for i, n in enumerate(names)},
                  'tile total', int(st[m][:, 0].median()))
    # start-time histogram of all blocks in 20 us bins
    import collections
    bins = collections.Counter((start / 20).long().tolist())
    print('   starts per 20 us:', [bins.get(i, 0) for i in range(int(end.max() / 20) + 1)])
# the library GEMM for comparison cannot be stamped; report its wall time only
b = torch.randn([P, d], device=dev); wd = torch.randn([bs, d], device=dev)
for _ in range(200): torch.matmul(b, wd.T)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(50): torch.matmul(b, wd.T)
torch.cuda.synchronize(); print('torch matmul us', (time.perf_counter() - t0) / 50 * 1e6)
