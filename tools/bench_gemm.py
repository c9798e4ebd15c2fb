"""Grouped GEMM micro-benchmark at the routed-FFN shapes (P = 16384 rows, 4 buckets, K = N = 1024)
and the LoRA-linear shape (8192 rows, one bucket): register path (fp32 operands split in the
k-loop) against the image path (pre-split operands, LDS-DMA), the split passes timed apart."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
d, dff, T = 1024, 4096, 8192 * int(os.environ.get('MULT', 1))     # MULT=16: ~1 ms launches (clock readings)
P, bs = 2 * T, dff // 4
dev = 'cuda'
torch.manual_seed(0)


def timeit(fn, n=30, w=10):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


a = torch.randn([T, d], device=dev)
gather = (torch.randperm(P, device=dev) % T).int()
even = os.environ.get('EVEN', '0') == '1'
cuts = [0, P // 4, P // 2, 3 * P // 4, P] if even else [0, P * 4000 // 16384, P * 8300 // 16384, P * 12100 // 16384, P]
offsets = torch.tensor(cuts, dtype=torch.int32, device=dev)
one = torch.tensor([0, T], dtype=torch.int32, device=dev)
w1 = torch.randn([dff, d], device=dev)
wq = torch.randn([d, d], device=dev)
h = torch.randn([P, bs], device=dev)
res = {}


def both(tag, flops, call):
    us = timeit(lambda: call(None, None))
    res[tag + '.regs'] = {'us': round(us, 1), 'TFLOPs': round(flops / us / 1e6, 1)}
    imgs = call.images()
    us = timeit(lambda: call(*imgs))
    assert ext.LAST_GEMM_PATH == 'image'
    res[tag + '.image'] = {'us': round(us, 1), 'TFLOPs': round(flops / us / 1e6, 1)}
    us = timeit(lambda: call(None, imgs[1]))
    assert ext.LAST_GEMM_PATH == 'a32'
    res[tag + '.a32'] = {'us': round(us, 1), 'TFLOPs': round(flops / us / 1e6, 1)}


class FfnUp:                      # x[gather] . W1_g^T : bt, gathered rows
    def images(self):
        return ext.split_bf16(a), ext.split_bf16(w1)

    def __call__(self, ai, wi):
        return ext.grouped_gemm_fused(a, w1, offsets, 4, bs, d, bs * d, d, 1, P, gather=gather,
                                      a_image=ai, w_image=wi)


class FfnBack:                    # dH . W1_g : bn
    def images(self):
        return ext.split_bf16(h), ext.split_bf16(w1)

    def __call__(self, ai, wi):
        return ext.grouped_gemm_fused(h, w1, offsets, 4, d, bs, bs * d, 1, d, P, a_image=ai, w_image=wi)


class Linear:                     # x . Wq^T : one bucket, 8192 rows
    def images(self):
        return ext.split_bf16(a), ext.split_bf16(wq)

    def __call__(self, ai, wi):
        return ext.grouped_gemm_fused(a, wq, one, 1, d, d, 0, d, 1, T, a_image=ai, w_image=wi)


class LinearBack:                 # dY . Wq : one bucket, bn
    def images(self):
        return ext.split_bf16(a), ext.split_bf16(wq)

    def __call__(self, ai, wi):
        return ext.grouped_gemm_fused(a, wq, one, 1, d, d, 0, 1, d, T, a_image=ai, w_image=wi)


both('ffn_bt', 2.0 * P * bs * d, FfnUp())
both('ffn_bn', 2.0 * P * bs * d, FfnBack())
both('linear_bt', 2.0 * T * d * d, Linear())
both('linear_bn', 2.0 * T * d * d, LinearBack())
for tag, t in (('split_8192x1024', a), ('split_16384x1024', h), ('split_w_4096x1024', w1),
               ('split_w_1024x1024', wq)):
    us = timeit(lambda: ext.split_bf16(t))
    res[tag] = {'us': round(us, 1), 'GBps': round(2 * t.numel() * 4 / us / 1e3, 1)}
b = torch.randn([P, d], device=dev)
wd = torch.randn([bs, d], device=dev)
us = timeit(lambda: torch.matmul(b, wd.T))
res['torch_fp32'] = {'us': round(us, 1), 'TFLOPs': round(2.0 * P * bs * d / us / 1e6, 1)}
print(json.dumps(res))

# ---- the FFN's first GEMM as the layer calls it: what each epilogue feature costs (image path)
if os.environ.get('EPILOGUES', '1') == '1':
    res2 = {}
    coeff = (torch.rand([P], device=dev) + 0.5)
    b1 = torch.randn([4, bs], device=dev)
    u = torch.randn([T, 16], device=dev)
    r1 = 0.05 * torch.randn([dff, 16], device=dev)
    ai, wi = ext.split_bf16(a), ext.split_bf16(w1)
    an, wn = ext.row_norms(a), ext.row_norms(w1)

    def up(**kw):
        return ext.grouped_gemm_fused(a, w1, offsets, 4, bs, d, bs * d, d, 1, P, gather=gather,
                                      a_image=ai, w_image=wi, **kw)
    full = dict(bias=b1, rowscale=coeff, a2=u, gather2=gather, b2=r1, b2_group_stride=bs * 16)
    cases = {
        'plain': {},
        'bias+rowscale': dict(bias=b1, rowscale=coeff),
        'ext': dict(a2=u, gather2=gather, b2=r1, b2_group_stride=bs * 16),
        'bias+rowscale+ext': full,
        'act_gelu': dict(full, epilogue=ext.EPI_ACT, activation=ext.ACT_GELU),
        'act_gelu_keep': dict(full, epilogue=ext.EPI_ACT, activation=ext.ACT_GELU, keep_preact=True),
        'act_relu': dict(full, epilogue=ext.EPI_ACT, activation=ext.ACT_RELU, a_norm=an, w_norm=wn),
        'act_relu_noqueue': dict(full, epilogue=ext.EPI_ACT, activation=ext.ACT_RELU, a_norm=an,
                                 w_norm=wn, relu_queue_entries=0),
    }
    for tag, kw in cases.items():
        res2[tag] = round(timeit(lambda: up(**kw)), 1)
    print(json.dumps({'ffn_up_image_us': res2}))
