"""Routed FFN at BERT-large dimensions: grouped-MFMA path vs per-block torch path vs dense."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from torch import nn
from naive_gpt import layers, ext
from naive_gpt.layers.sparse import grouped

torch.manual_seed(0)
d, dff, N, S, r = 1024, 4096, 16, 512, 16
dev = 'cuda'
routed = layers.LoRARoutedFFN(d_lora=r, block_size=dff // 4, d_model=d, d_feedforward=dff,
                              activation=nn.ReLU()).to(dev)
for n_, p in routed.named_parameters():
    if n_.endswith('lora.right.weight'):
        p.data.normal_(0, 0.02)
dense = layers.Feedforward(d_model=d, d_feedforward=dff, p_dropout=0.0, activation=nn.ReLU()).to(dev)
x = torch.randn([N, S, d], device=dev, requires_grad=True)


def timeit(fn, n=10, w=3):
    for _ in range(w):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3


def step(m):
    def f():
        x.grad = None
        m.zero_grad(set_to_none=True)
        m(x).sum().backward()
    return f


res = {}
res['routed_grouped_ms'] = timeit(step(routed))
orig = grouped.usable
grouped.usable = lambda *a, **k: False
res['routed_torch_loop_ms'] = timeit(step(routed))
grouped.usable = orig
res['dense_ms'] = timeit(step(dense))
# raw grouped GEMM rate: P = T*2 rows, 4 buckets
T = N * S
P = 2 * T
a = torch.randn([T, d], device=dev)
gather = (torch.arange(P, device=dev) % T).int()
offsets = torch.tensor([0, P // 4, P // 2, 3 * P // 4, P], dtype=torch.int32, device=dev)
w1 = torch.randn([dff, d], device=dev)
f = lambda: ext.grouped_gemm(a, w1, offsets, 4, dff // 4, d, (dff // 4) * d, d, 1, gather=gather)
ms = timeit(f, 20, 5)
flops = 2.0 * P * (dff // 4) * d
res['grouped_gemm_bt_ms'] = ms
res['grouped_gemm_bt_TFLOPs'] = flops / ms / 1e9
h = torch.randn([P, dff // 4], device=dev)
f2 = lambda: ext.grouped_gemm(h, w1, offsets, 4, d, dff // 4, (dff // 4) * d, 1, d)
ms2 = timeit(f2, 20, 5)
res['grouped_gemm_bn_ms'] = ms2
res['grouped_gemm_bn_TFLOPs'] = flops / ms2 / 1e9
b = torch.randn([P, d], device=dev)
wd = torch.randn([dff // 4, d], device=dev)
ms3 = timeit(lambda: torch.matmul(b, wd.T), 20, 5)
res['torch_matmul_same_flops_ms'] = ms3
res['torch_matmul_TFLOPs'] = flops / ms3 / 1e9
res['tokens_per_s_routed_grouped'] = T / res['routed_grouped_ms'] * 1e3
# the gated (LLaMA) variant at the same dimensions
routed_l = layers.LoRARoutedLLaMaFFN(d_lora=r, block_size=dff // 4, d_model=d, d_feedforward=dff,
                                     activation=nn.SiLU()).to(dev)
for n_, p in routed_l.named_parameters():
    if n_.endswith('lora.right.weight'):
        p.data.normal_(0, 0.02)
dense_l = layers.LLaMaFeedforward(d_model=d, d_feedforward=dff, activation=nn.SiLU()).to(dev)
for p in dense_l.parameters():
    p.requires_grad = False
res['llama_routed_grouped_ms'] = timeit(step(routed_l))
grouped.usable = lambda *a, **k: False
res['llama_routed_torch_loop_ms'] = timeit(step(routed_l))
grouped.usable = orig
res['llama_dense_frozen_ms'] = timeit(step(dense_l))
print(json.dumps(res))
