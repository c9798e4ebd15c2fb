"""Shader clock and socket power while the grouped GEMM (image path) runs back to back, against an
idle chip and a library bf16 GEMM: is the matrix pipe's rate set by the clock the part can hold?"""
import json, os, subprocess, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext


def sample():
    try:
        out = subprocess.run(['rocm-smi', '--showclocks', '--showpower', '--json'], capture_output=True,
                             text=True, timeout=20).stdout
        j = json.loads(out)
        card = j[sorted(j)[0]]
        return {k: v for k, v in card.items() if 'sclk' in k.lower() or 'power' in k.lower() or 'mclk' in k.lower()}
    except Exception as e:      # noqa
        return {'error': repr(e)}


def watch(fn, seconds=4.0):
    stop, seen = threading.Event(), []

    def loop():
        while not stop.is_set():
            seen.append(sample())
            time.sleep(0.2)
    th = threading.Thread(target=loop)
    fn(); torch.cuda.synchronize()
    th.start()
    t0 = time.perf_counter()
    n = 0
    while time.perf_counter() - t0 < seconds:
        for _ in range(50):
            fn()
        torch.cuda.synchronize()
        n += 50
    us = (time.perf_counter() - t0) / n * 1e6
    stop.set(); th.join()
    return us, seen[1:-1][:6]


dev = 'cuda'
rows, d = 24576, 1024
a = torch.randn([rows, d], device=dev)
w = torch.randn([d, d], device=dev)
ai, wi = ext.split_bf16(a), ext.split_bf16(w)
one = torch.tensor([0, rows], dtype=torch.int32, device=dev)
print('idle', sample())
us, seen = watch(lambda: ext.grouped_gemm_fused(a, w, one, 1, d, d, 0, d, 1, rows, a_image=ai, w_image=wi))
print('grouped image gemm %d x %d x %d: %.1f us, %.0f TFLOP/s executed' % (rows, d, d, us, 6.0 * rows * d * d / us / 1e6))
for s in seen:
    print('   ', s)
ab, wb = a.bfloat16(), w.bfloat16()
us, seen = watch(lambda: torch.matmul(ab, wb.T))
print('library bf16 gemm: %.1f us, %.0f TFLOP/s' % (us, 2.0 * rows * d * d / us / 1e6))
for s in seen:
    print('   ', s)
big = torch.empty([1 << 28], device=dev)
us, seen = watch(lambda: big.mul_(1.0001))
print('elementwise 1 GiB in place: %.1f us, %.0f GB/s' % (us, 2 * big.numel() * 4 / us / 1e3))
for s in seen:
    print('   ', s)
