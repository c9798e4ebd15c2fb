"""spt_pq_loss_forward / _backward alone at the BERT-large step's shape (z [16, 512, 16, 64], codebook
[8, 16, 8]): HIP-event time per call.  SPT_PQ_LANES=2 runs the rounds-1-2 backward (two lanes per
sub-vector) for comparison."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext


def timeit(fn, n=50, w=10):
    for _ in range(w):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return round(e0.elapsed_time(e1) / n * 1e3, 1)


torch.manual_seed(0)
res = {'lanes': os.environ.get('SPT_PQ_LANES', '2')}
for shape in ([16, 512, 16, 64], [2, 16, 512, 16, 64], [1, 2048, 32, 128]):
    z = torch.randn(shape, device='cuda')
    table = torch.randn([shape[-1] // 8, 16, 8], device='cuda')
    g = torch.ones([], device='cuda')
    acc = torch.zeros_like(z)
    tag = 'x'.join(map(str, shape))
    res[tag] = {'forward': timeit(lambda: ext.pq_loss_forward(z, table)),
                'backward': timeit(lambda: ext.pq_loss_backward(z, table, g)),
                'backward_accumulate': timeit(lambda: ext.pq_loss_backward(z, table, g, accumulate_into=acc)),
                'MB': z.numel() * 4 / 1e6}
print(json.dumps(res))
