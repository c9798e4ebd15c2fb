"""rocprofv3 target: the matrix-core attention kernels at the configs[2] shape, 20 rounds.
EXACT=0: no norm bounds in the forward, no exact clamp mask in the backward (ext.EXACT_CLAMP)."""
import os, sys
ROOT = os.environ.get('SPT_TREE') or os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
from naive_gpt import ext
ext.EXACT_CLAMP = os.environ.get('EXACT', '1') != '0'
N, S, H, E = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 512, 16, 64)))
torch.manual_seed(0)
q, k, v = [torch.randn([N, S, H, E], device='cuda') for _ in range(3)]
# SPT_QK_SCALE > ~1.12: the slice's largest |q| |k| scale reaches the clamp (the guard's per-tile test runs)
q, k = q * float(os.environ.get('SPT_QK_SCALE', 1)), k * float(os.environ.get('SPT_QK_SCALE', 1))
table = torch.randn([E // 8, 16, 8], device='cuda')
qc, kc = ext.pq_encode_heads(q, table), ext.pq_encode_heads(k, table)
idx = ext.lookup_forward_cuda(torch.empty([8]), qc, kc).flatten(1)
tiles = ext.attention_mfma_prepare(idx, S, lookup_pattern=True)
for _ in range(20):
    y, rs = ext.attention_mfma_forward(tiles, q, k, v, E ** -0.5, 10.0, y_transposed=True)
    ext.attention_mfma_backward(tiles, q, k, v, y, y, rs, E ** -0.5, 10.0, transposed=True)
torch.cuda.synchronize()
