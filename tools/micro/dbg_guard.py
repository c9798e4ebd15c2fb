"""What the clamp guard sees inside a real BERT-large step: the forward's norm bounds per layer,
the largest |score| any pair of the slice can reach (Cauchy-Schwarz) and the largest that occurs."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'spt-proto_amd'))
import torch
import bench
from naive_gpt import ext, utils

bench.set_config('bert-large')
dev = torch.device('cuda:0')
LAYERS = int(os.environ.get('SPT_LAYERS', 4))
model = bench.build_model('sparse', dev, LAYERS)
tuner = utils.SparseTuner(model)
seen = []
real = ext.attention_mfma_forward


def spy(tiles, q, k, v, scale, clamp, *a, **kw):
    y, row_sum = real(tiles, q, k, v, scale, clamp, *a, **kw)
    B, S = row_sum.shape
    stats = torch.as_strided(row_sum, [B * S + 16 * B], [1])
    b = stats[B * S:].view(B, 8, 2)
    N, S_, H, E = q.shape
    qn = q.float().pow(2).sum(-1).amax().item()
    kn = k.float().pow(2).sum(-1).amax().item()
    sc = torch.einsum('nshe,nthe->nhst', q.float(), k.float()).abs().amax().item() * scale
    seen.append((b[..., 0].amax().item(), b[..., 1].amax().item(), qn, kn, sc, scale, clamp, tiles.nnz // S))
    return y, row_sum


ext.attention_mfma_forward = spy
gen = torch.Generator(device=dev).manual_seed(1)
for step in range(int(os.environ.get('SPT_STEPS', 2))):
    batch = torch.randint(3, bench.VOCAB, [4, bench.S + 2], device=dev, generator=gen)
    tuner.training_step(batch, pq_loss=True)
for i, s in enumerate(seen):
    if (i // LAYERS) % 5 != 0 and i // LAYERS != int(os.environ.get('SPT_STEPS', 2)) - 1:
        continue
    if i % LAYERS not in (0, LAYERS // 2, LAYERS - 1):
        continue
    print('step', i // LAYERS, 'layer', i % LAYERS, end=' ')
    reach = (s[0] * s[1]) ** 0.5 * s[5]
    print('bounds q2 %.1f k2 %.1f | true %.1f %.1f | reach %.2f | max |score| %.2f | scale %.4f clamp %.1f Z %d'
          % (s[0], s[1], s[2], s[3], reach, s[4], s[5], s[6], s[7]))
