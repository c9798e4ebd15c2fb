"""Scores placed on the +-10 clamp (tests/test_gpu_mfma_attention.py::_scores_on_the_clamp): which
rows of grad_q / grad_k differ from the oracle, with the exact recomputation on and off."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'spt-proto_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
import numpy as np
import torch
from naive_gpt import ext
from oracle import ext_stub
from test_gpu_mfma_attention import _scores_on_the_clamp, oracle_chain, CLAMP

N, H, S, Z, E = (int(x) for x in (sys.argv[1:6] if len(sys.argv) > 5 else (1, 8, 80, 8, 64)))
B = N * H
q, k, v, indices, scale = _scores_on_the_clamp(N, H, S, Z, E, 11 * S + E, 2.0 ** -15)
gy = torch.randn([B, S, E], generator=torch.Generator().manual_seed(5))
scores, attn, _ = oracle_chain(indices, q, k, v, scale)
flat = lambda t: t.transpose(1, 2).contiguous().view(B, S, E)
heads = lambda t: t.view(N, H, S, E).transpose(1, 2)
indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32)
f, t = torch.scalar_tensor(False), torch.scalar_tensor(True)
raw_scores = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, flat(q), flat(k)) * np.float32(scale)
dp = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, gy, flat(v))
ds = ext_stub.softmax_backward_cuda(indptr, indices, attn, dp)
raw = torch.where(raw_scores.abs() < CLAMP, ds * np.float32(scale), torch.zeros_like(ds))
want_q = heads(ext_stub.spmm_forward_cuda(f, f, indptr, indices, raw, flat(k)))
want_k = heads(ext_stub.spmm_forward_cuda(t, f, indptr, indices, raw, flat(q)))
tiles = ext.attention_mfma_prepare(indices.cuda(), S)
y, row_sum = ext.attention_mfma_forward(tiles, q.cuda(), k.cuda(), v.cuda(), scale, CLAMP)
bounds = row_sum.flatten().new_empty(0)
nb = ext.load_library().spt_attention_mfma_bounds_floats(B)
st = torch.empty(0)
full = torch.as_strided(row_sum, [B * S + nb], [1]).cpu()
bd = full[B * S:].view(B, 8, 2)
qn = flat(q).square().sum(-1).max(-1).values
kn = flat(k).square().sum(-1).max(-1).values
print('bounds q: kernel', bd[:3, 0, 0].tolist(), 'true', qn[:3].tolist())
print('bounds k: kernel', bd[:3, :2, 1].tolist(), 'true', kn[:3].tolist())
for exact in (True, False):
    ext.EXACT_CLAMP = exact
    gq, gk, _ = ext.attention_mfma_backward(tiles, q.cuda(), k.cuda(), v.cuda(), y, gy.cuda(), row_sum, scale, CLAMP)
    for name, got, want in (('grad_q', gq, want_q), ('grad_k', gk, want_k)):
        err = (got.cpu() - want).abs().amax(-1)          # [N, S, H]
        bar = 2e-4 * want.abs().max()
        bad = (err > bar).nonzero()
        print('exact' if exact else 'split', name, 'max rel', (err.max() / want.abs().max()).item(),
              'rows off', len(bad), 'of', err.numel(), 'first', bad[:6].tolist())

# flip signature: is an off row's error (one cell's dS) * k[col], with either sign?
ext.EXACT_CLAMP = True
gq, gk, _ = ext.attention_mfma_backward(tiles, q.cuda(), k.cuda(), v.cuda(), y, gy.cuda(), row_sum, scale, CLAMP)
err = (gq.cpu() - want_q)                                   # [N, S, H, E]
rowerr = err.abs().amax(-1)
bad = (rowerr > 2e-4 * want_q.abs().max()).nonzero()
dsu = (ds * np.float32(scale)).view(B, S, Z)
idx3 = indices.view(B, S, Z).long()
kf = flat(k)
for n, srow, h in bad[:8].tolist():
    b = n * H + h
    e = err[n, srow, h].double()
    cands = dsu[b, srow].double().unsqueeze(-1) * kf[b][idx3[b, srow]].double()      # [Z, E]
    live = idx3[b, srow] <= srow
    res = torch.minimum((cands - e).norm(dim=-1), (cands + e).norm(dim=-1)) / e.norm()
    res[~live] = 9
    p = int(res.argmin())
    rs = raw_scores.view(B, S, Z)[b, srow]
    print('row', (n, srow, h), 'err', float(e.norm()), 'best single-cell match residual', float(res[p]),
          'cell col', int(idx3[b, srow, p]), 'raw score', float(rs[p]), 'dist to clamp', float(abs(abs(rs[p]) - CLAMP)),
          'n near', int(((rs.abs() - CLAMP).abs() < 1e-3).sum()), 'mult', int((idx3[b, srow] == idx3[b, srow, p]).sum()))
