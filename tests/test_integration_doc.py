"""INTEGRATION.md, route B: the ctypes `ext` stub shown to a maintainer of the reference is
executed as written (only the library path substituted) and its seven functions are compared
with the oracle on a small case.  Keeps the document and include/spt_hip.h in step."""
import os
import re
import types

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, 'spt-proto_amd', 'lib', 'libspt_hip.so')


def stub_source():
    text = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    blocks = re.findall(r'```python\n(.*?)```', text, flags=re.S)
    code = [b for b in blocks if 'binds include/spt_hip.h' in b]
    assert len(code) == 1
    return code[0].replace('/path/to/libspt_hip.so', LIB)


def test_stub_names_the_seven_operators():
    src = stub_source()
    for name in ['cdist_forward_cuda', 'cdist_backward_cuda', 'lookup_forward_cuda',
                 'spmm_forward_cuda', 'sddmm_forward_cuda', 'softmax_forward_cuda',
                 'softmax_backward_cuda']:
        assert 'def {}('.format(name) in src
    version = int(re.search(r'#define SPT_ABI_VERSION (\d+)',
                            open(os.path.join(ROOT, 'include', 'spt_hip.h')).read()).group(1))
    assert 'spt_abi_version() == {}'.format(version) in src


@pytest.mark.gpu
def test_stub_runs_and_matches_the_oracle():
    from oracle import ext_stub
    mod = types.ModuleType('doc_ext')
    exec(compile(stub_source(), 'INTEGRATION.md', 'exec'), mod.__dict__)
    gen = torch.Generator().manual_seed(0)
    B, S, E, M, C, D, coeff = 3, 128, 64, 8, 16, 8, 8
    dev = 'cuda'

    def both(name, *args):
        return getattr(mod, name)(*[a.to(dev) for a in args]), getattr(ext_stub, name)(*args)

    # cdist
    query = torch.randn([M, B * S, D], generator=gen)
    table = torch.randn([M, C, D], generator=gen)
    (gd, gi), (wd, wi) = both('cdist_forward_cuda', query, table)
    assert torch.equal(gi.cpu(), wi) and torch.allclose(gd.cpu(), wd, atol=1e-5)
    go = torch.randn([M, B * S, C], generator=gen)
    (gq, gt), (wq, wt) = both('cdist_backward_cuda', query, table, go)
    assert torch.allclose(gq.cpu(), wq, atol=1e-4) and torch.allclose(gt.cpu(), wt, rtol=1e-3, atol=1e-2)
    # lookup (bit-exact)
    qc = torch.randint(0, 8, [B, S, M], generator=gen, dtype=torch.int32)
    kc = torch.randint(0, 8, [B, S, M], generator=gen, dtype=torch.int32)
    config = torch.empty([coeff])                      # CPU tensor: only its size is read
    got = mod.lookup_forward_cuda(config, qc.to(dev), kc.to(dev))
    want = ext_stub.lookup_forward_cuda(config, qc, kc)
    assert torch.equal(got.cpu(), want)
    indices = want.flatten(1)
    Z = S // coeff
    indptr = torch.arange(0, S * Z + 1, Z, dtype=torch.int32)
    f, t = torch.scalar_tensor(False), torch.scalar_tensor(True)
    q = torch.randn([B, S, E], generator=gen)
    k = torch.randn([B, S, E], generator=gen)
    got = mod.sddmm_forward_cuda(f, t, indptr.to(dev), indices.to(dev), q.to(dev), k.to(dev))
    scores = ext_stub.sddmm_forward_cuda(f, t, indptr, indices, q, k)
    assert torch.allclose(got.cpu(), scores, rtol=1e-4, atol=1e-4)
    scores = (scores * 0.125).clamp(-10, 10)
    got = mod.softmax_forward_cuda(indptr.to(dev), indices.to(dev), scores.to(dev))
    attn = ext_stub.softmax_forward_cuda(indptr, indices, scores)
    assert torch.allclose(got.cpu(), attn, rtol=1e-4, atol=1e-6)
    ga = torch.randn(attn.shape, generator=gen)
    got = mod.softmax_backward_cuda(indptr.to(dev), indices.to(dev), attn.to(dev), ga.to(dev))
    want = ext_stub.softmax_backward_cuda(indptr, indices, attn, ga)
    assert torch.allclose(got.cpu(), want, rtol=1e-3, atol=1e-6)
    for trans in (f, t):
        got = mod.spmm_forward_cuda(trans, f, indptr.to(dev), indices.to(dev), attn.to(dev), k.to(dev))
        want = ext_stub.spmm_forward_cuda(trans, f, indptr, indices, attn, k)
        assert torch.allclose(got.cpu(), want, rtol=1e-3, atol=1e-4)
