"""pytest configuration: the `gpu` marker and import paths.

`-m "not gpu"` runs here (no GPU): oracle vs golden vectors, host logic, C-ABI load.
`-m gpu` runs on an MI355X: the HIP kernels through the C ABI against the oracle.
"""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, 'spt-proto_amd')
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line(
        'markers', 'gpu: needs a real MI355X (run with `-m gpu` on the GPU box)'
    )


def pytest_collection_modifyitems(config, items):
    import torch
    if torch.cuda.is_available():
        return
    skip = pytest.mark.skip(reason='no GPU in this container')
    for item in items:
        if 'gpu' in item.keywords:
            item.add_marker(skip)


@pytest.fixture
def oracle_ext(monkeypatch):
    """Install the CPU oracle behind naive_gpt.ext for host-logic tests."""
    import numpy as np
    import torch
    from naive_gpt import ext
    from oracle import ext_stub

    def sddmm(tl, tr, indptr, indices, query, key, scale=1.0, clamp=0.0):
        out = ext_stub.sddmm_forward_cuda(torch.scalar_tensor(False), torch.scalar_tensor(True),
                                          indptr, indices, query, key)
        out = out * np.float32(scale)
        return out.clamp_(-clamp, clamp) if clamp > 0 else out

    def spmm(tl, tr, indptr, indices, values, x):
        return ext_stub.spmm_forward_cuda(torch.scalar_tensor(bool(tl)), torch.scalar_tensor(False),
                                          indptr, indices, values, x)

    for name in ['cdist_forward_cuda', 'cdist_backward_cuda', 'lookup_forward_cuda',
                 'softmax_forward_cuda', 'softmax_backward_cuda']:
        monkeypatch.setattr(ext, name, getattr(ext_stub, name))
    monkeypatch.setattr(ext, 'sddmm_forward_cuda', sddmm)
    monkeypatch.setattr(ext, 'spmm_forward_cuda', spmm)
    monkeypatch.setattr(ext, 'transposed_for', lambda indptr, indices, d_head=64: None)
    monkeypatch.setattr(ext, 'spmm_transposed',
                        lambda t, indptr, indices, values, x: spmm(True, False, indptr, indices,
                                                                   values, x))
    monkeypatch.setattr(ext, 'cdist_encode',
                        lambda q, t: ext_stub.cdist_forward_cuda(q, t)[1])
    from naive_gpt.layers.sparse.attention import _SparseCore
    _SparseCore._indptr_cache.clear()
    yield
    _SparseCore._indptr_cache.clear()
