"""CPU ORACLE presented with the reference's ``naive_gpt.ext`` surface.

Test infrastructure only.  The seven callables of ``extension/entry.cpp:43-56``
taking / returning CPU ``torch.Tensor``s, computed by ``oracle/spt_oracle.c``.
Two uses, both outside the product path:

* ``tests/golden/make_golden.py`` installs it as ``sys.modules['naive_gpt.ext']``
  under the *imported reference* layers, so the reference's own orchestration code
  (``naive_gpt/layers/sparse/attention.py:84-142``) produces full-layer goldens;
* CPU tests install it under this repo's ``naive_gpt`` mirror to check host logic
  (autograd wiring, layers) without a GPU.
"""
import torch

from . import oracle as _o


def _np(t: torch.Tensor):
    return t.detach().cpu().contiguous().numpy()


def _check(x: torch.Tensor, dim: int, name: str):
    # CHECK_DIM of extension/common.h:13-18 minus the device test
    if x.dim() != dim:
        raise RuntimeError('{} must be of dim {}'.format(name, dim))
    if not x.is_contiguous():
        raise RuntimeError(
            '{} custom kernel requires contiguous tensor'.format(name)
        )


def cdist_forward_cuda(query, table):
    _check(query, 3, 'query')
    _check(table, 3, 'table')
    distance, indices = _o.cdist_forward(_np(query), _np(table))
    return [torch.from_numpy(distance), torch.from_numpy(indices)]


def cdist_backward_cuda(query, table, grad_output):
    _check(grad_output, 3, 'grad_output')
    gq, gt = _o.cdist_backward(_np(query), _np(table), _np(grad_output))
    return [torch.from_numpy(gq), torch.from_numpy(gt)]


def lookup_forward_cuda(config, query, key):
    _check(query, 3, 'query')
    _check(key, 3, 'key')
    out = _o.lookup_forward(_np(query), _np(key), int(config.size(0)))
    return torch.from_numpy(out)


def sddmm_forward_cuda(trans_lhs, trans_rhs, indptr, indices, query, key):
    assert not bool(trans_lhs.item()) and bool(trans_rhs.item())
    out = _o.sddmm_forward(_np(indptr), _np(indices), _np(query), _np(key))
    return torch.from_numpy(out)


def spmm_forward_cuda(trans_lhs, trans_rhs, indptr, indices, values, x):
    assert not bool(trans_rhs.item())
    out = _o.spmm_forward(
        bool(trans_lhs.item()), _np(indptr), _np(indices), _np(values), _np(x)
    )
    return torch.from_numpy(out)


def softmax_forward_cuda(indptr, indices, values):
    out = _o.softmax_forward(_np(indptr), _np(indices), _np(values))
    return torch.from_numpy(out)


def softmax_backward_cuda(indptr, indices, output, grad_output):
    out = _o.softmax_backward(
        _np(indptr), _np(indices), _np(output), _np(grad_output)
    )
    return torch.from_numpy(out)
