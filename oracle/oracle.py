"""CPU ORACLE bindings (test infrastructure, NOT product code).

numpy/ctypes front-end of ``oracle/spt_oracle.c`` -- the plain-C restatement of
the reference's seven native operators (``extension/entry.cpp:43-56``).  Only
``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module; nothing under ``spt-proto_amd/`` does.

Parity status: cdist / softmax / sddmm / spmm are pinned against the reference's
own formulas and its importable Python path (``tests/golden``); the CSR structure
produced by lookup is PARITY UNPINNED beyond the reference's recall property
(``test/kernel/test_lookup.py:57-75``) -- see ``spt_oracle.c`` header.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, 'libspt_oracle.so')
_lib = None

_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int32)


def build(force: bool = False) -> str:
    """Compile spt_oracle.c with the committed Makefile (gcc)."""
    src = os.path.join(_HERE, 'spt_oracle.c')
    stale = (not os.path.exists(_LIB_PATH)) or (
        os.path.exists(src)
        and os.path.getmtime(src) > os.path.getmtime(_LIB_PATH)
    )
    if force or stale:
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libspt_oracle.so'])
    return _LIB_PATH


def lib() -> ctypes.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = ctypes.CDLL(_LIB_PATH)
        _lib.spt_oracle_lookup_forward.restype = ctypes.c_int
    return _lib


def _f32(a):
    a = np.ascontiguousarray(a, dtype=np.float32)
    return a, a.ctypes.data_as(_f32p)


def _i32(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_i32p)


def cdist_forward(query, table, want_distance: bool = True):
    """query [M,NQ,D], table [M,C,D] -> (distance [M,NQ,C] | None, indices [M,NQ])."""
    query, qp = _f32(query)
    table, tp = _f32(table)
    M, NQ, D = query.shape
    C = table.shape[1]
    assert table.shape == (M, C, D)
    indices = np.empty([M, NQ], dtype=np.int32)
    distance = np.empty([M, NQ, C], dtype=np.float32) if want_distance else None
    lib().spt_oracle_cdist_forward(
        qp, tp, distance.ctypes.data_as(_f32p) if want_distance else None,
        indices.ctypes.data_as(_i32p), M, NQ, C, D
    )
    return distance, indices


def cdist_backward(query, table, grad_output):
    query, qp = _f32(query)
    table, tp = _f32(table)
    grad_output, gp = _f32(grad_output)
    M, NQ, D = query.shape
    C = table.shape[1]
    assert grad_output.shape == (M, NQ, C)
    grad_query = np.empty_like(query)
    grad_table = np.empty_like(table)
    lib().spt_oracle_cdist_backward(
        qp, tp, gp, grad_query.ctypes.data_as(_f32p),
        grad_table.ctypes.data_as(_f32p), M, NQ, C, D
    )
    return grad_query, grad_table


def lookup_forward(query, key, sparse_coeff: int):
    """query/key [B,S,M] int32 codes -> top-k column ids [B,S,S//sparse_coeff]."""
    query, qp = _i32(query)
    key, kp = _i32(key)
    assert query.shape == key.shape and query.ndim == 3
    B, S, M = query.shape
    if sparse_coeff <= 0 or S % sparse_coeff != 0:
        raise RuntimeError('lookup: seq_length % sparsity != 0')
    out = np.empty([B, S, S // sparse_coeff], dtype=np.int32)
    rc = lib().spt_oracle_lookup_forward(
        qp, kp, out.ctypes.data_as(_i32p), B, S, M, int(sparse_coeff)
    )
    if rc != 0:
        raise RuntimeError('lookup: shape precondition failed (lookup.cu:103-106)')
    return out


def sddmm_forward(indptr, indices, query, key):
    indptr, ip = _i32(indptr)
    indices, xp = _i32(indices)
    query, qp = _f32(query)
    key, kp = _f32(key)
    B, S, E = query.shape
    nnz = indices.shape[-1]
    assert indptr.shape == (S + 1,) and indices.shape == (B, nnz)
    out = np.zeros([B, nnz], dtype=np.float32)
    lib().spt_oracle_sddmm_forward(
        ip, xp, qp, kp, out.ctypes.data_as(_f32p), B, S, E, nnz
    )
    return out


def spmm_forward(trans_lhs: bool, indptr, indices, values, x):
    indptr, ip = _i32(indptr)
    indices, xp = _i32(indices)
    values, vp = _f32(values)
    x, dp = _f32(x)
    B, S, E = x.shape
    nnz = indices.shape[-1]
    assert indptr.shape == (S + 1,) and values.shape == indices.shape == (B, nnz)
    y = np.empty([B, S, E], dtype=np.float32)
    lib().spt_oracle_spmm_forward(
        int(bool(trans_lhs)), ip, xp, vp, dp, y.ctypes.data_as(_f32p),
        B, S, E, nnz
    )
    return y


def softmax_forward(indptr, indices, values):
    indptr, ip = _i32(indptr)
    indices, xp = _i32(indices)
    values, vp = _f32(values)
    B, nnz = indices.shape
    S = indptr.shape[0] - 1
    out = np.zeros([B, nnz], dtype=np.float32)
    lib().spt_oracle_softmax_forward(
        ip, xp, vp, out.ctypes.data_as(_f32p), B, S, nnz
    )
    return out


def softmax_backward(indptr, indices, output, grad_output):
    indptr, ip = _i32(indptr)
    indices, xp = _i32(indices)
    output, op = _f32(output)
    grad_output, gp = _f32(grad_output)
    B, nnz = indices.shape
    S = indptr.shape[0] - 1
    out = np.zeros([B, nnz], dtype=np.float32)
    lib().spt_oracle_softmax_backward(
        ip, xp, op, gp, out.ctypes.data_as(_f32p), B, S, nnz
    )
    return out
