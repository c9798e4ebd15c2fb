"""``naive_gpt.ext`` -- the native-operator module of SPT-proto, MI355X edition.

Same seven callables, names and argument order as the reference's pybind11 module
(``extension/entry.cpp:43-56``); the ``_cuda`` suffix is API, not a dependency.
Each call validates its tensors the way the reference's ``CHECK_DIM`` /
``CHECK_TYPE`` macros do (``extension/common.h:13-21``), allocates the outputs
through torch's caching allocator, and enqueues hand-written gfx950 kernels from
``libspt_hip.so`` (C ABI: ``include/spt_hip.h``) on torch's *current* HIP stream.

There is no CPU path and no fallback: without the built library, or with CPU
tensors, every call raises ``RuntimeError``.
"""
import collections
import ctypes
import logging
import os
import weakref

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# SPT_HIP_LIBRARY: another build of the same library (kernel A/B experiments, tools/)
LIB_PATH = os.environ.get('SPT_HIP_LIBRARY') or os.path.join(os.path.dirname(_HERE), 'lib',
                                                            'libspt_hip.so')

_c_int = ctypes.c_int
_c_f32 = ctypes.c_float
_c_ptr = ctypes.c_void_p

# name -> argtypes, exactly the prototypes of include/spt_hip.h
_PROTOTYPES = {
    'spt_abi_version': ([], _c_int),
    'spt_strerror': ([_c_int], ctypes.c_char_p),
    'spt_cdist_forward': ([_c_ptr] * 4 + [_c_int] * 4 + [_c_ptr], _c_int),
    'spt_cdist_backward_workspace_bytes': ([_c_int] * 4, ctypes.c_int64),
    'spt_cdist_backward': ([_c_ptr] * 6 + [_c_int] * 4 + [_c_ptr], _c_int),
    'spt_lookup_forward': ([_c_ptr] * 3 + [_c_int] * 4 + [_c_ptr], _c_int),
    'spt_pq_encode_heads': ([_c_ptr] * 3 + [_c_int] * 6 + [_c_ptr], _c_int),
    'spt_pq_encode_heads_bf16': ([_c_ptr] * 3 + [_c_int] * 6 + [_c_ptr], _c_int),
    'spt_pq_loss_workspace_bytes': ([ctypes.c_int64] + [_c_int] * 3, ctypes.c_int64),
    'spt_pq_loss_forward': ([_c_ptr] * 4 + [ctypes.c_int64] + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_pq_loss_forward_codes': ([_c_ptr] * 5 + [_c_int] * 6 + [_c_ptr], _c_int),
    'spt_pq_loss_backward': ([_c_ptr] * 6 + [ctypes.c_int64] + [_c_int] * 4 + [_c_ptr], _c_int),
    'spt_pq_loss_forward_codes_parts': ([_c_ptr] * 5 + [_c_int] * 7 + [_c_ptr], _c_int),
    'spt_pq_loss_backward_parts': ([_c_ptr] * 6 + [ctypes.c_int64] + [_c_int] * 5 + [_c_ptr], _c_int),
    'spt_sddmm_forward': (
        [_c_ptr] * 5 + [_c_int] * 4 + [_c_f32, _c_f32, _c_int, _c_int, _c_ptr], _c_int
    ),
    'spt_sddmm_form': ([_c_int] * 4, _c_int),
    'spt_spmm_form': ([_c_int] * 5, _c_int),
    'spt_spmm_workspace_bytes': ([_c_int] * 5, ctypes.c_int64),
    'spt_spmm_forward': ([_c_int] + [_c_ptr] * 6 + [_c_int] * 6 + [_c_ptr], _c_int),
    'spt_csr_transpose_workspace_bytes': ([_c_int] * 4, ctypes.c_int64),
    'spt_csr_transpose': ([_c_ptr] * 3 + [_c_int] * 4 + [_c_ptr], _c_int),
    'spt_spmm_transposed_workspace_bytes': ([_c_int] * 4, ctypes.c_int64),
    'spt_spmm_transposed': ([_c_ptr] * 6 + [_c_int] * 6 + [_c_ptr], _c_int),
    'spt_grouped_gemm': ([_c_ptr] * 7 + [_c_int] * 5 + [ctypes.c_longlong, _c_int, _c_int, _c_ptr],
                         _c_int),
    'spt_sparse_attention_forward': ([_c_ptr] * 7 + [_c_int] * 4 + [_c_f32, _c_f32, _c_int, _c_int,
                                                                 _c_int, _c_ptr], _c_int),
    'spt_sparse_attention_backward_rows': ([_c_ptr] * 9 + [_c_int] * 4 + [_c_f32, _c_f32] +
                                           [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_attention_mfma_supported': ([_c_int] * 3, _c_int),
    'spt_attention_mfma_tiles_bytes': ([_c_int] * 4, ctypes.c_int64),
    'spt_attention_mfma_prepare': ([_c_ptr] * 2 + [_c_int] * 4 + [_c_ptr], _c_int),
    'spt_attention_mfma_bounds_floats': ([_c_int], _c_int),
    'spt_attention_mfma_forward': ([_c_ptr, _c_int] + [_c_ptr] * 6 + [_c_int] * 4 +
                                   [_c_f32, _c_f32, _c_int, _c_int, _c_ptr], _c_int),
    'spt_attention_mfma_backward': ([_c_ptr, _c_int] + [_c_ptr] * 11 + [_c_int] * 4 +
                                    [_c_f32, _c_f32, _c_int, _c_int, _c_ptr], _c_int),
    'spt_attention_mfma_forward_bf16': ([_c_ptr, _c_int] + [_c_ptr] * 6 + [_c_int] * 4 +
                                        [_c_f32, _c_f32, _c_int, _c_int, _c_ptr], _c_int),
    'spt_attention_mfma_backward_bf16': ([_c_ptr, _c_int] + [_c_ptr] * 11 + [_c_int] * 4 +
                                         [_c_f32, _c_f32, _c_int, _c_int, _c_ptr], _c_int),
    'spt_grouped_gemm_fused': ([_c_ptr, _c_ptr], _c_int),
    'spt_grouped_gemm_pdot_width': ([_c_int], _c_int),
    'spt_grouped_gemm_image_path': ([_c_ptr], _c_int),
    'spt_split_bf16_bytes': ([ctypes.c_longlong, _c_int], ctypes.c_size_t),
    'spt_split_bf16': ([_c_ptr, _c_ptr, ctypes.c_longlong, _c_int, ctypes.c_longlong, _c_ptr], _c_int),
    'spt_rotary': ([_c_ptr, _c_int, _c_int, _c_ptr, ctypes.c_longlong, _c_ptr, _c_ptr] + [_c_int] * 5 + [_c_ptr],
                   _c_int),
    'spt_swiglu_forward': ([_c_ptr] * 3 + [ctypes.c_longlong, _c_ptr], _c_int),
    'spt_swiglu_backward': ([_c_ptr] * 6 + [ctypes.c_longlong, _c_int, _c_ptr], _c_int),
    'spt_embedding_rows_backward_workspace_bytes': ([ctypes.c_longlong, _c_int], ctypes.c_longlong),
    'spt_embedding_rows_backward': ([_c_ptr, ctypes.c_longlong, _c_ptr, _c_ptr, _c_ptr, ctypes.c_longlong, _c_ptr,
                                     ctypes.c_longlong, _c_int, ctypes.c_longlong, _c_ptr], _c_int),
    'spt_rows_combine': ([_c_ptr] * 4 + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_rows_combine_side': ([_c_ptr] * 5 + [_c_int, _c_ptr] + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_ffn_coeff_grad': ([_c_ptr, _c_ptr, _c_int] + [_c_ptr] * 6 + [_c_f32, _c_ptr, _c_int, _c_int, _c_ptr],
                           _c_int),
    'spt_layernorm_partial_rows': ([ctypes.c_longlong], _c_int),
    'spt_add_layernorm_forward': ([_c_ptr] * 8 + [ctypes.c_longlong, _c_int, _c_f32, _c_int, _c_ptr], _c_int),
    'spt_layernorm_backward': ([_c_ptr] * 10 + [ctypes.c_longlong, _c_int, _c_int, _c_ptr], _c_int),
    'spt_cross_entropy_grad': ([_c_ptr, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_ptr, _c_ptr,
                                _c_ptr, ctypes.c_longlong, _c_ptr], _c_int),
    'spt_lora_down': ([_c_ptr, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_ptr, _c_int,
                       _c_ptr, ctypes.c_longlong, _c_int] + [_c_ptr] * 2 + [_c_int, _c_ptr], _c_int),
    'spt_lora_down2': ([_c_ptr, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_ptr, _c_int, _c_ptr, _c_int,
                        _c_ptr, ctypes.c_longlong, _c_int] + [_c_ptr] * 2 + [_c_int, _c_ptr], _c_int),
    'spt_lora_down_grouped_cols': ([_c_ptr, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_ptr, ctypes.c_longlong,
                                    _c_int, _c_ptr, _c_int, _c_ptr, ctypes.c_longlong, _c_ptr], _c_int),
    'spt_lora_down_tables': ([_c_ptr, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_ptr, _c_int,
                              ctypes.c_longlong] + [_c_ptr] * 3 + [_c_int, _c_ptr], _c_int),
    'spt_lora_down_grouped': ([_c_ptr, ctypes.c_longlong, ctypes.c_longlong, _c_int, _c_ptr, ctypes.c_longlong,
                               _c_int, _c_ptr, _c_int, _c_ptr, ctypes.c_longlong] + [_c_ptr] * 3, _c_int),
    'spt_tall_tn_workspace_bytes': ([ctypes.c_longlong, _c_int, _c_int, _c_int], ctypes.c_longlong),
    'spt_tall_tn_batch': ([_c_int, _c_ptr, ctypes.c_longlong, _c_ptr, ctypes.c_longlong, _c_ptr, _c_ptr, _c_int,
                           ctypes.c_longlong, _c_int, _c_int, _c_ptr, _c_int, _c_ptr, _c_ptr], _c_int),
    'spt_tall_tn': ([_c_ptr, ctypes.c_longlong, _c_ptr, ctypes.c_longlong, _c_ptr, _c_ptr, _c_int,
                     ctypes.c_longlong, _c_int, _c_int, _c_ptr, _c_int, _c_ptr, _c_ptr], _c_int),
    'spt_route_topk': ([_c_ptr] * 5 + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_route_topk_coeff': ([_c_ptr] * 8 + [_c_f32] + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_route_topk_logits': ([_c_ptr, _c_int] + [_c_ptr] * 9 + [_c_f32] + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_route_coeff_backward': ([_c_ptr] * 3 + [_c_f32, _c_ptr] + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_route_logit_backward': ([_c_ptr] * 3 + [_c_f32, _c_ptr, _c_ptr] + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_softmax_forward': ([_c_ptr] * 4 + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_softmax_backward': ([_c_ptr] * 5 + [_c_int] * 3 + [_c_ptr], _c_int),
    'spt_softmax_backward_clamped': ([_c_ptr] * 5 + [_c_f32, _c_f32, _c_ptr] + [_c_int] * 3 + [_c_ptr],
                                     _c_int),
}
ABI_VERSION = 39

_lib = None

# ---- which engine ran ---------------------------------------------------------------------------
# Every layer-level choice between a kernel of this library and something else (a library GEMM, a
# torch composition, another kernel of ours) is noted here: `PATH_COUNTS[(site, path)]` counts the
# calls (tests assert on it: a parity test must know WHICH engine it compared), and the first time a
# CUDA call leaves its primary path the reason is logged once (logger 'naive_gpt', WARNING).
PATH_COUNTS = collections.Counter()
_PATH_LOGGED = set()
_log = logging.getLogger('naive_gpt')


def note_path(site: str, path: str, fallback: bool = False, why='') -> None:
    """`why`: a string, or a callable that makes one (hot call sites: formatted only when logged)."""
    PATH_COUNTS[(site, path)] += 1
    if not fallback:
        return
    if callable(why):
        why = why()
    if (site, path, why) not in _PATH_LOGGED \
            and sum(1 for k in _PATH_LOGGED if k[:2] == (site, path)) < 4:     # (a few shapes per site)
        _PATH_LOGGED.add((site, path, why))
        _log.warning('naive_gpt: %s takes the %s path%s', site, path, ' (' + why + ')' if why else '')


def paths_taken(site: str = None) -> dict:
    """{(site, path): calls} so far (of one site when given)."""
    return {k: v for k, v in PATH_COUNTS.items() if site is None or k[0] == site}


def reset_paths() -> None:
    PATH_COUNTS.clear()


def load_library() -> ctypes.CDLL:
    """dlopen libspt_hip.so and bind the C ABI; raises if it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            'naive_gpt.ext: {} is missing -- build it with '
            '`make -C spt-proto_amd/csrc` (or __graft_entry__.build()); '
            'there is no CPU fallback'.format(LIB_PATH)
        )
    lib = ctypes.CDLL(LIB_PATH)
    for name, (argtypes, restype) in _PROTOTYPES.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = restype
    if lib.spt_abi_version() != ABI_VERSION:
        raise RuntimeError('naive_gpt.ext: libspt_hip.so ABI version mismatch')
    _lib = lib
    return lib


def _raise(lib, rc: int, what: str):
    msg = lib.spt_strerror(rc)
    msg = msg.decode() if msg else 'unknown'
    # launch failures surface like the reference's CUDA_CHECH (common.h:23-33)
    raise RuntimeError('{}: {} ({})'.format(what, msg, rc))


def _check_dim(x: torch.Tensor, dim: int, name: str):
    # CHECK_DIM, extension/common.h:13-18
    if not isinstance(x, torch.Tensor):
        raise TypeError('{} must be a torch.Tensor'.format(name))
    if not x.is_cuda:
        raise RuntimeError('{} must be a CUDA tensor'.format(name))
    if x.dim() != dim:
        raise RuntimeError('{} must be of dim {}'.format(name, dim))
    if not x.is_contiguous():
        raise RuntimeError(
            '{} custom kernel requires contiguous tensor'.format(name)
        )


def _check_type(x: torch.Tensor, dtype: torch.dtype, name: str):
    # CHECK_TYPE, extension/common.h:20-21
    if x.dtype != dtype:
        raise RuntimeError('{} must be type of {}'.format(name, dtype))


def _require(cond: bool, msg: str):
    # TORCH_CHECK
    if not cond:
        raise RuntimeError(msg)


def _same_device(*tensors):
    dev = tensors[0].device
    for t in tensors[1:]:
        _require(t.device == dev, 'all tensors must be on the same device')
    return dev


# Launches of this library enqueued so far (every wrapper asks `_stream` for its stream once).  A
# tuner that replays a captured step compares the count with what it was after its last replay:
# a difference means EAGER work -- of this model or of any other in the process -- went to the stream
# in between (utils/tuning.py: SparseTuner.training_step).
LAUNCHES = 0


def _stream(dev) -> int:
    # torch's current stream on `dev` as the raw hipStream_t (the C call behind
    # torch.cuda.current_stream(dev).cuda_stream, without the Stream object: ~0.3 us against 4 --
    # a fine-tune step makes ~600 of these calls)
    global LAUNCHES
    LAUNCHES += 1
    idx = dev.index
    return torch._C._cuda_getCurrentRawStream(idx if idx is not None else torch._C._cuda_getDevice())


def held_by_caches() -> list:
    """Every device tensor the module-level caches of this package hold for later calls: frozen
    weights' row norms and (SPT_WEIGHT_IMAGES=keep) images.  A captured HIP graph that read one of
    them reads its ADDRESS at every replay, whatever becomes of the cache entry (`drop_images()`,
    a parameter re-homed): `SparseTuner.capture` keeps this list alive for the life of its graph."""
    held = []
    for cache in (_WEIGHT_IMAGES, _WEIGHT_NORMS):
        held += [entry[2] for entry in cache.data.values()]
    return held


class _NoGuard:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NO_GUARD = _NoGuard()


def _on(dev):
    """`torch.cuda.device(dev)` only when `dev` is not already the current device (the guard's
    enter / exit cost ~5 us of host time per call; allocations and launches below go to `dev`)."""
    idx = dev.index
    if idx is None or idx == torch._C._cuda_getDevice():
        return _NO_GUARD
    return torch.cuda.device(dev)


def _flag(t) -> bool:
    # trans_lhs / trans_rhs are CPU bool scalars read on the host (sddmm.cpp:53-56)
    return bool(t.item()) if isinstance(t, torch.Tensor) else bool(t)


def cdist_forward_cuda(query: torch.Tensor, table: torch.Tensor):
    """extension/cdist.cu:185-250 -> [distance [M,NQ,C] f32, indices [M,NQ] i32]."""
    _check_dim(query, 3, 'query')
    _check_dim(table, 3, 'table')
    _check_type(query, torch.float32, 'query')
    _require(query.size(0) == table.size(0), 'query.size(0) == table.size(0)')
    _require(query.size(-1) == table.size(-1), 'query.size(-1) == table.size(-1)')
    _require(query.dtype == table.dtype, 'query.scalar_type() == table.scalar_type()')
    dev = _same_device(query, table)
    M, NQ, D = query.shape
    C = table.size(1)
    lib = load_library()
    with _on(dev):
        distance = torch.empty([M, NQ, C], dtype=torch.float32, device=dev)
        indices = torch.empty([M, NQ], dtype=torch.int32, device=dev)
        rc = lib.spt_cdist_forward(
            query.data_ptr(), table.data_ptr(), distance.data_ptr(),
            indices.data_ptr(), M, NQ, C, D, _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'cdist_forward_cuda')
    return [distance, indices]


def cdist_encode(query: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """Indices-only form of cdist_forward_cuda (distance not materialised)."""
    _check_dim(query, 3, 'query')
    _check_dim(table, 3, 'table')
    _check_type(query, torch.float32, 'query')
    _check_type(table, torch.float32, 'table')
    _require(query.size(0) == table.size(0), 'query.size(0) == table.size(0)')
    _require(query.size(-1) == table.size(-1), 'query.size(-1) == table.size(-1)')
    dev = _same_device(query, table)
    M, NQ, D = query.shape
    lib = load_library()
    with _on(dev):
        indices = torch.empty([M, NQ], dtype=torch.int32, device=dev)
        rc = lib.spt_cdist_forward(
            query.data_ptr(), table.data_ptr(), None, indices.data_ptr(),
            M, NQ, table.size(1), D, _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'cdist_encode')
    return indices


def cdist_backward_cuda(query: torch.Tensor, table: torch.Tensor,
                        grad_output: torch.Tensor):
    """extension/cdist.cu:252-333 -> [grad_query, grad_table]."""
    _check_dim(query, 3, 'query')
    _check_dim(table, 3, 'table')
    _check_dim(grad_output, 3, 'grad_output')
    _check_type(query, torch.float32, 'query')
    _require(query.size(0) == table.size(0), 'query.size(0) == table.size(0)')
    _require(query.size(-1) == table.size(-1), 'query.size(-1) == table.size(-1)')
    _require(query.size(0) == grad_output.size(0), 'query.size(0) == grad_output.size(0)')
    _require(query.size(1) == grad_output.size(1), 'query.size(1) == grad_output.size(1)')
    _require(table.size(1) == grad_output.size(-1), 'table.size(1) == grad_output.size(-1)')
    _require(query.dtype == table.dtype, 'query.scalar_type() == table.scalar_type()')
    _require(query.dtype == grad_output.dtype,
             'query.scalar_type() == grad_output.scalar_type()')
    dev = _same_device(query, table, grad_output)
    M, NQ, D = query.shape
    C = table.size(1)
    lib = load_library()
    with _on(dev):
        grad_query = torch.empty_like(query)
        grad_table = torch.empty_like(table)
        nbytes = lib.spt_cdist_backward_workspace_bytes(M, NQ, C, D)
        workspace = torch.empty([max(int(nbytes), 16)], dtype=torch.uint8, device=dev)
        rc = lib.spt_cdist_backward(
            query.data_ptr(), table.data_ptr(), grad_output.data_ptr(),
            grad_query.data_ptr(), grad_table.data_ptr(), workspace.data_ptr(),
            M, NQ, C, D, _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'cdist_backward_cuda')
    return [grad_query, grad_table]


def lookup_forward_cuda(config: torch.Tensor, query: torch.Tensor,
                        key: torch.Tensor) -> torch.Tensor:
    """extension/lookup.cu:87-174; ``config.size(0)`` is the sparsity coefficient."""
    _check_dim(key, 3, 'key')
    _check_dim(query, 3, 'query')
    _check_type(key, torch.int32, 'key')
    _check_type(query, torch.int32, 'query')
    _require(query.shape == key.shape, 'query.sizes() == key.sizes()')
    dev = _same_device(query, key)
    sparsity = int(config.size(0))
    B, S, M = query.shape
    _require(S % 16 == 0, 'seq_length % BLOCK_SIZE == 0')
    _require(sparsity > 0 and S % sparsity == 0, 'seq_length % sparsity == 0')
    Z = S // sparsity
    _require(Z % 16 == 0, 'nonzeros % BLOCK_SIZE == 0')
    lib = load_library()
    with _on(dev):
        output = torch.empty([B, S, Z], dtype=torch.int32, device=dev)
        rc = lib.spt_lookup_forward(
            query.data_ptr(), key.data_ptr(), output.data_ptr(),
            B, S, M, sparsity, _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'lookup_forward_cuda')
    return output


def _check_csr(indptr, indices):
    _check_dim(indptr, 1, 'indptr')
    _check_dim(indices, 2, 'indices')
    _check_type(indptr, torch.int32, 'indptr')
    _check_type(indices, torch.int32, 'indices')


def _dense_dims(t: torch.Tensor, heads: int, name: str):
    """(B, S, E) of a dense operand given as [B, S, E] (heads == 0) or [N, S, heads, E]."""
    if heads > 0:
        _check_dim(t, 4, name)
        _require(t.size(2) == heads, '{}.size(2) must equal heads'.format(name))
        return t.size(0) * heads, t.size(1), t.size(3)
    _check_dim(t, 3, name)
    return t.size(0), t.size(1), t.size(2)


def head_layout_supported(S: int, E: int, B: int) -> bool:
    """Shapes for which the kernels read / write the [N, S, H, E] layout directly."""
    return E in (64, 128) and S * E * 4 <= 128 * 1024 and B >= 32


def pq_encode_heads(z: torch.Tensor, table: torch.Tensor) -> torch.Tensor:
    """z [N, S, H, E] fp32 (or bf16 storage: widened exactly, same arithmetic) -> PQ codes
    [N * H, S, M] int32 (``spt_pq_encode_heads`` / ``_bf16``)."""
    _check_dim(z, 4, 'z')
    _check_dim(table, 3, 'table')
    _require(z.dtype in (torch.float32, torch.bfloat16), 'z: float32 or bfloat16')
    _check_type(table, torch.float32, 'table')
    dev = _same_device(z, table)
    N, S, H, E = z.shape
    M, C, D = table.shape
    _require(E == M * D, 'z.size(-1) == n_subspaces * d_codeword')
    lib = load_library()
    with _on(dev):
        codes = torch.empty([N * H, S, M], dtype=torch.int32, device=dev)
        fn = lib.spt_pq_encode_heads if z.dtype == torch.float32 else lib.spt_pq_encode_heads_bf16
        rc = fn(z.data_ptr(), table.data_ptr(), codes.data_ptr(), N, S, H, M, C, D, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'pq_encode_heads')
    return codes


def pq_loss_supported(z: torch.Tensor, table: torch.Tensor) -> bool:
    """Shapes the fused PQ training loss covers (``spt_pq_loss_*``, pq_loss.hip)."""
    if table.dim() != 3 or z.dim() < 2:
        return False
    M, C, D = table.shape
    return (z.is_cuda and z.dtype == torch.float32 and table.dtype == torch.float32
            and z.size(-1) == M * D and C == 16 and D in (4, 8)
            and 0 < M <= 32 and (M & (M - 1)) == 0 and z.numel() > 0)


def _pq_loss_args(z: torch.Tensor, table: torch.Tensor):
    _check_type(z, torch.float32, 'z')
    _check_type(table, torch.float32, 'table')
    _check_dim(table, 3, 'table')
    _require(z.is_contiguous() and table.is_contiguous(), 'z, table contiguous')
    dev = _same_device(z, table)
    M, C, D = table.shape
    _require(z.size(-1) == M * D, 'z.size(-1) == n_subspaces * d_codeword')
    n_vectors = z.numel() // (M * D)
    lib = load_library()
    nbytes = lib.spt_pq_loss_workspace_bytes(n_vectors, M, C, D)
    _require(nbytes > 0, 'pq_loss: unsupported codebook shape (C == 16, D in {4, 8}, M = 2^k)')
    return lib, dev, n_vectors, (M, C, D), nbytes


def pq_loss_forward(z: torch.Tensor, table: torch.Tensor, want_codes: bool = False):
    """-> 0-dim loss of ``PQBase.forward('train', z)`` (quantizer.py:80-111); with ``want_codes``
    (z [N, S, H, E]) also the PQ codes [N * H, S, M] of ``pq_encode_heads``, from the same pass."""
    lib, dev, n_vectors, (M, C, D), nbytes = _pq_loss_args(z, table)
    with _on(dev):
        loss = torch.empty([], dtype=torch.float32, device=dev)
        scratch = torch.empty([nbytes // 4], dtype=torch.float32, device=dev)
        if want_codes:
            _check_dim(z, 4, 'z')
            N, S, H, _ = z.shape
            codes = torch.empty([N * H, S, M], dtype=torch.int32, device=dev)
            rc = lib.spt_pq_loss_forward_codes(z.data_ptr(), table.data_ptr(), loss.data_ptr(),
                                               scratch.data_ptr(), codes.data_ptr(), N, S, H, M, C, D,
                                               _stream(dev))
        else:
            rc = lib.spt_pq_loss_forward(z.data_ptr(), table.data_ptr(), loss.data_ptr(),
                                         scratch.data_ptr(), n_vectors, M, C, D, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'pq_loss_forward')
    return (loss, codes) if want_codes else loss


def pq_loss_backward(z: torch.Tensor, table: torch.Tensor, grad_loss: torch.Tensor,
                     accumulate_into: torch.Tensor = None):
    """-> (grad_z, grad_table) for a 0-dim device ``grad_loss`` (no host read).  With
    ``accumulate_into`` (fp32, z's shape, contiguous) the gradient is ADDED to that tensor, which
    is returned as grad_z."""
    lib, dev, n_vectors, (M, C, D), nbytes = _pq_loss_args(z, table)
    _check_type(grad_loss, torch.float32, 'grad_loss')
    _require(grad_loss.numel() == 1 and grad_loss.device == z.device, 'grad_loss: device scalar')
    if accumulate_into is not None:
        _require(accumulate_into.dtype == torch.float32 and accumulate_into.shape == z.shape
                 and accumulate_into.is_contiguous() and accumulate_into.device == z.device,
                 'accumulate_into: contiguous fp32 tensor of z\'s shape')
    with _on(dev):
        grad_z = torch.empty_like(z) if accumulate_into is None else accumulate_into
        grad_table = torch.empty_like(table)
        scratch = torch.empty([nbytes // 4], dtype=torch.float32, device=dev)
        rc = lib.spt_pq_loss_backward(z.data_ptr(), table.data_ptr(), grad_loss.data_ptr(),
                                      grad_z.data_ptr(), grad_table.data_ptr(),
                                      scratch.data_ptr(), n_vectors, M, C, D,
                                      int(accumulate_into is not None), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'pq_loss_backward')
    return grad_z, grad_table


def back_to_back(a: torch.Tensor, b: torch.Tensor) -> bool:
    """b starts where a ends, in the same allocation (both contiguous, same shape and dtype)"""
    return (a.shape == b.shape and a.dtype == b.dtype and a.is_contiguous() and b.is_contiguous()
            and a.device == b.device and b.data_ptr() == a.data_ptr() + a.numel() * a.element_size()
            and a.untyped_storage().data_ptr() == b.untyped_storage().data_ptr())


def _as_pair(a: torch.Tensor) -> torch.Tensor:
    """[2, *a.shape]: `a` and the tensor that lies behind it (`back_to_back`)"""
    return a.as_strided((2,) + tuple(a.shape), (a.numel(),) + tuple(a.stride()))


def pq_loss_forward_pair(q: torch.Tensor, k: torch.Tensor, table: torch.Tensor):
    """``spt_pq_loss_forward_codes_parts`` for q, k [N, S, H, E] that lie `back_to_back`:
    -> (loss_q + loss_k, codes_q, codes_k) from one pass over both."""
    _require(back_to_back(q, k), 'pq_loss_forward_pair: k behind q in one buffer')
    _check_dim(q, 4, 'q')
    lib, dev, n_vectors, (M, C, D), nbytes = _pq_loss_args(_as_pair(q), table)
    N, S, H, _ = q.shape
    with _on(dev):
        loss = torch.empty([], dtype=torch.float32, device=dev)
        scratch = torch.empty([nbytes // 4], dtype=torch.float32, device=dev)
        codes = torch.empty([2, N * H, S, M], dtype=torch.int32, device=dev)
        rc = lib.spt_pq_loss_forward_codes_parts(q.data_ptr(), table.data_ptr(), loss.data_ptr(),
                                                 scratch.data_ptr(), codes.data_ptr(), 2, N, S, H, M, C, D,
                                                 _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'pq_loss_forward_pair')
    return loss, codes[0], codes[1]


def pq_loss_backward_pair(q: torch.Tensor, k: torch.Tensor, table: torch.Tensor, grad_loss: torch.Tensor,
                          grad_q: torch.Tensor, grad_k: torch.Tensor):
    """``spt_pq_loss_backward_parts``: the gradient of loss_q + loss_k ADDED into grad_q / grad_k (each
    pair `back_to_back`) in one pass -> grad_table (the sum of both)."""
    _require(back_to_back(q, k) and back_to_back(grad_q, grad_k) and grad_q.shape == q.shape
             and grad_q.dtype == torch.float32, 'pq_loss_backward_pair: (q, k) and (grad_q, grad_k) back to back')
    lib, dev, n_vectors, (M, C, D), nbytes = _pq_loss_args(_as_pair(q), table)
    _check_type(grad_loss, torch.float32, 'grad_loss')
    _require(grad_loss.numel() == 1 and grad_loss.device == q.device, 'grad_loss: device scalar')
    with _on(dev):
        grad_table = torch.empty_like(table)
        scratch = torch.empty([nbytes // 4], dtype=torch.float32, device=dev)
        rc = lib.spt_pq_loss_backward_parts(q.data_ptr(), table.data_ptr(), grad_loss.data_ptr(),
                                            grad_q.data_ptr(), grad_table.data_ptr(), scratch.data_ptr(),
                                            n_vectors // 2, 2, M, C, D, 1, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'pq_loss_backward_pair')
    return grad_table


def sddmm_forward_cuda(trans_lhs, trans_rhs, indptr: torch.Tensor,
                       indices: torch.Tensor, query: torch.Tensor,
                       key: torch.Tensor, scale: float = 1.0,
                       clamp: float = 0.0, query_heads: int = 0,
                       key_heads: int = 0) -> torch.Tensor:
    """extension/sddmm.cpp:3-73.  ``scale``/``clamp`` (extension of the reference
    signature, defaults = plain operator) fold attention.py:125-127 into the store."""
    B, S, E = _dense_dims(query, query_heads, 'query')
    _require((B, S, E) == _dense_dims(key, key_heads, 'key'), 'query.sizes() == key.sizes()')
    _check_csr(indptr, indices)
    _check_type(query, torch.float32, 'query')
    _require(B == indices.size(0), 'query.size(0) == indices.size(0)')
    _require(query.dtype == key.dtype, 'query.scalar_type() == key.scalar_type()')
    _require(not _flag(trans_lhs) and _flag(trans_rhs),
             'sddmm: only (trans_lhs=False, trans_rhs=True) is implemented')
    dev = _same_device(indptr, indices, query, key)
    nnz = indices.size(-1)
    _require(indptr.size(-1) == S + 1, 'indptr.size(-1) == seq_length + 1')
    lib = load_library()
    note_path('sddmm', 'matrix_cores' if lib.spt_sddmm_form(B, S, E, nnz) else 'gather')
    with _on(dev):
        output = torch.empty([B, nnz], dtype=torch.float32, device=dev)
        rc = lib.spt_sddmm_forward(
            indptr.data_ptr(), indices.data_ptr(), query.data_ptr(),
            key.data_ptr(), output.data_ptr(), B, S, E, nnz,
            float(scale), float(clamp), int(query_heads), int(key_heads), _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'sddmm_forward_cuda')
    return output


def _check_spmm(indptr, indices, values, x, x_heads=0):
    B, S, E = _dense_dims(x, x_heads, 'x')
    _check_csr(indptr, indices)
    _check_dim(values, 2, 'values')
    _check_type(values, torch.float32, 'values')
    _check_type(x, torch.float32, 'x')
    _require(indices.shape == values.shape, 'indices.sizes() == values.sizes()')
    _require(B == indices.size(0), 'x.size(0) == indices.size(0)')
    _require(indptr.size(-1) == S + 1, 'indptr.size(-1) == seq_length + 1')
    return _same_device(indptr, indices, values, x), B, S, E


def _alloc_dense(B, S, E, heads, dev):
    if heads > 0:
        return torch.empty([B // heads, S, heads, E], dtype=torch.float32, device=dev)
    return torch.empty([B, S, E], dtype=torch.float32, device=dev)


def fused_attention_supported(S: int, E: int, B: int, nnz: int) -> bool:
    """Shapes ``spt_sparse_attention_forward`` covers (fused_attention.hip)."""
    if S <= 0 or nnz % S != 0:
        return False
    Z = nnz // S
    return E == 64 and 0 < Z <= 64 and Z % 4 == 0 and S * E * 4 <= 128 * 1024 and B >= 32


def sparse_attention_forward(indices: torch.Tensor, q: torch.Tensor, k: torch.Tensor,
                             v: torch.Tensor, scale: float, clamp: float,
                             y_transposed: bool = False, causal: bool = False):
    """One launch for sddmm -> scale, clamp -> softmax -> spmm on uniform CSR rows.

    q, k, v: ``[N, S, H, E]`` (head layout); indices ``[N*H, nnz]``.  Returns
    ``(scores, attn, y)``: the clamped scores and probabilities ``[N*H, nnz]`` and y as
    ``[N*H, S, E]`` or, ``y_transposed``, ``[N*H, E, S]``.  ``causal`` promises column <= row
    for every entry (lookup's output): K / V are then streamed into LDS just ahead of the rows
    that use them."""
    _check_dim(q, 4, 'q')
    _check_type(q, torch.float32, 'q')
    _check_type(indices, torch.int32, 'indices')
    _require(q.shape == k.shape == v.shape, 'q, k, v: same shape')
    _require(q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
             and indices.is_contiguous(), 'contiguous operands')
    dev = _same_device(indices, q, k, v)
    N, S, H, E = q.shape
    B, nnz = N * H, indices.size(-1)
    _require(indices.dim() == 2 and indices.size(0) == B, 'indices: [N * H, nnz]')
    lib = load_library()
    with _on(dev):
        scores = torch.empty([B, nnz], dtype=torch.float32, device=dev)
        attn = torch.empty([B, nnz], dtype=torch.float32, device=dev)
        y = torch.empty([B, E, S] if y_transposed else [B, S, E], dtype=torch.float32, device=dev)
        rc = lib.spt_sparse_attention_forward(
            indices.data_ptr(), q.data_ptr(), k.data_ptr(), v.data_ptr(), scores.data_ptr(),
            attn.data_ptr(), y.data_ptr(), B, S, E, nnz, float(scale), float(clamp), H,
            int(bool(y_transposed)), int(bool(causal)), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'sparse_attention_forward')
    return scores, attn, y


def attention_mfma_supported(S: int, E: int, nnz: int) -> bool:
    """Shapes the matrix-core attention kernels cover (mfma_attention.hip)."""
    return bool(load_library().spt_attention_mfma_supported(int(S), int(E), int(nnz)))


class MfmaTiles:
    """The CSR entries bucketed by (32-row tile, 32-key tile): ``spt_attention_mfma_prepare``.
    Depends on ``indices`` only; shared by the forward and the backward of a layer step."""

    __slots__ = ('buffer', 'batch', 'seq', 'nnz', 'layout')

    def __init__(self, buffer, batch, seq, nnz, layout):
        self.buffer, self.batch, self.seq, self.nnz, self.layout = buffer, batch, seq, nnz, layout

    def broke_promise(self) -> bool:
        """(synchronises) compact layout: did the pattern repeat a column outside key tile 0?"""
        return bool(self.buffer[4:8].view(torch.int32).item() & 1)


TILES_FULL, TILES_COMPACT = 0, 1
# (tests) False: the backward decides the clamp's gradient mask on its split-bf16 scores -- the
# behaviour before ABI 39, kept switchable so that a test can show what the exact mask changes
EXACT_CLAMP = True


def attention_mfma_prepare(indices: torch.Tensor, seq_length: int,
                           lookup_pattern: bool = False) -> MfmaTiles:
    """``lookup_pattern``: the caller vouches that only columns < 32 repeat inside a row -- true
    of every output of ``lookup_forward_cuda``, whose one repeated column is the padding
    column 0 -- which allows the compact tile layout (a quarter of the memory)."""
    layout = TILES_COMPACT if lookup_pattern else TILES_FULL
    _check_dim(indices, 2, 'indices')
    _check_type(indices, torch.int32, 'indices')
    _require(indices.is_contiguous(), 'contiguous indices')
    dev = _same_device(indices)
    B, nnz = indices.shape
    lib = load_library()
    size = lib.spt_attention_mfma_tiles_bytes(B, int(seq_length), nnz, layout)
    _require(size > 0, 'attention_mfma: unsupported shape (d_head 64, Z <= 256, Z % 4 == 0, S <= 2048)')
    with _on(dev):
        buf = torch.empty([size], dtype=torch.uint8, device=dev)
        rc = lib.spt_attention_mfma_prepare(indices.data_ptr(), buf.data_ptr(), B, int(seq_length),
                                            nnz, layout, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'attention_mfma_prepare')
    return MfmaTiles(buf, B, int(seq_length), nnz, layout)


def _bounds_behind(row_sum: torch.Tensor, B: int, S: int, n_bounds: int) -> int:
    """Device address of the norm bounds that :func:`attention_mfma_forward` left behind the row
    sums (0: `row_sum` is some other [B, S] tensor -- the backward then decides the clamp mask on
    its split-bf16 scores, as before ABI 39)."""
    have = row_sum.untyped_storage().nbytes() - 4 * row_sum.storage_offset()
    return row_sum.data_ptr() + 4 * B * S if have >= 4 * (B * S + n_bounds) else 0


def attention_mfma_forward(tiles, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                           scale: float, clamp: float, y_transposed: bool = False):
    """The attention core on the matrix cores (``spt_attention_mfma_forward``).

    q, k, v: ``[N, S, H, E]``; ``tiles``: :func:`attention_mfma_prepare` of the ``[N*H, nnz]``
    indices (or the indices themselves).  Returns ``(y, row_sum)``: y ``[N*H, S, E]`` or
    (``y_transposed``) ``[N*H, E, S]`` and the softmax denominators ``[N*H, S]`` the backward
    needs; no ``[N*H, nnz]`` array is produced.

    ``row_sum`` is a view of the first ``N*H*S`` floats of a slightly longer buffer: behind them
    lie the forward's bounds on the row norms of q and k (``spt_attention_mfma_bounds_floats``),
    with which :func:`attention_mfma_backward` decides the clamp's gradient mask exactly.  Hand the
    SAME tensor (or one sharing its storage: ``save_for_backward`` does) to the backward."""
    _check_dim(q, 4, 'q')
    _require(q.dtype in (torch.float32, torch.bfloat16), 'q: float32 or bfloat16 (bf16 storage)')
    _require(q.dtype == k.dtype == v.dtype, 'q, k, v: same dtype')
    _require(q.shape == k.shape == v.shape, 'q, k, v: same shape')
    _require(q.is_contiguous() and k.is_contiguous() and v.is_contiguous(), 'contiguous operands')
    N, S, H, E = q.shape
    B = N * H
    if isinstance(tiles, torch.Tensor):
        _require(E in (64, 128), 'attention_mfma: d_head 64 or 128')
        tiles = attention_mfma_prepare(tiles, S)
    _require(tiles.batch == B and tiles.seq == S, 'tiles: prepared for [N * H, nnz] indices at S')
    dev = _same_device(tiles.buffer, q, k, v)
    lib = load_library()
    with _on(dev):
        y = torch.empty([B, E, S] if y_transposed else [B, S, E], dtype=q.dtype, device=dev)
        stats = torch.empty([B * S + lib.spt_attention_mfma_bounds_floats(B)], dtype=torch.float32,
                            device=dev)
        row_sum = stats[:B * S].view(B, S)
        fn = (lib.spt_attention_mfma_forward if q.dtype == torch.float32
              else lib.spt_attention_mfma_forward_bf16)
        rc = fn(
            tiles.buffer.data_ptr(), tiles.layout, q.data_ptr(), k.data_ptr(), v.data_ptr(), y.data_ptr(),
            row_sum.data_ptr(), (row_sum.data_ptr() + 4 * B * S) if EXACT_CLAMP else 0, B, S, E, tiles.nnz, float(scale),
            float(clamp), H, int(bool(y_transposed)), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'attention_mfma_forward')
    return y, row_sum


def attention_mfma_backward(tiles: MfmaTiles, q: torch.Tensor, k: torch.Tensor, v: torch.Tensor,
                            y: torch.Tensor, grad_y: torch.Tensor, row_sum: torch.Tensor,
                            scale: float, clamp: float, transposed: bool = False):
    """Backward of :func:`attention_mfma_forward` (``spt_attention_mfma_backward``).

    y, grad_y: the forward's output and its gradient, ``[N*H, S, E]`` or (``transposed``)
    ``[N*H, E, S]``.  Returns ``(grad_q, grad_k, grad_v)``, each ``[N, S, H, E]`` -- three views of
    ONE ``[3, N, S, H, E]`` buffer (the joint dX product of the projections reads them as one
    operand), so a surviving reference to any one of them keeps all three alive.

    ``row_sum``: the tensor the forward returned for the same q, k (see there: the norm bounds
    behind it make the clamp's gradient mask that of the fp32 scores; ``EXACT_CLAMP = False``, or
    a ``row_sum`` without them, leaves the mask to the split-bf16 scores)."""
    _check_dim(q, 4, 'q')
    _require(q.dtype in (torch.float32, torch.bfloat16), 'q: float32 or bfloat16 (bf16 storage)')
    _require(q.dtype == k.dtype == v.dtype == y.dtype == grad_y.dtype, 'q, k, v, y, grad_y: same dtype')
    _check_type(row_sum, torch.float32, 'row_sum')
    _require(q.shape == k.shape == v.shape, 'q, k, v: same shape')
    for t in (q, k, v, y, grad_y, row_sum):
        _require(t.is_contiguous(), 'contiguous operands')
    N, S, H, E = q.shape
    B = N * H
    _require(tiles.batch == B and tiles.seq == S, 'tiles: prepared for [N * H, nnz] indices at S')
    _require(y.numel() == B * S * E and grad_y.numel() == B * S * E, 'y, grad_y: N * H * S * E elements')
    _require(row_sum.numel() == B * S, 'row_sum: [N * H, S]')
    dev = _same_device(tiles.buffer, q, k, v, y, grad_y, row_sum)
    lib = load_library()
    with _on(dev):
        # (one buffer: the projections' backward contracts the three as ONE matrix product when
        # they are equally spaced in memory -- layers/tuning/lora.py, `qkv_dx`)
        grad_q, grad_k, grad_v = torch.empty([3, *q.shape], dtype=q.dtype, device=dev).unbind(0)
        delta = torch.empty([2, B, S], dtype=torch.float32, device=dev)   # (scratch: spt_hip.h)
        fn = (lib.spt_attention_mfma_backward if q.dtype == torch.float32
              else lib.spt_attention_mfma_backward_bf16)
        bounds = _bounds_behind(row_sum, B, S, lib.spt_attention_mfma_bounds_floats(B)) if EXACT_CLAMP else 0
        rc = fn(
            tiles.buffer.data_ptr(), tiles.layout, q.data_ptr(), k.data_ptr(), v.data_ptr(), y.data_ptr(),
            grad_y.data_ptr(), row_sum.data_ptr(), bounds, delta.data_ptr(), grad_q.data_ptr(),
            grad_k.data_ptr(), grad_v.data_ptr(), B, S, E, tiles.nnz, float(scale), float(clamp),
            H, int(bool(transposed)), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'attention_mfma_backward')
    return grad_q, grad_k, grad_v


def sparse_attention_backward_rows(indices: torch.Tensor, grad_y: torch.Tensor, v: torch.Tensor,
                                   k: torch.Tensor, scores: torch.Tensor, attn: torch.Tensor,
                                   scale: float, clamp: float, grad_y_transposed: bool = False,
                                   causal: bool = False):
    """Row-wise half of the fused attention backward (``spt_sparse_attention_backward_rows``).

    v, k: ``[N, S, H, E]``; grad_y: ``[N*H, S, E]`` or (``grad_y_transposed``) ``[N*H, E, S]``.
    Returns ``(grad_raw [N*H, nnz], grad_q [N, S, H, E], grad_y_rows [N*H, S, E])`` -- the
    last is ``grad_y`` itself when it was not transposed."""
    _check_dim(v, 4, 'v')
    _check_type(v, torch.float32, 'v')
    _check_type(indices, torch.int32, 'indices')
    _require(v.shape == k.shape, 'v, k: same shape')
    for t in (indices, grad_y, v, k, scores, attn):
        _require(t.is_contiguous(), 'contiguous operands')
    dev = _same_device(indices, grad_y, v, k, scores, attn)
    N, S, H, E = v.shape
    B, nnz = N * H, indices.size(-1)
    _require(grad_y.numel() == B * S * E, 'grad_y: N * H * S * E elements')
    _require(scores.shape == attn.shape == indices.shape, 'scores, attn: the shape of indices')
    lib = load_library()
    with _on(dev):
        grad_raw = torch.empty([B, nnz], dtype=torch.float32, device=dev)
        grad_q = torch.empty_like(v)
        rows = torch.empty([B, S, E], dtype=torch.float32, device=dev) if grad_y_transposed \
            else grad_y.view(B, S, E)
        rc = lib.spt_sparse_attention_backward_rows(
            indices.data_ptr(), grad_y.data_ptr(), v.data_ptr(), k.data_ptr(),
            scores.data_ptr(), attn.data_ptr(), grad_raw.data_ptr(), grad_q.data_ptr(),
            rows.data_ptr() if grad_y_transposed else None, B, S, E, nnz, float(scale),
            float(clamp), H, int(bool(grad_y_transposed)), int(bool(causal)), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'sparse_attention_backward_rows')
    return grad_raw, grad_q, rows


def csr_transpose(indptr: torch.Tensor, indices: torch.Tensor, d_head: int = 64) -> torch.Tensor:
    """Transposed structure of a batched CSR pattern, as an opaque uint8 buffer for
    ``spmm_transposed``.  Depends on (indptr, indices) and on the head size of the products
    that will use it (it selects the flat or the chunked form, include/spt_hip.h), so one
    build serves every A^T product of a backward pass."""
    _check_csr(indptr, indices)
    dev = _same_device(indptr, indices)
    B, nnz = indices.shape
    S = indptr.size(-1) - 1
    lib = load_library()
    with _on(dev):
        nbytes = lib.spt_csr_transpose_workspace_bytes(B, S, nnz, int(d_head))
        buf = torch.empty([max(int(nbytes), 16)], dtype=torch.uint8, device=dev)
        if nnz > 0:
            rc = lib.spt_csr_transpose(indptr.data_ptr(), indices.data_ptr(),
                                       buf.data_ptr(), B, S, nnz, int(d_head), _stream(dev))
            if rc != 0:
                _raise(lib, rc, 'csr_transpose')
    return buf


def transposed_for(indptr: torch.Tensor, indices: torch.Tensor, d_head: int = 64) -> torch.Tensor:
    """``csr_transpose`` memoised on the ``indices`` tensor object itself (keyed by the
    tensors' in-place version counters and the head size), so the A^T products of one
    backward pass -- grad_K in sddmm's backward, grad_V in spmm's -- share one build.  The
    cache dies with the tensor."""
    key = (indices._version, indptr.data_ptr(), indptr._version, int(d_head))
    cached = getattr(indices, '_spt_transposed', None)
    if cached is not None and cached[0] == key:
        return cached[1]
    buf = csr_transpose(indptr, indices, d_head)
    indices._spt_transposed = (key, buf)
    return buf


def spmm_transposed(transposed: torch.Tensor, indptr: torch.Tensor,
                    indices: torch.Tensor, values: torch.Tensor,
                    x: torch.Tensor, x_heads: int = 0, y_heads: int = 0) -> torch.Tensor:
    """y = A^T . x with a structure from ``csr_transpose`` built for this head size (same
    result as ``spmm_forward_cuda(True, False, ...)``)."""
    dev, B, S, E = _check_spmm(indptr, indices, values, x, x_heads)
    nnz = indices.size(-1)
    lib = load_library()
    with _on(dev):
        output = _alloc_dense(B, S, E, y_heads, dev)
        if nnz == 0:
            return output.zero_()
        _require(transposed.numel() >= lib.spt_csr_transpose_workspace_bytes(B, S, nnz, E),
                 'spmm_transposed: the structure was built for another head size')
        nbytes = lib.spt_spmm_transposed_workspace_bytes(B, S, E, nnz)
        scratch = torch.empty([max(int(nbytes), 16)], dtype=torch.uint8, device=dev)
        rc = lib.spt_spmm_transposed(indptr.data_ptr(), transposed.data_ptr(), values.data_ptr(),
                                     x.data_ptr(), output.data_ptr(), scratch.data_ptr(),
                                     B, S, E, nnz, int(x_heads), int(y_heads), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'spmm_transposed')
    return output


def spmm_forward_cuda(trans_lhs, trans_rhs, indptr: torch.Tensor,
                      indices: torch.Tensor, values: torch.Tensor,
                      x: torch.Tensor, x_heads: int = 0, y_heads: int = 0) -> torch.Tensor:
    """extension/spmm.cpp:3-72; ``trans_lhs`` selects A.x (False) or A^T.x (True)."""
    dev, B, S, E = _check_spmm(indptr, indices, values, x, x_heads)
    _require(not _flag(trans_rhs), 'spmm: trans_rhs=True is not implemented')
    trans = int(_flag(trans_lhs))
    nnz = indices.size(-1)
    lib = load_library()
    with _on(dev):
        output = _alloc_dense(B, S, E, y_heads, dev)
        if nnz == 0:
            return output.zero_()
        workspace = None
        note_path('spmm_t' if trans else 'spmm', 'matrix_cores' if lib.spt_spmm_form(trans, B, S, E, nnz) else 'gather')
        if trans:
            nbytes = lib.spt_spmm_workspace_bytes(trans, B, S, E, nnz)
            workspace = torch.empty([max(int(nbytes), 16)], dtype=torch.uint8, device=dev)
        rc = lib.spt_spmm_forward(
            trans, indptr.data_ptr(), indices.data_ptr(), values.data_ptr(),
            x.data_ptr(), output.data_ptr(),
            workspace.data_ptr() if workspace is not None else None,
            B, S, E, nnz, int(x_heads), int(y_heads), _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'spmm_forward_cuda')
    return output


def softmax_forward_cuda(indptr: torch.Tensor, indices: torch.Tensor,
                         values: torch.Tensor) -> torch.Tensor:
    """extension/softmax.cu:84-114."""
    _check_csr(indptr, indices)
    _check_dim(values, 2, 'values')
    _check_type(values, torch.float32, 'values')
    _require(indices.shape == values.shape, 'indices.sizes() == values.sizes()')
    dev = _same_device(indptr, indices, values)
    B, nnz = indices.shape
    S = indptr.size(-1) - 1
    lib = load_library()
    with _on(dev):
        output = torch.empty_like(values)
        rc = lib.spt_softmax_forward(
            indptr.data_ptr(), indices.data_ptr(), values.data_ptr(),
            output.data_ptr(), B, S, nnz, _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'softmax_forward_cuda')
    return output


def softmax_backward_cuda(indptr: torch.Tensor, indices: torch.Tensor,
                          output: torch.Tensor,
                          grad_output: torch.Tensor) -> torch.Tensor:
    """extension/softmax.cu:116-148."""
    _check_csr(indptr, indices)
    _check_dim(output, 2, 'output')
    _check_dim(grad_output, 2, 'grad_output')
    _check_type(output, torch.float32, 'output')
    _check_type(grad_output, torch.float32, 'grad_output')
    _require(indices.shape == output.shape, 'indices.sizes() == output.sizes()')
    _require(indices.shape == grad_output.shape,
             'indices.sizes() == grad_output.sizes()')
    dev = _same_device(indptr, indices, output, grad_output)
    B, nnz = indices.shape
    S = indptr.size(-1) - 1
    lib = load_library()
    with _on(dev):
        grad_values = torch.empty_like(output)
        rc = lib.spt_softmax_backward(
            indptr.data_ptr(), indices.data_ptr(), output.data_ptr(),
            grad_output.data_ptr(), grad_values.data_ptr(), B, S, nnz,
            _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'softmax_backward_cuda')
    return grad_values


def grouped_gemm(a: torch.Tensor, weight: torch.Tensor, offsets: torch.Tensor,
                 n_groups: int, n: int, k: int, w_group_stride: int, w_ldn: int,
                 w_ldk: int, gather: torch.Tensor = None, bias: torch.Tensor = None,
                 rowscale: torch.Tensor = None, n_rows: int = None) -> torch.Tensor:
    """Token-bucketed grouped GEMM (``spt_grouped_gemm``): for the rows of bucket g,
    ``out = rowscale * (a[gather] @ W_g^T + bias[g])`` with
    ``W_g(n, k) = weight.flatten()[g * w_group_stride + n * w_ldn + k * w_ldk]``.
    ``offsets`` [n_groups + 1] int32 stays on the device."""
    _check_type(a, torch.float32, 'a')
    _check_type(weight, torch.float32, 'weight')
    _check_type(offsets, torch.int32, 'offsets')
    _require(a.is_cuda and weight.is_cuda and offsets.is_cuda, 'grouped_gemm needs CUDA tensors')
    _require(a.dim() == 2 and a.stride(1) == 1, 'a must be [rows, k] with unit inner stride')
    _require(weight.is_contiguous(), 'weight must be contiguous')
    _require(offsets.numel() == n_groups + 1, 'offsets must have n_groups + 1 entries')
    dev = _same_device(a, weight, offsets)
    if n_rows is None:
        n_rows = gather.numel() if gather is not None else a.size(0)
    for t, name in ((gather, 'gather'), (bias, 'bias'), (rowscale, 'rowscale')):
        if t is not None:
            _require(t.is_cuda and t.is_contiguous(), name + ' must be a contiguous CUDA tensor')
    if gather is not None:
        _check_type(gather, torch.int32, 'gather')
    lib = load_library()
    with _on(dev):
        out = torch.empty([n_rows, n], dtype=torch.float32, device=dev)
        if n_rows == 0:
            return out
        rc = lib.spt_grouped_gemm(
            a.data_ptr(), gather.data_ptr() if gather is not None else None,
            weight.data_ptr(), bias.data_ptr() if bias is not None else None,
            rowscale.data_ptr() if rowscale is not None else None,
            offsets.data_ptr(), out.data_ptr(), n_rows, k, n, n_groups, a.stride(0),
            w_group_stride, w_ldn, w_ldk, _stream(dev)
        )
    if rc != 0:
        _raise(lib, rc, 'grouped_gemm')
    return out


class _GroupedDesc(ctypes.Structure):
    """``SptGroupedGemm`` of include/spt_hip.h."""
    _fields_ = [
        ('a', _c_ptr), ('gather', _c_ptr), ('w', _c_ptr), ('bias', _c_ptr),
        ('rowscale', _c_ptr), ('offsets', _c_ptr), ('out', _c_ptr),
        ('n_rows', ctypes.c_int32), ('k', ctypes.c_int32), ('n', ctypes.c_int32),
        ('n_groups', ctypes.c_int32), ('lda', ctypes.c_int32),
        ('w_group_stride', ctypes.c_int64), ('w_ldn', ctypes.c_int32), ('w_ldk', ctypes.c_int32),
        ('a2', _c_ptr), ('gather2', _c_ptr), ('b2', _c_ptr),
        ('lda2', ctypes.c_int32), ('r', ctypes.c_int32),
        ('b2_group_stride', ctypes.c_int64), ('b2_ldn', ctypes.c_int32),
        ('epilogue', ctypes.c_int32), ('activation', ctypes.c_int32),
        ('out2', _c_ptr), ('h_in', _c_ptr), ('s_in', _c_ptr),
        ('pdot_main', _c_ptr), ('pdot_act', _c_ptr), ('pdot_ld', ctypes.c_int32),
        ('a_image', _c_ptr), ('w_image', _c_ptr), ('a_norm', _c_ptr), ('w_norm', _c_ptr),
        ('relu_queue', _c_ptr), ('relu_queue_bytes', ctypes.c_int64), ('ldo', ctypes.c_int64),
        ('accumulate', ctypes.c_int32), ('a_seg_k', ctypes.c_int32), ('a_seg_stride', ctypes.c_int64),
        ('b2_seg_stride', ctypes.c_int64),
    ]


EPI_PLAIN, EPI_ACT, EPI_DACT = 0, 1, 2
LAST_GEMM_USED_IMAGES = False      # which operand path the last grouped_gemm_fused call took
LAST_GEMM_PATH = 'register'        # ... by name: 'image' | 'a32' | 'register'
LAST_RELU_QUEUE = None              # (tests) the near-the-kink queue of the last ReLU GEMM
ACT_RELU, ACT_GELU, ACT_SILU = 0, 1, 2


def _ptr(t):
    return t.data_ptr() if t is not None else None


class SplitImage:
    """The pre-split bf16 image of an fp32 matrix (``spt_split_bf16``): every 32 columns of a
    row as one 128-byte block [hi | lo].  Keeps the source's shape for the checks."""
    __slots__ = ('buffer', 'rows', 'cols')

    def __init__(self, buffer, rows, cols):
        self.buffer, self.rows, self.cols = buffer, rows, cols


def split_bf16(x: torch.Tensor) -> SplitImage:
    """fp32 [rows, cols] (unit inner stride) -> its hi / lo bf16 image, one launch."""
    _check_type(x, torch.float32, 'x')
    _require(x.is_cuda and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0,
             'split_bf16: CUDA fp32 matrix with unit inner stride and 16-byte aligned rows')
    rows, cols = x.shape
    lib = load_library()
    dev = x.device
    with _on(dev):
        buf = torch.empty([lib.spt_split_bf16_bytes(rows, cols)], dtype=torch.uint8, device=dev)
        if rows > 0:
            rc = lib.spt_split_bf16(x.data_ptr(), buf.data_ptr(), rows, cols, x.stride(0),
                                    _stream(dev))
            if rc != 0:
                _raise(lib, rc, 'split_bf16')
    return SplitImage(buf, rows, cols)


# Images of the activations most recently split.  The q / k / v projections of a block read ONE
# tensor: the first splits it, the other two find the image here.  An entry is identified by
# the OWNER object (the tensor the layer was called with, held through a weak reference -- no
# lifetime is extended, and an address handed to another tensor cannot alias) and its
# `_version` (every in-place write moves it).  Two entries: they are as large as activations.
_IMAGE_CACHE = collections.OrderedDict()
IMAGE_CACHE_ENTRIES = 2


def image_of(x: torch.Tensor, owner: torch.Tensor = None) -> SplitImage:
    """`split_bf16(x)` through the small cache above; `owner`: the tensor object `x` is a view
    of (default: x itself) -- what callers that share an input have in common."""
    owner = x if owner is None else owner
    if owner.is_inference():
        return split_bf16(x)
    key = id(owner)
    hit = _IMAGE_CACHE.get(key)
    if hit is not None and hit[0]() is owner and hit[1] == (owner._version, tuple(x.shape)):
        _IMAGE_CACHE.move_to_end(key)
        return hit[2]
    img = split_bf16(x)
    _IMAGE_CACHE[key] = (weakref.ref(owner, lambda _, k=key: _IMAGE_CACHE.pop(k, None)),
                         (owner._version, tuple(x.shape)), img)
    _IMAGE_CACHE.move_to_end(key)
    while len(_IMAGE_CACHE) > IMAGE_CACHE_ENTRIES:
        _IMAGE_CACHE.popitem(last=False)
    return img


def cached_image(x: torch.Tensor, owner: torch.Tensor = None):
    """The cached image `image_of(x, owner)` would return, or None."""
    owner = x if owner is None else owner
    if owner.is_inference():
        return None
    hit = _IMAGE_CACHE.get(id(owner))
    if hit is not None and hit[0]() is owner and hit[1] == (owner._version, tuple(x.shape)):
        return hit[2]
    return None


def put_image(x: torch.Tensor, owner: torch.Tensor, img: SplitImage) -> None:
    """Hand the cache an image made elsewhere (``lora_down(..., want_image=True)``)."""
    owner = x if owner is None else owner
    if owner.is_inference():
        return
    key = id(owner)
    _IMAGE_CACHE[key] = (weakref.ref(owner, lambda _, k=key: _IMAGE_CACHE.pop(k, None)),
                         (owner._version, tuple(x.shape)), img)
    _IMAGE_CACHE.move_to_end(key)
    while len(_IMAGE_CACHE) > IMAGE_CACHE_ENTRIES:
        _IMAGE_CACHE.popitem(last=False)


def drop_images():
    _IMAGE_CACHE.clear()
    _WEIGHT_IMAGES.clear()
    _WEIGHT_NORMS.clear()


# Frozen weights: their images never change, but keeping them costs as many bytes again as the
# fp32 weights (+1.2 GB on the 24-layer BERT-large-dims model, 8 % of its peak).  Default
# (SPT_WEIGHT_IMAGES unset or 'step'): a weight is re-split every time it is used (4 MiB ->
# ~4 us, ~1.5 % of a step) and nothing is kept; SPT_WEIGHT_IMAGES=keep holds the image for the
# life of the parameter (dropped when `_version` moves).
class _PerParameter:
    """{parameter -> value} keyed by identity (a tensor's `==` is elementwise, which rules out
    weakref.WeakKeyDictionary); an entry dies with its parameter or when `_version` moves."""

    def __init__(self):
        self.data = {}

    def get(self, w):
        hit = self.data.get(id(w))
        if hit is not None and hit[0]() is w and hit[1] == (w.data_ptr(), w._version):
            return hit[2]
        return None

    def put(self, w, value):
        key = id(w)
        self.data[key] = (weakref.ref(w, lambda _, k=key: self.data.pop(k, None)),
                          (w.data_ptr(), w._version), value)

    def clear(self):
        self.data.clear()


_WEIGHT_IMAGES = _PerParameter()
KEEP_WEIGHT_IMAGES = os.environ.get('SPT_WEIGHT_IMAGES', 'step') == 'keep'


def row_norms(x: torch.Tensor) -> torch.Tensor:
    """Euclidean norm of every row of a matrix (fp32 [rows])."""
    return torch.linalg.vector_norm(x.detach(), dim=1)


_WEIGHT_NORMS = _PerParameter()


def weight_row_norms(w: torch.Tensor) -> torch.Tensor:
    """`row_norms` of a (frozen) weight, kept with the parameter like its image."""
    if w.is_inference():
        return row_norms(w)
    norms = _WEIGHT_NORMS.get(w)
    if norms is None:
        norms = row_norms(w)
        _WEIGHT_NORMS.put(w, norms)
    return norms


def weight_image(w: torch.Tensor) -> SplitImage:
    if not KEEP_WEIGHT_IMAGES or w.is_inference():
        return split_bf16(w.detach())
    img = _WEIGHT_IMAGES.get(w)
    if img is None:
        img = split_bf16(w.detach())
        _WEIGHT_IMAGES.put(w, img)
    return img


def grouped_gemm_fused(a: torch.Tensor, weight: torch.Tensor, offsets: torch.Tensor,
                       n_groups: int, n: int, k: int, w_group_stride: int, w_ldn: int,
                       w_ldk: int, n_rows: int, gather=None, bias=None, rowscale=None,
                       a2=None, gather2=None, b2=None, b2_group_stride: int = 0,
                       epilogue: int = EPI_PLAIN, activation: int = ACT_RELU,
                       keep_preact: bool = False, h_in=None, s_in=None,
                       a_image: SplitImage = None, w_image: SplitImage = None,
                       a_norm: torch.Tensor = None, w_norm: torch.Tensor = None,
                       relu_queue_entries: int = None, out: torch.Tensor = None,
                       accumulate: bool = False, raw_dots: bool = False,
                       a_segments: tuple = None, b2_segment_stride: int = 0):
    """``spt_grouped_gemm_fused``: the block GEMM of a routed FFN with its LoRA side
    product, rowscale / bias, and the activation (EPI_ACT) or its derivative plus the two
    row dots of the coefficient gradient (EPI_DACT) folded in (include/spt_hip.h).

    Returns ``out`` (EPI_PLAIN), ``(out, preact | None)`` (EPI_ACT) or
    ``(out, dot_main [P], dot_act [P])`` (EPI_DACT).  ``out``: a buffer [n_rows, >= n] to write
    into (EPI_PLAIN; its row stride may exceed n: ``ldo`` of include/spt_hip.h); with
    ``accumulate`` the product is ADDED to what `out` holds.  ``a_segments`` = (seg_k, seg_stride):
    `a` is the first of k / seg_k matrices [rows, seg_k] that lie seg_stride floats apart (the
    caller vouches for the others' memory), contracted one after the other.  ``b2_segment_stride``:
    `b2` [n, 16] is the first of a2.size(1) / 16 such tables lying that many floats apart."""
    for t, name in ((a, 'a'), (weight, 'weight')):
        _check_type(t, torch.float32, name)
    _check_type(offsets, torch.int32, 'offsets')
    _require(a.dim() == 2 and a.stride(1) == 1, 'a must be [rows, k] with unit inner stride')
    _require(weight.is_contiguous(), 'weight must be contiguous')
    dev = _same_device(a, weight, offsets)
    for t, name in ((gather, 'gather'), (gather2, 'gather2')):
        if t is not None:
            _check_type(t, torch.int32, name)
    for t in (gather, bias, rowscale, gather2, b2, h_in, s_in):
        if t is not None:
            _require(t.is_cuda and t.is_contiguous(), 'grouped_gemm_fused: contiguous CUDA operands')
    if a2 is not None:      # [rows, r]: may be a column slice of a wider matrix (lda2 = its row stride)
        _require(a2.is_cuda and a2.dim() == 2 and a2.stride(1) == 1 and a2.stride(0) % 4 == 0
                 and a2.data_ptr() % 16 == 0, 'a2: [rows, r] fp32, unit inner stride, 16-byte aligned rows')
    if epilogue == EPI_ACT and activation == ACT_RELU:
        # pre-activations within the split's error of zero are recomputed in fp32: the kernel
        # finds them from the row norms of both operands (include/spt_hip.h)
        if a_norm is None:
            a_norm = row_norms(a)
        if w_norm is None:          # |W_g(n, :)| for every (g, n): rows, or columns when n runs fastest
            w_norm = row_norms(weight.view(-1, w_ldn)) if w_ldk == 1 else \
                torch.linalg.vector_norm(weight.view(-1, w_ldk), dim=0)
        for t, rows, name in ((a_norm, a.size(0), 'a_norm'), (w_norm, n_groups * n, 'w_norm')):
            _check_type(t, torch.float32, name)
            _require(t.is_cuda and t.is_contiguous() and t.numel() == rows,
                     name + ': one fp32 norm per row')
    else:
        a_norm = w_norm = None
    # operand forms of the k-loop (include/spt_hip.h): both images; or (`w_image` alone) the weight's
    # image and the activation's own fp32 rows, split inside the kernel ("A32"); or neither
    images = w_image is not None
    if a_image is not None:
        _require(w_image is not None, 'a_image: only together with w_image')
        _require(a_image.rows == a.size(0) and a_image.cols == a.size(1) == k
                 and a.stride(0) == a.size(1), 'a_image: the image of the contiguous [*, k] matrix a')
    if images:
        _require(w_image.rows * w_image.cols == weight.numel()
                 and w_image.cols == (w_ldn if w_ldk == 1 else w_ldk),
                 'w_image: the image of the weight with rows of w_ldn (w_ldk) elements')
    lib = load_library()
    with _on(dev):
        _require(not accumulate or out is not None, 'accumulate: needs `out`')
        if out is None:
            out = torch.empty([n_rows, n], dtype=torch.float32, device=dev)
        else:
            _check_type(out, torch.float32, 'out')
            _require(epilogue == EPI_PLAIN and out.is_cuda and out.dim() == 2 and out.stride(1) == 1
                     and out.size(0) == n_rows and out.size(1) >= n and out.stride(0) >= out.size(1),
                     'out: [n_rows, >= n] fp32 with unit inner stride (EPI_PLAIN only)')
        preact = dot_main = dot_act = None
        if epilogue == EPI_ACT and keep_preact:
            preact = torch.empty_like(out)
        width = lib.spt_grouped_gemm_pdot_width(n)
        if epilogue == EPI_DACT:
            dot_main = torch.empty([n_rows, width], dtype=torch.float32, device=dev)
            dot_act = torch.empty([n_rows, width], dtype=torch.float32, device=dev)
        queue = None
        if a_norm is not None and relu_queue_entries != 0:
            # scratch for the near-the-kink queue (default: 25 x the expected fill)
            entries = relu_queue_entries or max(16384, n_rows * n // 64)
            queue = torch.empty([256 * 64 + 8 * entries], dtype=torch.uint8, device=dev)
        if n_rows > 0:
            desc = _GroupedDesc(
                a=_ptr(a), gather=_ptr(gather), w=_ptr(weight), bias=_ptr(bias),
                rowscale=_ptr(rowscale), offsets=_ptr(offsets), out=_ptr(out),
                n_rows=n_rows, k=k, n=n, n_groups=n_groups, lda=a.stride(0),
                w_group_stride=w_group_stride, w_ldn=w_ldn, w_ldk=w_ldk,
                a2=_ptr(a2), gather2=_ptr(gather2), b2=_ptr(b2),
                lda2=a2.stride(0) if a2 is not None else 0,
                r=a2.size(1) if a2 is not None else 0,
                b2_group_stride=b2_group_stride,
                b2_ldn=b2.stride(-2) if b2 is not None else 0,
                epilogue=epilogue, activation=activation, out2=_ptr(preact),
                h_in=_ptr(h_in), s_in=_ptr(s_in), pdot_main=_ptr(dot_main),
                pdot_act=_ptr(dot_act), pdot_ld=width,
                a_image=a_image.buffer.data_ptr() if a_image is not None else None,
                w_image=w_image.buffer.data_ptr() if images else None,
                a_norm=_ptr(a_norm), w_norm=_ptr(w_norm), relu_queue=_ptr(queue),
                relu_queue_bytes=queue.numel() if queue is not None else 0, ldo=out.stride(0),
                accumulate=int(bool(accumulate)),
                a_seg_k=a_segments[0] if a_segments else 0,
                a_seg_stride=a_segments[1] if a_segments else 0,
                b2_seg_stride=b2_segment_stride)
            global LAST_GEMM_USED_IMAGES, LAST_GEMM_PATH
            LAST_GEMM_PATH = ('register', 'image', 'a32')[lib.spt_grouped_gemm_image_path(ctypes.byref(desc))]
            LAST_GEMM_USED_IMAGES = LAST_GEMM_PATH != 'register'
            PATH_COUNTS[('grouped_gemm', LAST_GEMM_PATH)] += 1
            rc = lib.spt_grouped_gemm_fused(ctypes.byref(desc), _stream(dev))
            if rc != 0:
                _raise(lib, rc, 'grouped_gemm_fused')
    if epilogue == EPI_ACT:
        global LAST_RELU_QUEUE
        LAST_RELU_QUEUE = queue
        return out, preact
    if epilogue == EPI_DACT:
        if raw_dots:            # [P, width] partial row dots per column tile (ffn_coeff_grad sums them)
            return out, dot_main, dot_act
        return out, dot_main.sum(dim=-1), dot_act.sum(dim=-1)
    return out


def route_topk_supported(n_tokens: int, n_blocks: int) -> bool:
    return 0 < n_blocks <= 8 and 0 < n_tokens <= 65536


def route_topk(prob: torch.Tensor, k: int):
    """``spt_route_topk``: prob [T, G] fp32 -> (token [P], block [P], offsets [G + 1],
    pos [T, k]) int32, the (token, block) pairs of the per-token top-k sorted by block."""
    _check_dim(prob, 2, 'prob')
    _check_type(prob, torch.float32, 'prob')
    T, G = prob.shape
    _require(0 < k <= G, 'route_topk: 0 < k <= n_blocks')
    dev = prob.device
    lib = load_library()
    with _on(dev):
        token = torch.empty([T * k], dtype=torch.int32, device=dev)
        block = torch.empty([T * k], dtype=torch.int32, device=dev)
        offsets = torch.empty([G + 1], dtype=torch.int32, device=dev)
        pos = torch.empty([T, k], dtype=torch.int32, device=dev)
        rc = lib.spt_route_topk(prob.data_ptr(), token.data_ptr(), block.data_ptr(),
                                offsets.data_ptr(), pos.data_ptr(), T, G, k, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'route_topk')
    return token, block, offsets, pos


def tall_tn_supported(wide: torch.Tensor, narrow: torch.Tensor) -> bool:
    return (wide.is_cuda and wide.dtype == torch.float32 and narrow.dtype == torch.float32
            and wide.dim() == 2 and narrow.dim() == 2 and wide.stride(1) == 1 and narrow.stride(1) == 1
            and narrow.size(1) in (4, 16, 48) and wide.size(1) % 2 == 0 and wide.stride(0) % 2 == 0
            and wide.data_ptr() % 8 == 0 and wide.size(0) > 0
            and narrow.stride(0) % 4 == 0 and narrow.data_ptr() % 16 == 0)


def tall_tn(wide: torch.Tensor, narrow: torch.Tensor, gather: torch.Tensor = None,
            offsets: torch.Tensor = None, transposed: bool = False, split16: bool = False) -> torch.Tensor:
    """``spt_tall_tn``: wide^T . narrow -> [G, width, n] (``transposed``: [G, n, width];
    ``split16``: [G, n / 16, width, 16], one contiguous matrix per rank-16 table); G = 1
    without ``offsets``, else the row groups offsets[g] .. offsets[g + 1] (device int32).
    ``gather`` [rows] int32 picks the row of ``narrow`` for every row of ``wide``."""
    _require(not (transposed and split16), 'tall_tn: transposed or split16')
    _require(tall_tn_supported(wide, narrow), 'tall_tn: fp32 CUDA [rows, even width] x [*, 4 | 16 | 48]')
    rows, width = wide.shape
    n = narrow.size(1)
    G = 1 if offsets is None else offsets.numel() - 1
    if gather is not None:
        _check_type(gather, torch.int32, 'gather')
        _require(gather.is_contiguous() and gather.numel() == rows, 'gather: one int32 per row of wide')
    else:
        _require(narrow.size(0) == rows, 'narrow: one row per row of wide')
    if offsets is not None:
        _check_type(offsets, torch.int32, 'offsets')
        _require(offsets.is_contiguous() and G >= 1, 'offsets: [G + 1] int32')
    dev = wide.device
    lib = load_library()
    with _on(dev):
        shape = [G, n, width] if transposed else [G, n // 16, width, 16] if split16 else [G, width, n]
        out = torch.empty(shape, dtype=torch.float32, device=dev)
        work = torch.empty([lib.spt_tall_tn_workspace_bytes(rows, G, width, n)], dtype=torch.uint8, device=dev)
        rc = lib.spt_tall_tn(wide.data_ptr(), wide.stride(0), narrow.data_ptr(), narrow.stride(0),
                             _ptr(gather), _ptr(offsets), G, rows, width, n, out.data_ptr(),
                             2 if split16 else int(transposed), work.data_ptr(), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'tall_tn')
    return out


def tall_tn_batchable(wides, narrows) -> bool:
    """1 .. 4 (wide, narrow) pairs of one shape and one pair of row strides, each `tall_tn_supported`"""
    if not (1 <= len(wides) <= 4 and len(narrows) == len(wides)):
        return False
    w0, n0 = wides[0], narrows[0]
    return all(tall_tn_supported(w, nr) and w.shape == w0.shape and nr.shape == n0.shape
               and w.stride(0) == w0.stride(0) and nr.stride(0) == n0.stride(0) and w.device == w0.device
               for w, nr in zip(wides, narrows))


def tall_tn_batch(wides, narrows, gather: torch.Tensor = None, offsets: torch.Tensor = None):
    """``spt_tall_tn_batch``: the `tall_tn` results [G, width, n] of 1 .. 4 (wide, narrow) pairs of one
    shape and one pair of row strides (`tall_tn_batchable`), sharing `gather` and `offsets`: one
    pair of launches for all of them."""
    count = len(wides)
    _require(tall_tn_batchable(wides, narrows), 'tall_tn_batch: 1 .. 4 supported pairs of one shape')
    w0, n0 = wides[0], narrows[0]
    rows, width = w0.shape
    n = n0.size(1)
    G = 1 if offsets is None else offsets.numel() - 1
    if gather is not None:
        _check_type(gather, torch.int32, 'gather')
        _require(gather.is_contiguous() and gather.numel() == rows, 'gather: one int32 per row of wide')
    else:
        _require(n0.size(0) == rows, 'narrow: one row per row of wide')
    if offsets is not None:
        _check_type(offsets, torch.int32, 'offsets')
        _require(offsets.is_contiguous() and G >= 1, 'offsets: [G + 1] int32')
    dev = w0.device
    lib = load_library()
    array = ctypes.c_void_p * count
    with _on(dev):
        out = torch.empty([count, G, width, n], dtype=torch.float32, device=dev)
        work = torch.empty([count * lib.spt_tall_tn_workspace_bytes(rows, G, width, n)], dtype=torch.uint8,
                           device=dev)
        rc = lib.spt_tall_tn_batch(count, array(*[w.data_ptr() for w in wides]), w0.stride(0),
                                   array(*[t.data_ptr() for t in narrows]), n0.stride(0),
                                   _ptr(gather), _ptr(offsets), G, rows, width, n,
                                   array(*[out[i].data_ptr() for i in range(count)]), 0,
                                   work.data_ptr(), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'tall_tn_batch')
    return list(out.unbind(0))


def route_topk_coeff(prob: torch.Tensor, k: int, scale: float):
    """``spt_route_topk_coeff``: `route_topk` plus, from the same launch, token / block as int64
    and coeff[p] = scale * prob[token[p], block[p]].
    -> (token, block, offsets, pos, token64, block64, coeff)."""
    _check_dim(prob, 2, 'prob')
    _check_type(prob, torch.float32, 'prob')
    _require(prob.is_cuda and prob.is_contiguous(), 'route_topk_coeff: contiguous CUDA prob')
    T, G = prob.shape
    _require(0 < k <= G, 'route_topk: 0 < k <= n_blocks')
    dev = prob.device
    lib = load_library()
    with _on(dev):
        ints = torch.empty([3 * T * k + G + 1], dtype=torch.int32, device=dev)
        token, block, pos = ints[:T * k], ints[T * k:2 * T * k], ints[2 * T * k:3 * T * k].view(T, k)
        offsets = ints[3 * T * k:]
        longs = torch.empty([2, T * k], dtype=torch.int64, device=dev)
        coeff = torch.empty([T * k], dtype=torch.float32, device=dev)
        rc = lib.spt_route_topk_coeff(prob.data_ptr(), token.data_ptr(), block.data_ptr(),
                                      offsets.data_ptr(), pos.data_ptr(), longs[0].data_ptr(),
                                      longs[1].data_ptr(), coeff.data_ptr(), float(scale), T, G, k,
                                      _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'route_topk_coeff')
    return token, block, offsets, pos, longs[0], longs[1], coeff



def route_topk_logits(logits: torch.Tensor, bias: torch.Tensor, n_blocks: int, k: int, scale: float):
    """``spt_route_topk_logits``: `route_topk_coeff` from the router's LOGITS (x @ W^T without
    bias: [T, >= n_blocks], unit inner stride, 16-byte aligned rows -- a view of a wider matrix);
    the probabilities sigmoid(logits + bias) are formed in the same launch (the top-k ranks logits + bias:
    the ranking of the probabilities wherever two differ; equal probabilities of distinct logits go to
    the larger logit).
    -> (prob [T, n_blocks], token, block, offsets, pos, token64, block64, coeff)."""
    _check_type(logits, torch.float32, 'logits')
    _require(logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1 and logits.size(1) >= n_blocks
             and logits.data_ptr() % 16 == 0 and logits.stride(0) % 4 == 0,
             'route_topk_logits: [T, >= n_blocks] fp32, 16-byte aligned rows')
    if bias is not None:
        _check_type(bias, torch.float32, 'bias')
        _require(bias.is_contiguous() and bias.numel() == n_blocks and bias.device == logits.device,
                 'bias: [n_blocks]')
    T, G = logits.size(0), n_blocks
    _require(0 < k <= G, 'route_topk: 0 < k <= n_blocks')
    dev = logits.device
    lib = load_library()
    with _on(dev):
        prob = torch.empty([T, G], dtype=torch.float32, device=dev)
        token = torch.empty([T * k], dtype=torch.int32, device=dev)
        block = torch.empty([T * k], dtype=torch.int32, device=dev)
        offsets = torch.empty([G + 1], dtype=torch.int32, device=dev)
        pos = torch.empty([T, k], dtype=torch.int32, device=dev)
        token64 = torch.empty([T * k], dtype=torch.int64, device=dev)
        block64 = torch.empty([T * k], dtype=torch.int64, device=dev)
        coeff = torch.empty([T * k], dtype=torch.float32, device=dev)
        rc = lib.spt_route_topk_logits(logits.data_ptr(), logits.stride(0), _ptr(bias), prob.data_ptr(),
                                       token.data_ptr(), block.data_ptr(), offsets.data_ptr(), pos.data_ptr(),
                                       token64.data_ptr(), block64.data_ptr(), coeff.data_ptr(), float(scale),
                                       T, G, k, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'route_topk_logits')
    return prob, token, block, offsets, pos, token64, block64, coeff

def route_coeff_backward(dcoeff: torch.Tensor, pos: torch.Tensor, block: torch.Tensor,
                         scale: float, n_blocks: int, prob: torch.Tensor = None) -> torch.Tensor:
    """``spt_route_coeff_backward``: d prob [T, n_blocks] from d coeff [T * k]; with ``prob`` (the
    router's sigmoid outputs [T, n_blocks]) ``spt_route_logit_backward``: d logit instead."""
    _check_type(dcoeff, torch.float32, 'dcoeff')
    _check_type(pos, torch.int32, 'pos')
    _check_type(block, torch.int32, 'block')
    _require(dcoeff.is_cuda and dcoeff.is_contiguous() and pos.is_contiguous() and block.is_contiguous()
             and pos.dim() == 2 and dcoeff.numel() == pos.numel() == block.numel(),
             'route_coeff_backward: dcoeff [T * k], pos [T, k], block [T * k], contiguous')
    T, k = pos.shape
    dev = _same_device(dcoeff, pos, block)
    lib = load_library()
    with _on(dev):
        dprob = torch.empty([T, n_blocks], dtype=torch.float32, device=dev)
        if prob is not None:
            _check_type(prob, torch.float32, 'prob')
            _require(prob.is_contiguous() and prob.shape == (T, n_blocks) and prob.device == dev,
                     'prob: contiguous [T, n_blocks]')
            rc = lib.spt_route_logit_backward(dcoeff.data_ptr(), pos.data_ptr(), block.data_ptr(),
                                              float(scale), prob.data_ptr(), dprob.data_ptr(), T,
                                              n_blocks, k, _stream(dev))
        else:
            rc = lib.spt_route_coeff_backward(dcoeff.data_ptr(), pos.data_ptr(), block.data_ptr(),
                                              float(scale), dprob.data_ptr(), T, n_blocks, k, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'route_coeff_backward')
    return dprob


def rotary_supported(x: torch.Tensor, cos: torch.Tensor, sin: torch.Tensor) -> bool:
    return (x.is_cuda and x.dtype == cos.dtype == sin.dtype == torch.float32 and x.dim() == 4
            and x.is_contiguous() and x.size(-1) % 8 == 0 and x.numel() > 0 and x.data_ptr() % 16 == 0
            and cos.is_contiguous() and sin.is_contiguous() and cos.shape == sin.shape and cos.dim() == 2
            and cos.size(0) >= x.size(1) and cos.size(1) == x.size(-1) and cos.device == x.device
            and cos.data_ptr() % 16 == 0 and sin.data_ptr() % 16 == 0)


def rotary(parts, n_rot: int, cos: torch.Tensor, sin: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """``spt_rotary``: parts = 1 .. 3 tensors [N, S, H, E] of one shape; the first `n_rot` get the
    rotary embedding of their positions (tables [>= S, E]; ``transpose``: its adjoint), the others
    are copied -> ONE tensor [len(parts), N, S, H, E]."""
    x0 = parts[0]
    _require(1 <= len(parts) <= 3 and 0 <= n_rot <= len(parts)
             and all(rotary_supported(p, cos, sin) and p.shape == x0.shape for p in parts),
             'rotary: 1 .. 3 contiguous fp32 CUDA tensors [N, S, H, E % 8 == 0] of one shape')
    N, S, H, E = x0.shape
    dev = x0.device
    lib = load_library()
    array = ctypes.c_void_p * len(parts)
    with _on(dev):
        out = torch.empty([len(parts), N, S, H, E], dtype=torch.float32, device=dev)
        rc = lib.spt_rotary(array(*[p.data_ptr() for p in parts]), len(parts), n_rot, out.data_ptr(),
                            x0.numel(), cos.data_ptr(), sin.data_ptr(), N, S, H, E, int(bool(transpose)),
                            _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'rotary')
    return out


def swiglu_supported(g: torch.Tensor, s: torch.Tensor) -> bool:
    return (g.is_cuda and g.dtype == s.dtype == torch.float32 and g.shape == s.shape and g.dim() == 2
            and g.is_contiguous() and s.is_contiguous() and g.size(1) % 4 == 0 and g.numel() > 0
            and g.data_ptr() % 16 == 0 and s.data_ptr() % 16 == 0)


def swiglu_forward(g: torch.Tensor, s: torch.Tensor) -> torch.Tensor:
    """``spt_swiglu_forward``: silu(g) * s."""
    _require(swiglu_supported(g, s), 'swiglu: contiguous fp32 CUDA [rows, n % 4 == 0] pair')
    dev = g.device
    lib = load_library()
    with _on(dev):
        h = torch.empty_like(g)
        rc = lib.spt_swiglu_forward(g.data_ptr(), s.data_ptr(), h.data_ptr(), g.numel(), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'swiglu_forward')
    return h


def swiglu_backward(dh: torch.Tensor, g: torch.Tensor, s: torch.Tensor):
    """``spt_swiglu_backward``: -> (dg, ds, <dh, h> [rows], <dg, g> [rows], <ds, s> [rows])."""
    _require(swiglu_supported(g, s) and dh.shape == g.shape and dh.dtype == torch.float32
             and dh.is_contiguous() and dh.device == g.device and dh.data_ptr() % 16 == 0,
             'swiglu: contiguous fp32 CUDA [rows, n % 4 == 0] tensors')
    dev = g.device
    rows, n = g.shape
    lib = load_library()
    with _on(dev):
        dg, ds = torch.empty_like(g), torch.empty_like(g)
        dots = torch.empty([3, rows], dtype=torch.float32, device=dev)
        rc = lib.spt_swiglu_backward(dh.data_ptr(), g.data_ptr(), s.data_ptr(), dg.data_ptr(), ds.data_ptr(),
                                     dots.data_ptr(), rows, n, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'swiglu_backward')
    return dg, ds, dots[0], dots[1], dots[2]


def embedding_rows_backward(grad: torch.Tensor, ids: torch.Tensor, n_rows: int) -> torch.Tensor:
    """Gradient of ``table[ids]`` with respect to the table (``spt_embedding_rows_backward``):
    grad ``[T, width]`` fp32, ids ``[T]`` int64 -> ``[n_rows, width]``; deterministic, shape-static."""
    _check_dim(grad, 2, 'grad')
    _check_type(grad, torch.float32, 'grad')
    _check_type(ids, torch.int64, 'ids')
    T, width = grad.shape
    _require(ids.numel() == T and grad.stride(1) == 1, 'grad [T, width] with unit column stride, ids [T]')
    _require(width % 4 == 0 and grad.stride(0) % 4 == 0, 'width and row stride: multiples of 4')
    dev = _same_device(grad, ids)
    lib = load_library()
    with _on(dev):
        sid, order = torch.sort(ids.reshape(-1), stable=True)
        out = torch.empty([n_rows, width], dtype=torch.float32, device=dev)
        ws = torch.empty([lib.spt_embedding_rows_backward_workspace_bytes(T, width)], dtype=torch.uint8,
                         device=dev)
        rc = lib.spt_embedding_rows_backward(grad.data_ptr(), grad.stride(0), sid.data_ptr(), order.data_ptr(),
                                             out.data_ptr(), width, ws.data_ptr(), T, width, int(n_rows),
                                             _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'embedding_rows_backward')
    return out


def rows_combine(rows: torch.Tensor, pos: torch.Tensor, bias: torch.Tensor = None,
                 side: torch.Tensor = None, side_weight: torch.Tensor = None) -> torch.Tensor:
    """out[t] = bias + sum_j rows[pos[t, j]] (``spt_rows_combine``); pos [T, k] int32.  With ``side``
    [T, ns] and ``side_weight`` [ns, d] (``spt_rows_combine_side``) side @ side_weight is added on top."""
    _check_type(rows, torch.float32, 'rows')
    _check_type(pos, torch.int32, 'pos')
    _require(rows.dim() == 2 and rows.is_contiguous() and pos.dim() == 2 and pos.is_contiguous(),
             'rows [P, d], pos [T, k] contiguous')
    dev = _same_device(rows, pos)
    T, k = pos.shape
    d = rows.size(1)
    if side is not None:
        _require(side_weight is not None and side.dtype == side_weight.dtype == torch.float32
                 and side.is_contiguous() and side_weight.is_contiguous() and side.dim() == 2
                 and side.size(0) == T and side_weight.shape == (side.size(1), d)
                 and side.device == side_weight.device == dev and side_weight.data_ptr() % 16 == 0,
                 'rows_combine: side [T, ns], side_weight [ns, d], contiguous fp32')
    lib = load_library()
    with _on(dev):
        out = torch.empty([T, d], dtype=torch.float32, device=dev)
        if T > 0:
            if side is None:
                rc = lib.spt_rows_combine(rows.data_ptr(), pos.data_ptr(), _ptr(bias), out.data_ptr(),
                                          T, k, d, _stream(dev))
            else:
                rc = lib.spt_rows_combine_side(rows.data_ptr(), pos.data_ptr(), _ptr(bias), side.data_ptr(),
                                               side_weight.data_ptr(), side.size(1), out.data_ptr(),
                                               T, k, d, _stream(dev))
            if rc != 0:
                _raise(lib, rc, 'rows_combine')
    return out


def lora_down_supported(x: torch.Tensor, table: torch.Tensor) -> bool:
    """Shapes ``spt_lora_down`` takes: a CUDA fp32 matrix with unit inner stride and 16-byte
    aligned rows, K a multiple of 32, 16 / 32 / 48 / 64 table columns."""
    return (x.is_cuda and x.dtype == torch.float32 and table.dtype == torch.float32
            and x.dim() == 2 and table.dim() == 2 and x.size(0) > 0
            and x.stride(1) == 1 and x.stride(0) % 4 == 0 and x.data_ptr() % 16 == 0
            and x.size(1) == table.size(0) and x.size(1) % 32 == 0
            and table.size(1) % 16 == 0 and 0 < table.size(1) <= 64)


def lora_down(x: torch.Tensor, table: torch.Tensor, want_image: bool = False,
              want_norms: bool = False, block_major: bool = False, out: torch.Tensor = None,
              exact: bool = False, table2: torch.Tensor = None):
    """u = x @ table for a tall x [rows, K] and a table [K, n] of a few columns, as ONE pass over x
    (``spt_lora_down``); the same pass can also leave x's split image (:class:`SplitImage`) and its
    row 2-norms.  Returns u, or (u, image | None, norms | None) when a by-product is asked for.
    ``block_major``: u as [n / 16, rows, 16] (tables of several rank-16 adapters side by side:
    each adapter's product contiguous).  ``out``: a [rows, n] view to write u into (unit inner
    stride, any row stride: a column slice of a wider matrix).  ``exact``: u in exact fp32 instead of
    split-bf16 products (for the u in front of a ReLU GEMM: include/spt_hip.h).  ``table2``
    (``spt_lora_down2``; needs ``block_major`` and ``exact``): a row-major [n2 <= 16, K] matrix (an nn.Linear
    weight) whose product x @ table2.T fills one more block: u[-1][:, :n2]."""
    _require(lora_down_supported(x, table), 'lora_down: see lora_down_supported')
    table = table.contiguous()
    rows, k = x.shape
    n = table.size(1)
    dev = _same_device(x, table)
    lib = load_library()
    if table2 is not None:
        _check_type(table2, torch.float32, 'table2')
        _require(block_major and out is None and table2.dim() == 2 and table2.is_contiguous()
                 and table2.size(1) == k and 0 < table2.size(0) <= 16 and n + 16 <= 64
                 and table2.device == dev and table2.data_ptr() % 16 == 0,
                 'lora_down: table2 [n2 <= 16, K] contiguous, block_major output')
        with _on(dev):
            u = torch.empty([n // 16 + 1, rows, 16], dtype=torch.float32, device=dev)
            image = norms = None
            if want_image:
                image = SplitImage(torch.empty([lib.spt_split_bf16_bytes(rows, k)], dtype=torch.uint8,
                                               device=dev), rows, k)
            if want_norms:
                norms = torch.empty([rows], dtype=torch.float32, device=dev)
            rc = lib.spt_lora_down2(x.data_ptr(), x.stride(0), rows, k, table.data_ptr(), n,
                                    table2.data_ptr(), table2.size(0), u.data_ptr(), 0, 1,
                                    image.buffer.data_ptr() if want_image else None, _ptr(norms),
                                    int(bool(exact)), _stream(dev))
        if rc != 0:
            _raise(lib, rc, 'lora_down2')
        return (u, image, norms) if (want_image or want_norms) else u
    with _on(dev):
        if out is not None:
            _check_type(out, torch.float32, 'out')
            _require(not block_major and out.is_cuda and out.shape == (rows, n) and out.stride(1) == 1
                     and out.stride(0) >= n, 'lora_down: out [rows, n] with unit inner stride')
            u = out
        else:
            u = torch.empty([n // 16, rows, 16] if block_major else [rows, n], dtype=torch.float32, device=dev)
        image = norms = None
        if want_image:
            image = SplitImage(torch.empty([lib.spt_split_bf16_bytes(rows, k)], dtype=torch.uint8,
                                           device=dev), rows, k)
        if want_norms:
            norms = torch.empty([rows], dtype=torch.float32, device=dev)
        rc = lib.spt_lora_down(x.data_ptr(), x.stride(0), rows, k, table.data_ptr(), n, u.data_ptr(),
                               0 if block_major else u.stride(0),
                               int(bool(block_major)), image.buffer.data_ptr() if want_image else None,
                               _ptr(norms), int(bool(exact)), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'lora_down')
    return (u, image, norms) if (want_image or want_norms) else u


def spacing_of(tensors) -> int:
    """floats from each tensor to the next when they are contiguous fp32 tensors of one shape, disjoint
    and equally spaced inside one allocation (e.g. parameters of a flat buffer), else 0"""
    a = tensors[0]
    if not all(t.is_contiguous() and t.dtype == torch.float32 and t.shape == a.shape
               and t.untyped_storage().data_ptr() == a.untyped_storage().data_ptr() for t in tensors):
        return 0
    if len(tensors) < 2:
        return 0
    step = tensors[1].data_ptr() - a.data_ptr()
    if step < a.numel() * 4 or step % 16 != 0:
        return 0
    if any(t.data_ptr() - s.data_ptr() != step for s, t in zip(tensors[:-1], tensors[1:])):
        return 0
    return step // 4


def lora_down_tables(x: torch.Tensor, tables, want_image: bool = False):
    """``spt_lora_down_tables``: x @ [t_0 | t_1 | ..] for separate equally spaced tables [K, 16]
    (``spacing_of(tables) != 0``) -> (u [len(tables), rows, 16], image | None)."""
    step = spacing_of(tables)
    _require(step != 0 and lora_down_supported(x, tables[0]) and tables[0].size(1) == 16 and len(tables) <= 4,
             'lora_down_tables: 1 .. 4 equally spaced [K, 16] tables')
    rows, k = x.shape
    dev = _same_device(x, tables[0])
    lib = load_library()
    with _on(dev):
        u = torch.empty([len(tables), rows, 16], dtype=torch.float32, device=dev)
        image = None
        if want_image:
            image = SplitImage(torch.empty([lib.spt_split_bf16_bytes(rows, k)], dtype=torch.uint8,
                                           device=dev), rows, k)
        rc = lib.spt_lora_down_tables(x.data_ptr(), x.stride(0), rows, k, tables[0].data_ptr(), len(tables),
                                      step, u.data_ptr(), image.buffer.data_ptr() if want_image else None,
                                      None, 0, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'lora_down_tables')
    return u, image


def lora_down_stacked(xs, tables, offsets: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
    """``spt_lora_down_grouped_cols``: out[:, 16 g : 16 g + 16] = xs[g] @ tables[g] for G matrices
    xs[g] [rows, K] lying back to back (``spacing_of(xs) == rows * K``) and G equally spaced tables
    [K, 16], in one launch; offsets = [0, rows, 2 rows, ..] int32 on the device; out [rows, >= 16 G]."""
    G = len(xs)
    rows, k = xs[0].shape
    step = spacing_of(tables) if G > 1 else tables[0].numel()
    _require(G >= 1 and len(tables) == G and (G == 1 or spacing_of(xs) == rows * k) and step != 0
             and tables[0].shape == (k, 16) and lora_down_supported(xs[0], tables[0]),
             'lora_down_stacked: stacked inputs, equally spaced [K, 16] tables')
    _check_type(offsets, torch.int32, 'offsets')
    _check_type(out, torch.float32, 'out')
    _require(offsets.numel() == G + 1 and out.is_cuda and out.dim() == 2 and out.size(0) == rows
             and out.stride(1) == 1 and out.size(1) >= 16 * G and out.stride(0) % 4 == 0,
             'lora_down_stacked: offsets [G + 1], out [rows, >= 16 G]')
    dev = _same_device(xs[0], tables[0], offsets, out)
    lib = load_library()
    with _on(dev):
        rc = lib.spt_lora_down_grouped_cols(xs[0].data_ptr(), xs[0].stride(0), G * rows, k, tables[0].data_ptr(),
                                            step, 16, offsets.data_ptr(), G, out.data_ptr(), out.stride(0),
                                            _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'lora_down_stacked')
    return out


def lora_down_grouped_supported(x: torch.Tensor, tables: torch.Tensor) -> bool:
    """x [rows, K] as for `lora_down`; tables [G, K, n] contiguous, n in 16 .. 64, G <= 64."""
    return (tables.dim() == 3 and tables.is_contiguous() and 0 < tables.size(0) <= 64
            and lora_down_supported(x, tables[0]))


def lora_down_grouped(x: torch.Tensor, tables: torch.Tensor, offsets: torch.Tensor,
                      want_image: bool = False):
    """``spt_lora_down_grouped``: u[p] = x[p] @ tables[g(p)] for rows sorted by group (offsets
    [G + 1] int32 on the device, offsets[G] == rows).  Returns u [rows, n], or (u, image) with
    ``want_image``."""
    _require(lora_down_grouped_supported(x, tables), 'lora_down_grouped: see lora_down_grouped_supported')
    _check_type(offsets, torch.int32, 'offsets')
    G, k, n = tables.shape
    _require(offsets.is_contiguous() and offsets.numel() == G + 1, 'offsets: [G + 1] int32')
    rows = x.size(0)
    dev = _same_device(x, tables, offsets)
    lib = load_library()
    with _on(dev):
        u = torch.empty([rows, n], dtype=torch.float32, device=dev)
        image = None
        if want_image:
            image = SplitImage(torch.empty([lib.spt_split_bf16_bytes(rows, k)], dtype=torch.uint8,
                                           device=dev), rows, k)
        rc = lib.spt_lora_down_grouped(x.data_ptr(), x.stride(0), rows, k, tables.data_ptr(), k * n, n,
                                       offsets.data_ptr(), G, u.data_ptr(), 0,
                                       image.buffer.data_ptr() if want_image else None, None,
                                       _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'lora_down_grouped')
    return (u, image) if want_image else u


def cross_entropy_grad_(logits: torch.Tensor, n_classes: int, target: torch.Tensor,
                        scale: torch.Tensor, ignore_index: int = -100) -> torch.Tensor:
    """``spt_cross_entropy_grad``: per-row losses [rows] of softmax cross-entropy, and `logits`
    [rows, ld >= n_classes] OVERWRITTEN with (softmax - onehot) * scale (pad columns: zeros).
    `scale`: a one-element device tensor (1 / number of counted targets for the mean)."""
    _check_type(logits, torch.float32, 'logits')
    _check_type(target, torch.int64, 'target')
    _check_type(scale, torch.float32, 'scale')
    _require(logits.is_cuda and logits.dim() == 2 and logits.stride(1) == 1
             and logits.stride(0) % 4 == 0 and logits.stride(0) >= logits.size(1) >= n_classes
             and logits.data_ptr() % 16 == 0, 'logits: [rows, ld >= n_classes], 16-byte aligned rows')
    _require(target.is_contiguous() and target.numel() == logits.size(0) and scale.numel() == 1,
             'target: one int64 per row; scale: one float')
    dev = _same_device(logits, target, scale)
    rows = logits.size(0)
    lib = load_library()
    with _on(dev):
        loss = torch.empty([rows], dtype=torch.float32, device=dev)
        if rows > 0:
            rc = lib.spt_cross_entropy_grad(logits.data_ptr(), logits.stride(0), rows, n_classes,
                                            target.data_ptr(), scale.data_ptr(), loss.data_ptr(),
                                            ignore_index, _stream(dev))
            if rc != 0:
                _raise(lib, rc, 'cross_entropy_grad')
    return loss


LAYERNORM_WIDTHS = (256, 512, 1024, 2048)


def layernorm_supported(x: torch.Tensor, d: int, rms: bool = False) -> bool:
    return (x.is_cuda and x.dtype == torch.float32 and x.size(-1) == d and x.numel() > 0
            and (d in LAYERNORM_WIDTHS or (rms and d == 4096)))


def add_layernorm_forward(x: torch.Tensor, r, gamma: torch.Tensor, beta, eps: float, rms: bool = False):
    """``spt_add_layernorm_forward``: (s, y, mean, rstd) with s = x + r (s is x itself when r is
    None) and y = LayerNorm(s) -- or, ``rms``, RMSNorm(s) (beta None); x, r [..., d] contiguous."""
    d = x.size(-1)
    _require(layernorm_supported(x, d, rms) and x.is_contiguous() and gamma.is_contiguous()
             and gamma.numel() == d and gamma.dtype == torch.float32,
             'add_layernorm: contiguous CUDA fp32 [..., d], d in {256, 512, 1024, 2048} (RMS: also 4096)')
    if not rms:
        _require(beta is not None and beta.is_contiguous() and beta.numel() == d, 'beta: [d]')
    if r is not None:
        _require(r.shape == x.shape and r.is_contiguous() and r.dtype == torch.float32, 'r: like x')
    rows = x.numel() // d
    dev = _same_device(x, gamma)
    lib = load_library()
    with _on(dev):
        s = torch.empty_like(x) if r is not None else x
        y = torch.empty_like(x)
        mean = torch.empty([rows], dtype=torch.float32, device=dev)
        rstd = torch.empty([rows], dtype=torch.float32, device=dev)
        rc = lib.spt_add_layernorm_forward(x.data_ptr(), _ptr(r), gamma.data_ptr(),
                                           None if rms else beta.data_ptr(),
                                           s.data_ptr() if r is not None else None, y.data_ptr(),
                                           mean.data_ptr(), rstd.data_ptr(), rows, d, float(eps),
                                           int(bool(rms)), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'add_layernorm_forward')
    return s, y, mean, rstd


def layernorm_backward(s: torch.Tensor, dy: torch.Tensor, gamma: torch.Tensor, mean: torch.Tensor,
                       rstd: torch.Tensor, dskip=None, rms: bool = False):
    """``spt_layernorm_backward``: (dx, dgamma, dbeta); dx includes `dskip` when given (``rms``:
    dbeta is zeros)."""
    d = s.size(-1)
    _require(layernorm_supported(s, d, rms) and s.is_contiguous() and dy.is_contiguous()
             and dy.shape == s.shape and dy.dtype == torch.float32, 'layernorm_backward: s, dy [..., d]')
    if dskip is not None:
        _require(dskip.shape == s.shape and dskip.is_contiguous() and dskip.dtype == torch.float32,
                 'dskip: like s')
    rows = s.numel() // d
    dev = _same_device(s, dy, gamma, mean, rstd)
    lib = load_library()
    with _on(dev):
        dx = torch.empty_like(s)
        dparam = torch.empty([2, d], dtype=torch.float32, device=dev)
        partial = torch.empty([lib.spt_layernorm_partial_rows(rows), 2 * d], dtype=torch.float32, device=dev)
        rc = lib.spt_layernorm_backward(s.data_ptr(), dy.data_ptr(), gamma.data_ptr(), mean.data_ptr(),
                                        rstd.data_ptr(), _ptr(dskip), dx.data_ptr(), dparam.data_ptr(),
                                        dparam[1].data_ptr(), partial.data_ptr(), rows, d,
                                        int(bool(rms)), _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'layernorm_backward')
    return dx, dparam[0], dparam[1]


def ffn_coeff_grad(dot_main: torch.Tensor, dot_act: torch.Tensor, du: torch.Tensor, u: torch.Tensor,
                   dzt: torch.Tensor, z: torch.Tensor, token: torch.Tensor, coeff: torch.Tensor,
                   floor_value: float) -> torch.Tensor:
    """``spt_ffn_coeff_grad``: (sum dot_main + sum dot_act - <du, u[token]> - <dzt[token], z>) /
    max(coeff, floor) per row, one launch (the routed FFN's router-coefficient gradient)."""
    for t in (dot_main, dot_act, du, u, dzt, z, coeff):
        _check_type(t, torch.float32, 'ffn_coeff_grad operand')
        _require(t.is_cuda and t.is_contiguous(), 'ffn_coeff_grad: contiguous CUDA operands')
    _check_type(token, torch.int32, 'token')
    P, r = du.shape
    _require(dot_main.shape == dot_act.shape and dot_main.size(0) == P and z.shape == (P, r)
             and u.size(1) == r and dzt.shape == u.shape and token.numel() == P and coeff.numel() == P
             and r % 4 == 0, 'ffn_coeff_grad: shapes')
    dev = _same_device(dot_main, du, u, token, coeff)
    lib = load_library()
    with _on(dev):
        out = torch.empty([P], dtype=torch.float32, device=dev)
        rc = lib.spt_ffn_coeff_grad(dot_main.data_ptr(), dot_act.data_ptr(), dot_main.size(1),
                                    du.data_ptr(), u.data_ptr(), dzt.data_ptr(), z.data_ptr(),
                                    token.data_ptr(), coeff.data_ptr(), float(floor_value),
                                    out.data_ptr(), P, r, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'ffn_coeff_grad')
    return out


def softmax_backward_clamped(indptr: torch.Tensor, indices: torch.Tensor,
                             output: torch.Tensor, grad_output: torch.Tensor,
                             clamped_scores: torch.Tensor, scale: float,
                             clamp: float) -> torch.Tensor:
    """softmax backward chained through ``clamp(scale * raw, -clamp, clamp)``: returns the
    gradient wrt the raw sddmm output (``spt_softmax_backward_clamped``)."""
    _check_csr(indptr, indices)
    for t, name in ((output, 'output'), (grad_output, 'grad_output'),
                    (clamped_scores, 'clamped_scores')):
        _check_dim(t, 2, name)
        _check_type(t, torch.float32, name)
        _require(t.shape == indices.shape, 'indices.sizes() == {}.sizes()'.format(name))
    dev = _same_device(indptr, indices, output, grad_output, clamped_scores)
    B, nnz = indices.shape
    S = indptr.size(-1) - 1
    lib = load_library()
    with _on(dev):
        grad = torch.empty_like(output)
        rc = lib.spt_softmax_backward_clamped(
            indptr.data_ptr(), indices.data_ptr(), output.data_ptr(),
            grad_output.data_ptr(), clamped_scores.data_ptr(), float(scale), float(clamp),
            grad.data_ptr(), B, S, nnz, _stream(dev))
    if rc != 0:
        _raise(lib, rc, 'softmax_backward_clamped')
    return grad
