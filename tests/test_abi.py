"""The C-ABI library loads and exports every symbol include/spt_hip.h declares.
No compute call is made here (no GPU in the CPU test tier)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, 'include', 'spt_hip.h')
LIB = os.path.join(ROOT, 'spt-proto_amd', 'lib', 'libspt_hip.so')


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(spt_[a-z_0-9]+)\s*\(', text)))


def test_header_declares_the_seven_operators():
    names = declared_symbols()
    for op in ['cdist_forward', 'cdist_backward', 'lookup_forward',
               'sddmm_forward', 'spmm_forward', 'softmax_forward',
               'softmax_backward']:
        assert 'spt_' + op in names


def test_library_exports_every_declared_symbol():
    if not os.path.exists(LIB):
        pytest.fail('libspt_hip.so not built: run __graft_entry__.build()')
    lib = ctypes.CDLL(LIB)
    for name in declared_symbols():
        assert hasattr(lib, name), name
    lib.spt_abi_version.restype = ctypes.c_int
    header_version = int(re.search(r'#define SPT_ABI_VERSION (\d+)', open(HEADER).read()).group(1))
    assert lib.spt_abi_version() == header_version
    lib.spt_strerror.restype = ctypes.c_char_p
    lib.spt_strerror.argtypes = [ctypes.c_int]
    assert lib.spt_strerror(0) == b'ok'
    assert b'shape' in lib.spt_strerror(-2)


def test_python_binding_matches_header():
    from naive_gpt import ext
    assert sorted(ext._PROTOTYPES) == declared_symbols()
    ext.load_library()


def test_precondition_codes_without_gpu():
    """Argument validation happens on the host before any launch."""
    from naive_gpt import ext
    lib = ext.load_library()
    # null pointers -> SPT_EINVAL, checked before anything touches the device
    assert lib.spt_sddmm_forward(None, None, None, None, None, 1, 16, 16, 32,
                                 1.0, 0.0, 0, 0, None) == -1
    assert lib.spt_lookup_forward(None, None, None, 1, 16, 8, 8, None) == -1
    assert lib.spt_cdist_backward_workspace_bytes(8, 1024, 16, 8) > 0


def test_product_path_has_no_cpu_fallback():
    import torch
    from naive_gpt import ext
    q = torch.zeros([1, 16, 8])
    t = torch.zeros([1, 16, 8])
    with pytest.raises(RuntimeError, match='CUDA tensor'):
        ext.cdist_forward_cuda(q, t)
    # the product package never imports the oracle
    pkg = os.path.join(ROOT, 'spt-proto_amd')
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith('.py'):
                src = open(os.path.join(dirpath, f)).read()
                assert 'oracle' not in src.replace('oracle/', ''), f
