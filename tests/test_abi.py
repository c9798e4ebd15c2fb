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


def test_layout_predicates_of_the_in_place_paths():
    """`ext.back_to_back` / `ext.spacing_of` decide (on the host, from addresses only) whether q and k,
    dQ / dK / dV or the adapter tables of a flat parameter buffer can be read in place."""
    import torch
    from naive_gpt import ext
    pair = torch.zeros([2, 3, 4, 5, 8])
    assert ext.back_to_back(pair[0], pair[1])
    assert not ext.back_to_back(pair[1], pair[0])                    # order matters
    assert not ext.back_to_back(pair[0], pair[0].clone())            # another allocation
    assert not ext.back_to_back(pair[0][:, :2], pair[0][:, 2:])      # not contiguous
    flat = torch.zeros([6 * 64 + 8])
    tables = [flat[i * 128:i * 128 + 64].view(4, 16) for i in range(3)]      # 64 floats, 128 apart
    assert ext.spacing_of(tables) == 128
    assert ext.spacing_of(tables[:2]) == 128 and ext.spacing_of(tables[:1]) == 0
    assert ext.spacing_of([tables[0], tables[2], tables[1]]) == 0    # not ascending / unequal
    assert ext.spacing_of([tables[0], tables[1], flat[258:322].view(4, 16)]) == 0
    assert ext.spacing_of([t.clone() for t in tables]) == 0          # separate allocations
    assert ext.spacing_of([flat[0:64].view(4, 16), flat[66:130].view(4, 16)]) == 0   # 8-byte step: not 16-aligned


def test_new_entries_validate_their_arguments_without_gpu():
    from naive_gpt import ext
    lib = ext.load_library()
    assert lib.spt_lora_down2(None, 0, 0, 0, None, 0, None, 0, None, 0, 0, None, None, 1, None) == -1
    assert lib.spt_tall_tn_batch(0, None, 0, None, 0, None, None, 1, 0, 0, 0, None, 0, None, None) == -1
    assert lib.spt_route_topk_logits(None, 0, None, None, None, None, None, None, None, None, None, 1.0,
                                     0, 0, 0, None) == -1
    assert lib.spt_pq_loss_backward_parts(None, None, None, None, None, None, 0, 0, 0, 0, 0, 0, None) == -1
    assert lib.spt_rows_combine_side(None, None, None, None, None, 0, None, 0, 0, 0, None) == -1
    assert lib.spt_lora_down_tables(None, 0, 0, 0, None, 0, 0, None, None, None, 0, None) == -1
    assert lib.spt_lora_down_grouped_cols(None, 0, 0, 0, None, 0, 0, None, 0, None, 0, None) == -1
