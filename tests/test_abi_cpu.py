"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol declared in
include/semseg_hip.h (no compute calls without a GPU), and the product refuses to run without a HIP device."""
import ctypes
import os
import pytest
import torch

from semseg_amd import _lib


def test_header_parses_and_lists_entry_points():
    protos = _lib.parse_header()
    for name in ('ss_gemm_nt', 'ss_gemm_tn', 'ss_gemm_tn_colsum', 'ss_layernorm_fwd', 'ss_mit_attn_fwd', 'ss_mit_attn_bwd', 'ss_dwconv3_fwd',
                 'ss_bn_stats', 'ss_bilinear_fwd', 'ss_last_error', 'ss_version'):
        assert name in protos
    assert protos['ss_version'][1] == []
    assert protos['ss_gemm_nt'][1][0] is ctypes.c_int and protos['ss_gemm_nt'][1][2] is ctypes.c_void_p


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), 'libsemseg_hip.so not built (run __graft_entry__.build())'
    handle = _lib.lib()                     # raises on a missing symbol
    assert handle.ss_version() >= 1
    for name in _lib.parse_header():
        assert hasattr(handle, name)


def test_no_cpu_fallback():
    from semseg_amd import ops
    x = torch.zeros(8, 8)
    with pytest.raises(RuntimeError):
        ops.add(x, x)
    with pytest.raises(RuntimeError):
        ops.gemm_nt(x, x)


def test_argument_validation_reports_through_last_error():
    handle = _lib.lib()
    rc = handle.ss_gemm_nt(0, 0, None, 0, None, 0, None, 0, 1, 1, 1, None, None, 0, None, 0, 0, None)
    assert rc != 0 and b'ss_gemm_nt' in handle.ss_last_error()
    rc = handle.ss_layernorm_fwd(0, 1, 1, 1, 1, 1, 1, 10, 12, 1e-5, None)   # C not a multiple of 8: rejected before any launch
    assert rc != 0 and b'multiple of 8' in handle.ss_last_error()
