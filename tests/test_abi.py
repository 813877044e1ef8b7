"""CPU: the C-ABI library loads and exports exactly the symbols include/rspnet_hip.h declares (no compute calls)."""
import ctypes
import os
import re

from rspnet_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    text = open(os.path.join(ROOT, "include", "rspnet_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rsp_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree():
    assert header_functions() == sorted(_lib.SIGNATURES)


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    for name in header_functions():
        assert getattr(lib, name) is not None, name
    assert lib.rsp_version() >= 100
    assert lib.rsp_strerror(-2) == b"workspace too small"


def test_geometry_queries_without_gpu():
    lib = _lib.load()
    d = _lib.ConvDesc(32, 16, 56, 56, 64, 16, 56, 56, 128, 3, 3, 3, 1, 1, 1, 1, 1, 1, 64, 128)
    assert lib.rsp_conv3d_packed_fwd_elems(ctypes.byref(d)) == 128 * 27 * 64
    assert lib.rsp_conv3d_stat_tiles(ctypes.byref(d)) == 32 * 16 * 56 * 56 // 128
    bad = _lib.ConvDesc(32, 16, 56, 56, 64, 15, 56, 56, 128, 3, 3, 3, 1, 1, 1, 1, 1, 1, 64, 128)   # wrong Do
    assert lib.rsp_conv3d_packed_fwd_elems(ctypes.byref(bad)) == 0
    # invalid descriptor / null pointers are rejected before any launch
    assert lib.rsp_conv3d_fwd(ctypes.byref(bad), None, None, None, None, None, None, 0, None) == -1
    assert b"descriptor" in lib.rsp_last_error()


def test_ops_fail_loudly_without_gpu_tensors():
    import pytest
    import torch
    from rspnet_amd import ops
    be = ops.HipOps()
    with pytest.raises(_lib.RspError):
        be.momentum_update(torch.zeros(8), torch.zeros(8), 0.9)


def test_missing_library_is_a_hard_error(monkeypatch):
    """No CPU / eager fallback: without the built .so every op backend construction raises."""
    import pytest
    from rspnet_amd import ops
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/librspnet_hip.so")
    with pytest.raises(_lib.RspError):
        _lib.load()
    with pytest.raises(_lib.RspError):
        ops.HipOps()
