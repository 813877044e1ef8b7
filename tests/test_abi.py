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


def test_constant_division_matches_integer_division():
    """The conv kernels decode rows / k positions with multiply-high by a host-computed magic number (conv_igemm.hip FastDiv);
    the same arithmetic evaluated on the host must equal n // d for every divisor a descriptor can produce, up to 2^31 - 1."""
    import numpy as np
    lib = _lib.load()
    rng = np.random.default_rng(7)
    divisors = sorted(set(list(range(1, 130)) + [143, 144, 160, 192, 208, 224, 230, 232, 288, 343, 384, 460, 480, 512, 528, 832, 921,
                                                  1024, 2048, 4095, 4096, 4097, 65535, 65536, 1 << 20, (1 << 20) + 1, (1 << 30) - 1,
                                                  1 << 30, (1 << 31) - 1] + [int(x) for x in rng.integers(1, 1 << 31, 40)]))
    top = (1 << 31) - 1
    for d in divisors:
        ns = {0, 1, d - 1, d, d + 1, 2 * d - 1, 2 * d, top, top - 1, top - d, (top // d) * d, (top // d) * d - 1}
        ns |= {int(x) for x in rng.integers(0, 1 << 31, 64)}
        ns |= {k * d - 1 for k in rng.integers(1, max(2, top // d), 16)} | {k * d for k in rng.integers(1, max(2, top // d), 16)}
        for n in ns:
            n = int(n)
            if 0 <= n <= top:
                assert lib.rsp_fastdiv_check(d, n) == n // d, (d, n)
