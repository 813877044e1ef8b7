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


def test_tile_plan_option_changes_the_dispatch_not_the_interface():
    """rsp_conv3d_set_option("narrow_max_tiles", v): launches of less than one round of 128-wide tiles run on the 64-wide tile by
    default; 0 switches that off (the parity tests' second evaluation order), a negative value restores the default.  Host
    arithmetic only: checked through the dispatch predictor."""
    lib = _lib.load()
    d = _lib.ConvDesc(16, 4, 14, 14, 528, 4, 14, 14, 448, 1, 1, 1, 1, 1, 1, 0, 0, 0, 528, 448)      # S3D-G 14 x 14 pointwise: 98 x 4 tiles
    big = _lib.ConvDesc(32, 8, 28, 28, 256, 8, 28, 28, 256, 3, 3, 3, 1, 1, 1, 1, 1, 1, 256, 256)    # C3D conv3b: 1568 x 2 tiles
    name = lambda dd: lib.rsp_conv3d_kernel_name(ctypes.byref(dd), 0).decode()
    assert name(d).startswith("igemm_persist_kernel<128, 64,") and name(big).startswith("igemm_persist_kernel<128, 128,")
    prev = lib.rsp_conv3d_set_option(b"narrow_max_tiles", 0)
    try:
        assert prev == 512
        assert name(d).startswith("igemm_persist_kernel<128, 128,") and name(big).startswith("igemm_persist_kernel<128, 128,")
    finally:
        assert lib.rsp_conv3d_set_option(b"narrow_max_tiles", -1) == 0
    assert name(d).startswith("igemm_persist_kernel<128, 64,")
    # round 6: launches of less than one 128 x 64 unit per CU take the 32-wide tile (whole K per unit, no reduce launch); 33..64-column
    # launches of at least a round of 256-row tiles take the 256 x 64 instance
    tiny = _lib.ConvDesc(16, 4, 14, 14, 512, 4, 14, 14, 64, 1, 1, 1, 1, 1, 1, 0, 0, 0, 512, 64)       # S3D-G branch3 pointwise: 98 tiles
    tall = _lib.ConvDesc(32, 8, 28, 28, 64, 8, 28, 28, 64, 3, 3, 3, 1, 1, 1, 1, 1, 1, 64, 64)         # R3D-18 layer1: 784 tiles of 256 rows
    assert name(tiny).startswith("igemm_persist_kernel<128, 32,") and name(tall) == "igemm_persist_kernel<128, 64, 2, 2, true, 4>"
    assert lib.rsp_conv3d_set_option(b"narrow32_max_units", 0) == 256 and lib.rsp_conv3d_set_option(b"tall_min_tiles", 768) == 0
    try:
        assert name(tiny).startswith("igemm_persist_kernel<128, 64,") and name(tall) == "igemm_persist_kernel<256, 64, 4, 1, true, 3>"
    finally:
        assert lib.rsp_conv3d_set_option(b"narrow32_max_units", -1) == 0 and lib.rsp_conv3d_set_option(b"tall_min_tiles", -1) == 768
    assert name(tiny).startswith("igemm_persist_kernel<128, 32,") and name(tall).startswith("igemm_persist_kernel<128, 64,")
    assert lib.rsp_conv3d_set_option(b"no_such_option", 1) == -1 and b"unknown option" in lib.rsp_last_error()


def test_no_memset_or_memcpy_nodes_in_the_step_kernels():
    """Round 6: a hipMemsetAsync captured into a LINEAR HIP graph (rspnet_amd/graph_step.py, "lanes") was not reliably ordered against
    the kernels around it on this stack — R3D-18's shortcut input gradients came out wrong in a few replays per hundred
    (profiles/r06/experiments_r6.txt, r6race).  Everything a captured pretext step launches from this library is therefore a KERNEL: no
    hipMemset* / hipMemcpy* in the sources of the step's ops (augment.hip is the data path: never captured)."""
    import re
    csrc = os.path.join(ROOT, "rspnet_amd", "csrc")
    bad = []
    for name in sorted(os.listdir(csrc)):
        if not name.endswith((".hip", ".h")) or name == "augment.hip":
            continue
        with open(os.path.join(csrc, name)) as f:
            for i, line in enumerate(f, 1):
                code = line.split("//")[0]
                if re.search(r"\bhipMem(set|cpy)\w*\s*\(", code):
                    bad.append((name, i, line.strip()))
    assert not bad, bad
