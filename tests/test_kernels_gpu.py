"""GPU: every HIP kernel (through the C ABI, via rspnet_amd.ops.HipOps) against the torch-fp32 contract in
tests/cpu_ops.py on identical seeded inputs.  Tolerances are written per test; convs run on the exact-fp32 MFMA pipe
(fmaf chains), so they agree with the CPU reference to summation-order rounding (~1e-6 relative to |a|.|b|)."""
import numpy as np
import pytest
import torch

from cpu_ops import CpuOps
from rspnet_amd.ops import ConvGeom, PoolGeom

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
CPU = CpuOps()


@pytest.fixture(scope="module")
def hip():
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    return ops.backend()


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.rand(*shape, generator=g) * 2 - 1) * scale


def close(a, b, rtol, what=""):
    a = a.detach().cpu().double()
    b = b.detach().cpu().double()
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = (a - b).abs().max().item()
    ref = max(b.abs().max().item(), 1e-6)
    assert err <= rtol * ref, f"{what}: max err {err:.3e} vs ref max {ref:.3e} (rel {err / ref:.3e} > {rtol})"


# (N, D, H, W, Cin, Cout, k, s, p)
CONV_CASES = [
    (2, 4, 12, 12, 3, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),        # C3D conv1: Cin=3 scalar gather, BN=64 tile
    (2, 4, 10, 10, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # C3D conv2: vec4, 128x128 tile
    (1, 2, 7, 7, 256, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # C3D conv4a/5: small M, big K -> split-K
    (2, 4, 16, 16, 3, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),        # R3D-18 stem
    (2, 4, 8, 8, 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),        # R3D-18 strided 3x3x3
    (2, 4, 8, 8, 64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),        # R3D-18 downsample 1x1x1 s2
    (2, 3, 9, 9, 3, 83, (1, 7, 7), (1, 2, 2), (0, 3, 3)),          # R(2+1)D stem spatial: Cout=83 (odd)
    (2, 5, 6, 6, 83, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),         # R(2+1)D temporal: Cin=83 (scalar gather)
    (2, 5, 6, 6, 230, 128, (3, 1, 1), (2, 1, 1), (1, 0, 0)),       # R(2+1)D temporal stride 2, Cin%4==2
    (1, 4, 9, 9, 3, 64, (1, 7, 7), (2, 2, 2), (0, 3, 3)),          # S3D-G stem: (1,7,7) stride 2 in T too
    (2, 4, 7, 7, 192, 16, (1, 1, 1), (1, 1, 1), (0, 0, 0)),        # S3D-G pointwise to 16 ch (BN=32 tile)
    (2, 4, 7, 7, 16, 48, (1, 3, 3), (1, 1, 1), (0, 1, 1)),         # S3D-G (1,3,3), Cout=48
    (1, 1, 5, 5, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)),           # tiny: M=25 < one tile
    (1, 4, 130, 128, 32, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),    # 520 tiles = one whole round of 512 + a K-split tail of 8
    (1, 4, 70, 80, 32, 320, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # 175 x 3 tiles: whole rounds end mid-way (510), 3 column tiles
    # column / output-channel segments: the part of Cout (Cin for dgrad) beyond the last multiple of 128 runs on a narrower tile
    (2, 4, 12, 12, 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),      # R(2+1)D conv2 spatial: 144 = 128 + 16 (32-wide tile)
    (2, 4, 12, 12, 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),      # its temporal partner: dgrad N = 144
    (2, 4, 10, 10, 128, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1)),     # 288 = 256 + 32
    (1, 4, 14, 14, 96, 160, (3, 1, 1), (1, 1, 1), (1, 0, 0)),      # S3D-G: 160 = 128 + 32, Cin = 96
    (1, 2, 7, 7, 256, 576, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # 576 = 512 + 64, small M (K-split tail tiles in both segments)
    (2, 4, 9, 9, 64, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1)),         # 96-wide tile (4 waves along M); dgrad N = 64
    (2, 4, 9, 9, 96, 160, (3, 1, 1), (1, 1, 1), (1, 0, 0)),        # 160-wide tile forward, 96-wide tile in dgrad
    (1, 2, 7, 7, 160, 150, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # 150 columns in one 160-wide tile, K-split tail; dgrad N = 160
    # 129..144 columns: the 144-wide tile (four 32-wide column blocks + one 16-wide block on the 16x16x4 MFMA) — 132 / 140 columns
    # (ragged half block), whole rounds + K-split tail, strided (per-row output table in dgrad), depth-major rows with 144 columns
    (2, 4, 12, 12, 64, 132, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (1, 4, 70, 80, 32, 144, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # 175 tiles, bias + statistics on every one
    (2, 4, 9, 9, 144, 140, (3, 3, 3), (2, 2, 2), (1, 1, 1)),       # dgrad classes with N = 144; forward 140 columns
    (3, 4, 7, 7, 128, 144, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # slice-major K, depth-major rows (two linear runs per tile)
    # >= 768 tiles: the single-LDS-buffer mode of the 128- / 96-wide tiles (three workgroups per CU), with a K-split tail
    (1, 6, 130, 128, 16, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),    # 780 tiles of 128x128, K = 432
    (1, 6, 130, 128, 16, 96, (1, 3, 3), (1, 1, 1), (0, 1, 1)),     # 780 tiles of 128x96
    (1, 6, 130, 128, 128, 32, (1, 1, 1), (1, 1, 1), (0, 0, 0)),    # dgrad: 780 tiles of 128x128 over K = 32 (one chunk)
    # channel-slice-major K order (igemm_ks_kernel: C % 32 == 0, >= 48 chunks) on the 96-, 64- and 32-wide tiles; strided: per-class choice
    (1, 2, 6, 6, 128, 96, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 2, 6, 6, 128, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 2, 6, 6, 128, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 4, 8, 8, 32, 512, (3, 3, 3), (2, 2, 2), (1, 1, 1)),        # dgrad C = 512: the 8-tap class is slice-major, the others tap-major
    # depth-major row enumeration + skipped padding taps (slice-major kernel, N > 1, depth padding): frames of 147 / 49 / 16 rows per
    # sample (tiles straddle samples, ragged last tile, K-split tails), T = 1 (two of three depth taps are padding everywhere),
    # a strided layer (dgrad classes), a (3,1,1) temporal convolution, no depth padding at all (stays n-major)
    (3, 4, 7, 7, 128, 96, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (5, 2, 7, 7, 128, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (6, 1, 4, 4, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 5, 9, 9, 128, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    (3, 6, 5, 5, 1024, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (2, 6, 6, 6, 128, 64, (3, 3, 3), (1, 1, 1), (0, 1, 1)),
    (2, 8, 40, 40, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # 200 tiles x 1: whole frames per tile group, K-split tail
    # wgrad output-channel segments on the 32-wide tile
    (1, 3, 20, 20, 8, 20, (1, 3, 3), (1, 1, 1), (0, 1, 1)),        # 20 channels: one 32-wide tile
    (1, 3, 20, 20, 16, 272, (1, 1, 1), (1, 1, 1), (0, 0, 0)),      # 272 = 256 + 16; K = 16 (64-wide k tile: the 16 ride on the 64-row tile)
    # small launches of > 96 columns run on the 64-wide tile (narrow_tiles); with RSP_DIRECT_MAX_TILES set (the child run at the end of
    # this file) every small launch with Cin % 16 == 0 runs on igemm_direct_kernel: 1x1x1 stride 2, strided parity classes, slice-major
    # K with a K split, 16 / 150 / 576 columns — plus half-chunks that straddle taps (Cin = 112, 208: not multiples of 32) and a K
    # that ends inside a chunk (208 = 6.5 chunks)
    (2, 4, 14, 14, 208, 160, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 4, 14, 14, 112, 288, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (3, 4, 7, 7, 48, 72, (3, 3, 3), (2, 2, 2), (1, 1, 1)),         # Cin = 48, 72 columns (64 + 8), strided: dgrad classes with K = 72 * taps
    # 4-channel (zero-padded RGB) stems: direct LDS-halo kernel (conv_stem.hip)
    (2, 4, 12, 12, 4, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),        # C3D conv1: all 3 time-slices resident, 14-tap chunks
    (2, 5, 18, 20, 4, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),        # R3D stem: ring of 3 time-slices, stride 2 de-interleave
    (1, 3, 23, 17, 4, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)),        # R(2+1)D stem: Cout=45, ragged patches
    (1, 4, 9, 9, 4, 64, (1, 7, 7), (2, 2, 2), (0, 3, 3)),          # S3D-G stem: stride 2 in T too
    (2, 2, 16, 8, 4, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),         # narrow frame: 16x8 patches
    (1, 2, 33, 35, 4, 32, (3, 5, 5), (1, 1, 2), (1, 2, 2)),        # mixed strides, 5x5, Cout=32
]


@pytest.mark.parametrize("case", CONV_CASES, ids=lambda c: "x".join(map(str, c[:6])) + f"k{c[6]}s{c[7]}")
def test_conv_fwd_dgrad_wgrad(hip, case):
    N, D, H, W, Cin, Cout, k, s, p = case
    g = ConvGeom(N, D, H, W, Cin, Cout, k, s, p)
    x = rnd(N, D, H, W, Cin, seed=1)
    w = rnd(Cout, Cin, *k, seed=2, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    b = rnd(Cout, seed=3)
    # forward (+ bias, + stat partials)
    y_ref, st_ref = CPU.conv_fwd(g, x, w, b, True)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    y, st = hip.conv_fwd(g, xd, hip.conv_pack_fwd(g, wd), bd, True)
    close(y, y_ref, 2e-5, "conv fwd")
    close(st.double().sum(0), st_ref.double().sum(0), 2e-5, "stat partials")
    # forward without bias / stats
    y2, st2 = hip.conv_fwd(g, xd, hip.conv_pack_fwd(g, wd), None, False)
    assert st2 is None
    close(y2, y_ref - b, 2e-5, "conv fwd nobias")
    # dgrad
    dy = rnd(*y_ref.shape, seed=4)
    dx_ref = CPU.conv_dgrad(g, dy, w)
    dx = hip.conv_dgrad(g, dy.to(DEV), wd)
    close(dx, dx_ref, 2e-5, "dgrad")
    # wgrad (+ dbias)
    dw_ref = torch.empty_like(w)
    db_ref = torch.empty_like(b)
    CPU.conv_wgrad(g, x, dy, dw_ref, db_ref)
    dw = torch.empty_like(wd)
    db = torch.empty_like(bd)
    hip.conv_wgrad(g, xd, dy.to(DEV), dw, db)
    close(dw, dw_ref, 2e-5, "wgrad")
    close(db, db_ref, 2e-5, "dbias")


# Round 6 instances, at sizes the checker finishes in seconds (the thresholds that select them are planning options of the library):
#   * the 256 x 64 tile of the persistent kernel (tall_tiles: 33..64 columns; OFF by default — measured neutral-to-negative per step in
#     round 6, profiles/r06/experiments_r6.txt — and selected here through "tall_min_tiles") — tap-major and
#     slice-major K, depth-major rows with frames longer / shorter than a tile (two linear runs / the per-row table), a ragged last
#     tile, a K-split tail whose partial rows are 256-row tiles, bias + statistics on every tile, the input gradient of a 64-channel input;
#   * 32-wide tiles with the whole K per unit for launches of less than one 128 x 64 unit per CU (narrow_bn), both K orders.
TALL_CASES = [
    (2, 4, 24, 24, 16, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (3, 4, 16, 16, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (8, 4, 9, 7, 64, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 4, 130, 128, 32, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 5, 23, 19, 64, 8, (3, 1, 1), (1, 1, 1), (1, 0, 0)),         # the input gradient has 64 columns (forward: 8)
]
NARROW32_CASES = [
    (2, 4, 14, 14, 512, 64, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    (2, 2, 7, 7, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 4, 14, 14, 96, 208, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    (4, 1, 4, 4, 256, 512, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
]


@pytest.mark.parametrize("case", TALL_CASES + NARROW32_CASES, ids=lambda c: "x".join(map(str, c[:6])) + f"k{c[6]}s{c[7]}")
def test_conv_tall_and_narrow32_instances(hip, case):
    N, D, H, W, Cin, Cout, k, s, p = case
    g = ConvGeom(N, D, H, W, Cin, Cout, k, s, p)
    x = rnd(N, D, H, W, Cin, seed=21)
    w = rnd(Cout, Cin, *k, seed=22, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    b = rnd(Cout, seed=23)
    y_ref, st_ref = CPU.conv_fwd(g, x, w, b, True)
    dy = rnd(*y_ref.shape, seed=24)
    dx_ref = CPU.conv_dgrad(g, dy, w)
    xd, wd, bd = x.to(DEV), w.to(DEV), b.to(DEV)
    tall = case in TALL_CASES
    if tall:
        hip.set_option("tall_min_tiles", 2)
        hip.set_option("narrow32_max_units", 0)      # (at these sizes the launch would otherwise count as a tiny one)
    try:
        y, st = hip.conv_fwd(g, xd, hip.conv_pack_fwd(g, wd), bd, True)
        kf = hip.lib.rsp_last_conv_kernel().decode()
        dx = hip.conv_dgrad(g, dy.to(DEV), wd)
        kd = hip.lib.rsp_last_conv_kernel().decode()
    finally:
        if tall:
            hip.set_option("tall_min_tiles", -1)
            hip.set_option("narrow32_max_units", -1)
    close(y, y_ref, 2e-5, "conv fwd")
    close(st.double().sum(0), st_ref.double().sum(0), 2e-5, "stat partials")
    close(dx, dx_ref, 2e-5, "dgrad")
    if tall:
        assert "<256, 64," in (kf if Cout > 32 else kd), (kf, kd)
    elif s == (1, 1, 1):
        assert "<128, 32," in kf, kf


def test_conv_two_level_summation_instance(hip):
    """igemm_persist_fold_kernel (round 6, OFF by default: "two_level_min_chunks"): the long-K slice-major 128-wide launch with K summed
    in panels of 512 products.  Same convolution as the one-chain instance at the checker's tolerance, forward / statistics / input
    gradient, with a K-split tail; against an fp64 convolution its error is the smaller one (the point of the instance)."""
    N, D, H, W, Cin, Cout, k, s, p = (8, 4, 32, 32, 128, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1))      # 512 tiles of 128 x 128, K = 3456
    g = ConvGeom(N, D, H, W, Cin, Cout, k, s, p)
    x = rnd(N, D, H, W, Cin, seed=31)
    w = rnd(Cout, Cin, *k, seed=32, scale=(Cin * 27) ** -0.5)
    y_ref, st_ref = CPU.conv_fwd(g, x, w, None, True)
    y64 = torch.nn.functional.conv3d(x[:1].double().permute(0, 4, 1, 2, 3), w.double(), None, s, p).permute(0, 2, 3, 4, 1)      # (first sample)
    dy = rnd(*y_ref.shape, seed=34)
    dx_ref = CPU.conv_dgrad(g, dy, w)
    xd, wd = x.to(DEV), w.to(DEV)
    wp = hip.conv_pack_fwd(g, wd)
    y1, _ = hip.conv_fwd(g, xd, wp, None, True)
    k1 = hip.lib.rsp_last_conv_kernel().decode()
    hip.set_option("two_level_min_chunks", 8)
    try:
        y2, st2 = hip.conv_fwd(g, xd, wp, None, True)
        k2 = hip.lib.rsp_last_conv_kernel().decode()
        dx2 = hip.conv_dgrad(g, dy.to(DEV), wd)
        kd = hip.lib.rsp_last_conv_kernel().decode()
    finally:
        hip.set_option("two_level_min_chunks", -1)
    assert "fold" not in k1 and "igemm_persist_fold_kernel<128, 128" in k2, (k1, k2, kd)
    close(y2, y_ref, 2e-5, "conv fwd (two-level)")
    close(st2.double().sum(0), st_ref.double().sum(0), 2e-5, "stat partials (two-level)")
    close(dx2, dx_ref, 2e-5, "dgrad (two-level)")
    e1 = float((y1[:1].cpu().double() - y64).pow(2).mean().sqrt())
    e2 = float((y2[:1].cpu().double() - y64).pow(2).mean().sqrt())
    print(f"\nrms error against fp64 at K = {Cin * 27}: one chain {e1:.3e}, panels of 512 {e2:.3e}")
    assert e2 < e1


def _fuzz_cases(n=48, seed=20261002):
    """Seeded random conv geometries: odd sizes, mixed kernels / strides / paddings, channel counts on every code path
    (4-channel stems, scalar-gather fallbacks, 32/64/128-wide tiles, kernels smaller than the stride)."""
    import random as _r
    rng = _r.Random(seed)
    cases = []
    while len(cases) < n:
        k = tuple(rng.choice((1, 1, 3, 3, 5, 7)) for _ in range(3))
        s = tuple(rng.choice((1, 1, 2)) for _ in range(3))
        p = tuple(rng.randint(0, kk // 2) for kk in k)
        D, H, W = rng.randint(1, 6), rng.randint(3, 21), rng.randint(3, 21)
        if any((i + 2 * pp - kk) < 0 for i, kk, pp in zip((D, H, W), k, p)):
            continue
        cin = rng.choice((3, 4, 4, 8, 16, 20, 32, 64, 83))
        cout = rng.choice((8, 16, 32, 45, 64, 64, 96, 130))
        if cin * k[0] * k[1] * k[2] > 6000:
            continue
        cases.append((rng.randint(1, 3), D, H, W, cin, cout, k, s, p))
    return cases


@pytest.mark.parametrize("case", _fuzz_cases(), ids=lambda c: "x".join(map(str, c[:6])) + f"k{c[6]}s{c[7]}p{c[8]}")
def test_conv_fuzz(hip, case):
    N, D, H, W, Cin, Cout, k, s, p = case
    g = ConvGeom(N, D, H, W, Cin, Cout, k, s, p)
    x = rnd(N, D, H, W, Cin, seed=11)
    w = rnd(Cout, Cin, *k, seed=12, scale=(Cin * k[0] * k[1] * k[2]) ** -0.5)
    b = rnd(Cout, seed=13)
    y_ref, st_ref = CPU.conv_fwd(g, x, w, b, True)
    xd, wd = x.to(DEV), w.to(DEV)
    y, st = hip.conv_fwd(g, xd, hip.conv_pack_fwd(g, wd), b.to(DEV), True)
    close(y, y_ref, 2e-5, "conv fwd")
    close(st.double().sum(0), st_ref.double().sum(0), 5e-5, "stat partials")
    dy = rnd(*y_ref.shape, seed=14)
    close(hip.conv_dgrad(g, dy.to(DEV), wd), CPU.conv_dgrad(g, dy, w), 2e-5, "dgrad")
    dw_ref, db_ref = torch.empty_like(w), torch.empty_like(b)
    CPU.conv_wgrad(g, x, dy, dw_ref, db_ref)
    dw, db = torch.empty_like(wd), torch.empty(Cout, device=DEV)
    hip.conv_wgrad(g, xd, dy.to(DEV), dw, db)
    close(dw, dw_ref, 2e-5, "wgrad")
    close(db, db_ref, 2e-5, "dbias")


def _fuzz_depth_cases(n=20, seed=20261003):
    """Seeded random geometries for the padded-tap skipping: depth taps with depth padding, batches of 2 / 8 / 16 (one and eight
    parts of the depth-major enumeration), channel counts on the slice-major (C % 32 == 0, long K) and the tap-major walks (one tap
    per chunk, several taps per chunk, taps straddling chunks), strides, frames from a fraction of a tile to several tiles."""
    import random as _r
    rng = _r.Random(seed)
    cases = []
    while len(cases) < n:
        k = (rng.choice((3, 3, 3, 5, 7)), rng.choice((1, 3, 3)), rng.choice((1, 3, 3)))
        s = tuple(rng.choice((1, 1, 1, 2)) for _ in range(3))
        p = (rng.randint(1, k[0] // 2), rng.randint(0, k[1] // 2), rng.randint(0, k[2] // 2))
        N = rng.choice((2, 8, 8, 16))
        D, H, W = rng.randint(1, 6), rng.randint(4, 24), rng.randint(4, 24)
        if any((i + 2 * pp - kk) < 0 for i, kk, pp in zip((D, H, W), k, p)):
            continue
        cin = rng.choice((4, 32, 32, 64, 96, 128, 144, 160))
        cout = rng.choice((32, 64, 128, 144))
        flop = 2.0 * N * D * H * W * cin * cout * k[0] * k[1] * k[2] / (s[0] * s[1] * s[2])
        if flop > 25e9 or cin * k[0] * k[1] * k[2] > 14000:
            continue
        cases.append((N, D, H, W, cin, cout, k, s, p))
    return cases


@pytest.mark.parametrize("case", _fuzz_depth_cases(), ids=lambda c: "x".join(map(str, c[:6])) + f"k{c[6]}s{c[7]}p{c[8]}")
def test_conv_fuzz_padded_depth(hip, case):
    test_conv_fuzz(hip, case)


def test_conv_is_run_to_run_deterministic(hip):
    g = ConvGeom(1, 2, 7, 7, 256, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1))   # split-K path
    x, w = rnd(1, 2, 7, 7, 256, seed=5).to(DEV), rnd(512, 256, 3, 3, 3, seed=6).to(DEV)
    wp = hip.conv_pack_fwd(g, w)
    y1, s1 = hip.conv_fwd(g, x, wp, None, True)
    y2, s2 = hip.conv_fwd(g, x, wp, None, True)
    assert torch.equal(y1, y2) and torch.equal(s1, s2)
    dy = rnd(*y1.shape, seed=7).to(DEV)
    a, b = torch.empty_like(w), torch.empty_like(w)
    hip.conv_wgrad(g, x, dy, a)
    hip.conv_wgrad(g, x, dy, b)
    assert torch.equal(a, b)


@pytest.mark.parametrize("C,tiles_rows", [(64, 5000), (83, 130), (512, 98)])
def test_bn_finalize(hip, C, tiles_rows):
    rows = tiles_rows
    y = rnd(rows, C, seed=1) * 2 + 0.5
    bias = rnd(C, seed=2)
    gamma, beta = rnd(C, seed=3) + 1.5, rnd(C, seed=4)
    rm, rv = rnd(C, seed=5), rnd(C, seed=6) + 1.5
    tiles = (rows + 127) // 128
    part = torch.zeros(tiles, C, 2)
    for t in range(tiles):
        blk = y[t * 128:(t + 1) * 128].double()
        part[t, :, 0] = blk.sum(0).float()
        part[t, :, 1] = (blk * blk).sum(0).float()
    rm_ref, rv_ref = rm.clone(), rv.clone()
    mi_ref, ss_ref = CPU.bn_finalize(part, rows, bias, gamma, beta, 1e-5, 0.1, rm_ref, rv_ref)
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    mi, ss = hip.bn_finalize(part.to(DEV), rows, bias.to(DEV), gamma.to(DEV), beta.to(DEV), 1e-5, 0.1, rm_d, rv_d)
    close(mi, mi_ref, 1e-5, "mean/invstd")
    close(ss, ss_ref, 1e-5, "scale/shift")
    close(rm_d, rm_ref, 1e-5, "running_mean")
    close(rv_d, rv_ref, 1e-5, "running_var")
    # F.batch_norm agreement (the reference's op): running stats after one train-mode call
    rm2, rv2 = rm.clone(), rv.clone()
    torch.nn.functional.batch_norm((y + bias).t().reshape(1, C, rows), rm2, rv2, gamma, beta, True, 0.1, 1e-5)
    close(rm_d, rm2, 2e-5, "running_mean vs F.batch_norm")
    close(rv_d, rv2, 2e-5, "running_var vs F.batch_norm")


POOL_CASES = [
    # N, D, H, W, C, k, s, relu, residual
    (2, 4, 8, 8, 64, (1, 2, 2), (1, 2, 2), True, False),
    (2, 4, 8, 8, 128, (2, 2, 2), (2, 2, 2), True, False),
    (2, 3, 7, 7, 256, (1, 1, 1), (1, 1, 1), True, False),
    (2, 5, 7, 7, 64, (2, 2, 2), (2, 2, 2), True, False),     # odd sizes: floor mode drops the tail, dy must still be written
    (2, 3, 5, 5, 83, (1, 1, 1), (1, 1, 1), True, False),     # odd channel count -> scalar path
    (2, 3, 5, 5, 64, (1, 1, 1), (1, 1, 1), True, True),      # residual add before ReLU
    (2, 3, 5, 5, 64, (1, 1, 1), (1, 1, 1), False, False),    # BN only (shortcut branch)
    (1, 2, 4, 4, 1152, (1, 1, 1), (1, 1, 1), True, False),   # > 1024 channels
    (2, 8, 72, 64, 64, (1, 1, 1), (1, 1, 1), True, False),   # 73 728 positions: > 512 reduce blocks -> separate bwd finalize launch
    (2, 8, 72, 64, 64, (1, 2, 2), (1, 2, 2), True, False),
]


def _fuzz_bn_pools(n=20, seed=9):
    """Seeded random fused BN(+residual)(+ReLU)(+disjoint pool) units: ragged sizes (floor-mode tails), odd channel counts."""
    import random as _r
    rng = _r.Random(seed)
    out = []
    for _ in range(n):
        k = tuple(rng.choice((1, 1, 2)) for _ in range(3))
        D, H, W = (rng.randint(kk, 7) for kk in k)
        out.append((rng.randint(1, 3), D, H, W, rng.choice((8, 32, 64, 83, 96, 260)), k, k, rng.random() < 0.8,
                    k == (1, 1, 1) and rng.random() < 0.5))
    return out


@pytest.mark.parametrize("case", POOL_CASES + _fuzz_bn_pools(), ids=lambda c: "x".join(map(str, c[:5])) + f"k{c[5]}r{int(c[7])}{int(c[8])}")
def test_bn_act_pool_fwd_bwd(hip, case):
    N, D, H, W, C, k, s, relu, use_res = case
    pg = PoolGeom(N, D, H, W, C, k, s, (0, 0, 0))
    y = rnd(N, D, H, W, C, seed=1) * 2
    res = rnd(N, D, H, W, C, seed=2) if use_res else None
    gamma, beta = rnd(C, seed=3) + 1.5, rnd(C, seed=4) * 0.5
    # stats the way the conv epilogue would give them (single tile here)
    rows = N * D * H * W
    yy = y.reshape(rows, C).double()
    part = torch.stack([yy.sum(0), (yy * yy).sum(0)], 1).float().unsqueeze(0)
    mi, ss = CPU.bn_finalize(part, rows, None, gamma, beta, 1e-5, 0.1, None, None)
    out_ref = CPU.bn_act_pool_fwd(pg, y, ss, res, relu)
    yd, ssd, mid = y.to(DEV), ss.to(DEV), mi.to(DEV)
    resd = res.to(DEV) if use_res else None
    out = hip.bn_act_pool_fwd(pg, yd, ssd, resd, relu)
    close(out, out_ref, 1e-6, "bn_act_pool fwd")
    dout = rnd(*out_ref.shape, seed=5)
    dg_ref, db_ref = torch.empty(C), torch.empty(C)
    dy_ref, dres_ref = CPU.bn_act_pool_bwd(pg, y, res, dout, gamma, mi, ss, relu, use_res, dg_ref, db_ref)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dy, dres = hip.bn_act_pool_bwd(pg, yd, resd, dout.to(DEV), gamma.to(DEV), mid, ssd, relu, use_res, dg, db)
    close(dy, dy_ref, 2e-5, "bn bwd dy")
    close(dg, dg_ref, 2e-5, "dgamma")
    close(db, db_ref, 2e-5, "dbeta")
    if use_res:
        close(dres, dres_ref, 1e-6, "dres")


@pytest.mark.parametrize("B,P,C", [(4, 8, 512), (3, 98, 512), (2, 16, 1024)])
def test_head_fwd_bwd(hip, B, P, C):
    feat = rnd(B, P, 1, 1, C, seed=1)
    w1, w2 = rnd(128, C, seed=2, scale=C ** -0.5), rnd(128, C, seed=3, scale=C ** -0.5)
    b1, b2 = rnd(128, seed=4) * 0.1, rnd(128, seed=5) * 0.1
    o1r, o2r, pr, rr = CPU.head_fwd(feat, w1, b1, w2, b2)
    d = [t.to(DEV) for t in (feat, w1, b1, w2, b2)]
    o1, o2, pooled, raw = hip.head_fwd(*d)
    close(o1, o1r, 1e-5, "head out1")
    close(o2, o2r, 1e-5, "head out2")
    close(pooled, pr, 1e-5, "pooled")
    close(raw, rr, 1e-5, "raw")
    g1, g2 = rnd(B, 128, seed=6), rnd(B, 128, seed=7)
    ref = [torch.empty_like(w1), torch.empty_like(b1), torch.empty_like(w2), torch.empty_like(b2)]
    dfeat_ref = CPU.head_bwd(g1, g2, pr, rr, w1, w2, tuple(feat.shape), *ref)
    out = [torch.empty_like(t, device=DEV) for t in ref]
    dfeat = hip.head_bwd(g1.to(DEV), g2.to(DEV), pooled, raw, d[1], d[3], tuple(feat.shape), *out)
    close(dfeat, dfeat_ref, 2e-5, "dfeat")
    for a, b, n in zip(out, ref, ("dw1", "db1", "dw2", "db2")):
        close(a, b, 2e-5, n)


@pytest.mark.parametrize("B,K", [(4, 64), (32, 16384), (40, 1000)])
def test_logits_and_loss(hip, B, K):
    dim = 128
    f = [torch.nn.functional.normalize(rnd(B, dim, seed=i), dim=1) for i in range(6)]
    queue = torch.nn.functional.normalize(rnd(dim, K, seed=9), dim=0)
    ref = CPU.logits_fwd(*f, queue, 1 / 0.07)
    fd = [t.to(DEV) for t in f]
    qd = queue.to(DEV)
    out = hip.logits_fwd(*fd, qd, 1 / 0.07)
    for a, b, n in zip(out, ref, ("logits1", "logits2", "l_pos_M", "l_neg_M")):
        close(a, b, 1e-5, n)
    lr = CPU.loss_fwd_bwd(*ref, 2.0, 1.0, 1.0)
    lo = hip.loss_fwd_bwd(*out, 2.0, 1.0, 1.0)
    close(lo[0], lr[0], 1e-5, "losses")
    for a, b, n in zip(lo[1:], lr[1:], ("dlogits1", "dlogits2", "dlpos", "dlneg")):
        close(a, b, 2e-5, n)
    dq_ref = CPU.logits_bwd(*lr[1:], f[2], f[3], f[4], f[5], queue, 1 / 0.07)
    dq = hip.logits_bwd(*lo[1:], fd[2], fd[3], fd[4], fd[5], qd, 1 / 0.07)
    close(dq[0], dq_ref[0], 2e-5, "dqA")
    close(dq[1], dq_ref[1], 2e-5, "dqM")


def test_loss_weights_and_inactive_margin(hip):
    B, K1 = 6, 33
    l1, l2 = rnd(B, K1, seed=1) * 5, rnd(B, K1, seed=2) * 5
    lp = torch.tensor([[5.0], [0.1], [3.0], [-1.0], [2.0], [9.0]])
    ln = torch.tensor([[1.0], [0.2], [1.0], [1.0], [2.0], [0.0]])   # rows 0 and 5 are beyond the margin
    ref = CPU.loss_fwd_bwd(l1, l2, lp, ln, 2.0, 0.7, 1.3)
    out = hip.loss_fwd_bwd(l1.to(DEV), l2.to(DEV), lp.to(DEV), ln.to(DEV), 2.0, 0.7, 1.3)
    for a, b in zip(out, ref):
        close(a, b, 1e-5, "weighted loss")


def test_queue_enqueue_and_rows_gather(hip):
    q = rnd(128, 64, seed=1)
    keys = rnd(8, 128, seed=2)
    qd = q.to(DEV)
    hip.queue_enqueue(qd, 24, keys.to(DEV))
    CPU.queue_enqueue(q, 24, keys)
    assert torch.equal(qd.cpu(), q)
    x = rnd(16, 256, seed=3)
    idx = torch.tensor([3, 3, 0, 15, 7], dtype=torch.int32)
    assert torch.equal(hip.rows_gather(x.to(DEV), idx.to(DEV)).cpu(), CPU.rows_gather(x, idx))


def test_clip_gather(hip):
    im = rnd(4, 3, 32, 10, 12, seed=1)
    src = torch.tensor([2, 0, 3, 1, 2], dtype=torch.int32)
    step = torch.tensor([1, 2, 2, 1, 2], dtype=torch.int32)
    out = hip.clip_gather(im.to(DEV), src.to(DEV), step.to(DEV), 16)
    assert torch.equal(out.cpu(), CPU.clip_gather(im, src, step, 16))
    out4 = hip.clip_gather(im.to(DEV), src.to(DEV), step.to(DEV), 16, 4)        # zero channel padding for the stems
    assert torch.equal(out4.cpu(), CPU.clip_gather(im, src, step, 16, 4))
    # the step's three gathers as one launch (rsp_clip_gather_multi), 3 -> 4 channels (fast path) and as they are (per-job fallback)
    im2 = rnd(*im.shape, seed=12)
    src2 = torch.flip(src, dims=[0]).contiguous()
    jobs = [(im, src, step), (im2, src2, step), (im, src2, torch.ones_like(step))]
    for c_out in (4, None):
        outs = hip.clip_gather_multi([(a.to(DEV), b.to(DEV), c.to(DEV)) for a, b, c in jobs], 16, c_out)
        for o, (a, b, c) in zip(outs, jobs):
            assert torch.equal(o.cpu(), CPU.clip_gather(a, b, c, 16, c_out))
    assert float(out4[..., 3].abs().max()) == 0


@pytest.mark.parametrize("n", [4096, 1000003])
def test_momentum_and_sgd(hip, n):
    k, q = rnd(n, seed=1), rnd(n, seed=2)
    kd = k.to(DEV)
    hip.momentum_update(kd, q.to(DEV), 0.999)
    CPU.momentum_update(k, q, 0.999)
    close(kd, k, 1e-7, "momentum")
    p, g, buf = rnd(n, seed=3), rnd(n, seed=4), rnd(n, seed=5)
    for first in (True, False):
        pd, bd = p.to(DEV), buf.to(DEV)
        hip.sgd_step(pd, g.to(DEV), bd, 0.05, 0.9, 1e-4, 1.0, first)
        p2, b2 = p.clone(), buf.clone()
        CPU.sgd_step(p2, g, b2, 0.05, 0.9, 1e-4, 1.0, first)
        close(pd, p2, 1e-6, "sgd p")
        close(bd, b2, 1e-6, "sgd buf")


def test_c_abi_rejects_bad_arguments(hip):
    from rspnet_amd._lib import RspError
    g = ConvGeom(1, 2, 4, 4, 8, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    with pytest.raises(RspError):
        hip.conv_fwd(g, torch.zeros(1, 2, 4, 4, 8), torch.zeros(8 * 216, device=DEV), None, False)   # CPU tensor
    with pytest.raises(RspError):
        hip.queue_enqueue(torch.zeros(128, 64, device=DEV), 60, torch.zeros(8, 128, device=DEV))      # slab past K


MAXPOOL_CASES = [
    (2, 4, 9, 9, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)),     # R3D-18 stem pool
    (2, 3, 8, 8, 192, (1, 3, 3), (1, 2, 2), (0, 1, 1)),    # S3D-G maxPool1/2
    (2, 4, 6, 6, 48, (3, 3, 3), (1, 1, 1), (1, 1, 1)),     # S3D-G inception branch3 (stride 1)
    (2, 4, 6, 6, 83, (2, 2, 2), (2, 2, 2), (0, 0, 0)),     # disjoint, odd channels
    # stride 1 along W, 3-wide: the four-outputs-per-thread row kernel (ragged last quad, no W padding, strided H, Wo == 4)
    (1, 3, 7, 13, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 2, 5, 9, 8, (1, 3, 3), (1, 2, 1), (0, 0, 0)),
    (1, 2, 4, 4, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 5, 9, 6, 192, (2, 1, 3), (2, 1, 1), (0, 0, 1)),
    # the sliding 3x3x3 / stride 1 / pad 1 kernels (maxpool333_*): whole-row runs, split runs (14 = 2 x 7, 28 = 4 x 7, 9 = 5 + 4),
    # single-column and single-plane inputs
    (16, 2, 7, 7, 832, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 4, 14, 14, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 2, 5, 28, 16, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 3, 4, 9, 8, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (2, 1, 1, 1, 4, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    (1, 1, 6, 2, 12, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
]


def _fuzz_pools(n=24, seed=7):
    """Seeded random MaxPool3d geometries (overlapping / padded / disjoint windows; float4 and scalar channel counts)."""
    import random as _r
    rng = _r.Random(seed)
    out = []
    while len(out) < n:
        k = tuple(rng.choice((1, 2, 3)) for _ in range(3))
        s = tuple(rng.choice((1, 2)) for _ in range(3))
        p = tuple(rng.randint(0, kk // 2) for kk in k)
        D, H, W = rng.randint(1, 6), rng.randint(2, 13), rng.randint(2, 13)
        if any((i + 2 * pp - kk) < 0 for i, kk, pp in zip((D, H, W), k, p)):
            continue
        out.append((rng.randint(1, 3), D, H, W, rng.choice((4, 16, 48, 64, 83, 130)), k, s, p))
    return out


@pytest.mark.parametrize("case", MAXPOOL_CASES + _fuzz_pools(), ids=lambda c: "x".join(map(str, c[:5])) + f"k{c[5]}s{c[6]}p{c[7]}")
def test_maxpool_fwd_bwd(hip, case):
    N, D, H, W, C, k, s, p = case
    pg = PoolGeom(N, D, H, W, C, k, s, p)
    x = torch.relu(rnd(N, D, H, W, C, seed=1))          # post-ReLU input: exact-zero ties exercise "first maximum"
    o_ref, i_ref = CPU.maxpool_fwd(pg, x, True)
    o, idx = hip.maxpool_fwd(pg, x.to(DEV), True)
    assert torch.equal(o.cpu(), o_ref)
    assert torch.equal(idx.cpu(), i_ref)
    dout = rnd(*o_ref.shape, seed=2)
    close(hip.maxpool_bwd(pg, dout.to(DEV), idx), CPU.maxpool_bwd(pg, dout, i_ref), 1e-6, "maxpool bwd")
    o2, none = hip.maxpool_fwd(pg, x.to(DEV), False)
    assert none is None and torch.equal(o2.cpu(), o_ref)


@pytest.mark.parametrize("N,P,C", [(2, 32, 64), (3, 18, 192), (2, 8, 48)])
def test_gate_fwd_bwd_and_channel_slices(hip, N, P, C):
    x = torch.relu(rnd(N, P, 1, 1, C, seed=1))
    w, b = rnd(C, C, 1, 1, 1, seed=2, scale=C ** -0.5), rnd(C, seed=3)
    o_ref, m_ref, g_ref = CPU.gate_fwd(x, w, b)
    # write into a channel slice of a wider tensor, as the inception concat does
    wide = torch.zeros(N, P, 1, 1, C + 40, device=DEV)
    view = wide[..., 24:24 + C]
    o, mean, gate = hip.gate_fwd(x.to(DEV), w.to(DEV), b.to(DEV), out=view)
    close(view, o_ref, 1e-5, "gate out (slice)")
    assert float(wide[..., :24].abs().max()) == 0 and float(wide[..., 24 + C:].abs().max()) == 0
    close(mean, m_ref, 1e-5, "gate mean")
    close(gate, g_ref, 1e-5, "gate")
    dwide = rnd(N, P, 1, 1, C + 40, seed=4)
    dw_ref, db_ref = torch.empty_like(w), torch.empty_like(b)
    dx_ref = CPU.gate_bwd(x, dwide[..., 24:24 + C], w, m_ref, g_ref, dw_ref, db_ref)
    dw, db = torch.empty_like(w, device=DEV), torch.empty_like(b, device=DEV)
    dx = hip.gate_bwd(x.to(DEV), dwide.to(DEV)[..., 24:24 + C], w.to(DEV), mean, gate, dw, db)
    close(dx, dx_ref, 2e-5, "gate dx")
    close(dw, dw_ref, 2e-5, "gate dw")
    close(db, db_ref, 2e-5, "gate db")


def test_bn_act_pool_channel_slices(hip):
    N, D, H, W, C = 2, 3, 5, 5, 64
    pg = PoolGeom(N, D, H, W, C)
    y = rnd(N, D, H, W, C, seed=1) * 2
    gamma, beta = rnd(C, seed=3) + 1.5, rnd(C, seed=4) * 0.5
    rows = N * D * H * W
    yy = y.reshape(rows, C).double()
    part = torch.stack([yy.sum(0), (yy * yy).sum(0)], 1).float().unsqueeze(0)
    mi, ss = CPU.bn_finalize(part, rows, None, gamma, beta, 1e-3, 0.001, None, None)
    out_ref = CPU.bn_act_pool_fwd(pg, y, ss, None, True)
    wide = torch.zeros(N, D, H, W, C + 16, device=DEV)
    hip.bn_act_pool_fwd(pg, y.to(DEV), ss.to(DEV), None, True, out=wide[..., 16:])
    close(wide[..., 16:], out_ref, 1e-6, "sliced fwd")
    assert float(wide[..., :16].abs().max()) == 0
    dwide = rnd(N, D, H, W, C + 16, seed=5)
    dg_ref, db_ref = torch.empty(C), torch.empty(C)
    dy_ref, _ = CPU.bn_act_pool_bwd(pg, y, None, dwide[..., 16:], gamma, mi, ss, True, False, dg_ref, db_ref)
    dg, db = torch.empty(C, device=DEV), torch.empty(C, device=DEV)
    dy, _ = hip.bn_act_pool_bwd(pg, y.to(DEV), None, dwide.to(DEV)[..., 16:], gamma.to(DEV), mi.to(DEV), ss.to(DEV), True,
                                False, dg, db)
    close(dy, dy_ref, 2e-5, "sliced bwd dy")
    close(dg, dg_ref, 2e-5, "sliced dgamma")


def test_mlp_head_pieces(hip):
    B, C, dim = 5, 512, 128
    feat = rnd(B, 2, 3, 3, C, seed=1)
    close(hip.spatial_mean_fwd(feat.to(DEV)), CPU.spatial_mean_fwd(feat), 1e-6, "spatial mean")
    dm = rnd(B, C, seed=2)
    close(hip.spatial_mean_bwd(dm.to(DEV), tuple(feat.shape)), CPU.spatial_mean_bwd(dm, tuple(feat.shape)), 1e-6, "mean bwd")
    x, w, b = rnd(B, C, seed=3), rnd(dim, C, seed=4, scale=C ** -0.5), rnd(dim, seed=5)
    for relu in (True, False):
        y_ref = CPU.linear_fwd(x, w, b, relu)
        y = hip.linear_fwd(x.to(DEV), w.to(DEV), b.to(DEV), relu)
        close(y, y_ref, 1e-5, "linear fwd")
        dy = rnd(B, dim, seed=6)
        dw_ref, db_ref = torch.empty_like(w), torch.empty_like(b)
        dx_ref = CPU.linear_bwd(x, y_ref, dy, w, relu, dw_ref, db_ref)
        dw, db = torch.empty_like(w, device=DEV), torch.empty_like(b, device=DEV)
        dx = hip.linear_bwd(x.to(DEV), y, dy.to(DEV), w.to(DEV), relu, dw, db)
        close(dx, dx_ref, 2e-5, "linear dx")
        close(dw, dw_ref, 2e-5, "linear dw")
        close(db, db_ref, 2e-5, "linear db")
    r = rnd(B, dim, seed=7)
    close(hip.l2norm_fwd(r.to(DEV)), CPU.l2norm_fwd(r), 1e-6, "l2norm")
    g = rnd(B, dim, seed=8)
    close(hip.l2norm_bwd(r.to(DEV), g.to(DEV)), CPU.l2norm_bwd(r, g), 2e-5, "l2norm bwd")


# the full-size shapes whose dispatch decisions differ: slice-major / tap-major K walk, stem (resident and streaming), odd tiles,
# column segments, single- and multi-launch strided dgrad, the 343-tap stem on the implicit-GEMM path
NAME_CASES = [
    (32, 16, 112, 112, 4, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)),      # C3D conv1: stem_resident
    (32, 16, 56, 56, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # conv2: 54 chunks, slice-major since round 5
    (32, 4, 14, 14, 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1)),       # conv4b: slice-major
    (32, 16, 112, 112, 4, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),       # R3D stem
    (32, 8, 28, 28, 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)),        # strided: dgrad parity classes in one launch
    (32, 2, 7, 7, 256, 512, (3, 3, 3), (2, 2, 2), (1, 1, 1)),         # strided, few tiles: per-class launches
    (32, 16, 56, 56, 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # R(2+1)D: column segments
    (32, 16, 56, 56, 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    (16, 16, 224, 224, 4, 64, (1, 7, 7), (2, 2, 2), (0, 3, 3)),       # S3D-G stem (streaming stem kernel)
    (16, 8, 28, 28, 192, 96, (1, 1, 1), (1, 1, 1), (0, 0, 0)),        # 96-wide tile
    (16, 4, 14, 14, 160, 320, (1, 3, 3), (1, 1, 1), (0, 1, 1)),       # dgrad N = 160; forward 98 x 3 tiles: the 64-wide tile
    (32, 8, 28, 28, 64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),        # R3D-18 layer2.0 shortcut: 196 tiles, 64-wide
    (16, 4, 14, 14, 480, 400, (1, 1, 1), (1, 1, 1), (0, 0, 0)),       # S3D-G 14 x 14 pointwise trio: 64-wide tiles
    (16, 8, 28, 28, 192, 176, (1, 1, 1), (1, 1, 1), (0, 0, 0)),       # ... at 28 x 28: 784 x 2 tiles, the tile kernels
]


@pytest.mark.parametrize("case", NAME_CASES, ids=lambda c: "x".join(map(str, c[:6])) + f"k{c[6]}s{c[7]}")
def test_reported_kernel_name_is_the_kernel_that_ran(hip, case):
    """rsp_conv3d_kernel_name (what bench.py labels its roofline block and the traffic lookup with) re-derives the dispatch
    decision; the launchers record the instance they actually started (rsp_last_conv_kernel).  The two must agree for forward,
    dgrad and wgrad of every dispatch family."""
    import ctypes as C
    N, D, H, W, cin, cout, k, s, p = case
    g = ConvGeom(N, D, H, W, cin, cout, k, s, p)
    d = g.desc()
    lib = hip.lib
    want = [lib.rsp_conv3d_kernel_name(C.byref(d), which).decode() for which in (0, 1, 2)]
    x = torch.randn(N, D, H, W, cin, device=DEV)
    w = torch.randn(cout, cin, *k, device=DEV) * 0.05
    if cin == 4:       # an RGB stem: packed from its three real channels, as the engine's re-pack does (the name is that instance)
        ps0 = hip.pack_set([(g, 0, w[:, :3].contiguous())])
        ps0.run()
        wp = ps0.packed[0]
    else:
        wp = hip.conv_pack_fwd(g, w)
    y, _ = hip.conv_fwd(g, x, wp, None, True)
    assert lib.rsp_last_conv_kernel().decode() == want[0], ("fwd", want[0])
    dy = torch.randn_like(y)
    if cin > 4:
        ps = hip.pack_set([(g, 1, w)])
        ps.run()
        hip.conv_dgrad_packed(g, dy, ps.packed[0])
        assert lib.rsp_last_conv_kernel().decode() == want[1], ("dgrad", want[1])
    hip.conv_wgrad(g, x, dy, torch.empty_like(w))
    assert lib.rsp_last_conv_kernel().decode() == want[2], ("wgrad", want[2])


@pytest.mark.parametrize("case", [c for c in NAME_CASES if c[4] == 4 and c[6] != (7, 7, 7)], ids=lambda c: f"k{c[6]}s{c[7]}")
def test_stem_skips_the_padding_channel_of_rgb_filters(hip, case):
    """Filters re-packed from three input channels run the stem kernels' three-k-step instance (the zero fourth channel of the
    16-byte pixels is not multiplied); the result is that of the four-channel instance on zero-padded filters."""
    N, D, H, W, cin, cout, k, s, p = case
    N = 2
    g = ConvGeom(N, D, H, W, cin, cout, k, s, p)
    x = torch.randn(N, D, H, W, cin, device=DEV)
    x[..., 3] = 0
    w = torch.randn(cout, cin, *k, device=DEV) * 0.05
    w[:, 3] = 0
    bias = torch.randn(cout, device=DEV)
    ps = hip.pack_set([(g, 0, w[:, :3].contiguous())])
    ps.run()
    y3, st3 = hip.conv_fwd(g, x, ps.packed[0], bias, True)
    name3 = hip.lib.rsp_last_conv_kernel().decode()
    y4, st4 = hip.conv_fwd(g, x, hip.conv_pack_fwd(g, w), bias, True)
    name4 = hip.lib.rsp_last_conv_kernel().decode()
    assert name3.startswith("stem_") and name3.endswith(", 3>") and name4.endswith(", 4>"), (name3, name4)
    assert torch.equal(y3, y4) and torch.equal(st3, st4)


def test_packed_weights_are_shared_across_input_geometries(hip):
    """One packed copy per (conv, layout): the T_real values of diff_speed=[4,2,1], a partial batch or another clip size must
    not add copies (rspnet_amd/engine.py PackedWeights keys on the layout signature, not on N/D/H/W)."""
    from golden_util import load_spec
    from model_util import make_cfg
    from oracle import portable as P
    from rspnet_amd.moco import ModelFactory
    model = ModelFactory(make_cfg("c3d", 64)).build_moco_diffloss(device=DEV).module
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.fill_state(load_spec("c3d"), 5).items()})
    enc = model.encoder_q
    outs = {}
    for T, HW, B in ((16, 32, 4), (8, 32, 4), (16, 48, 2), (32, 32, 3)):
        x = torch.from_numpy(P.clips(5, 0, (B, 3, T, HW, HW))[0]).to(DEV)
        a, m = enc(x)
        outs[(T, HW, B)] = a
        n_entries = len(enc._packed._entries)
        if len(outs) == 1:
            first = n_entries
        assert n_entries == first, (T, HW, B, n_entries, first)
    # same clips again through the shared copies: identical result
    x = torch.from_numpy(P.clips(5, 0, (4, 3, 16, 32, 32))[0]).to(DEV)
    a2, _ = enc(x)
    assert torch.equal(a2, outs[(16, 32, 4)])


def test_channel_padded_unit_uses_the_parameters_own_lengths(hip):
    """R(2+1)D's odd mid-channel counts (83 / 230 / 921, models/r2plus1d_vcop.py:35-38) run zero-padded to a multiple of 4; the
    BatchNorm vectors and the weight gradient keep the parameter's own shape: rsp_bn_finalize_v / rsp_bn_act_pool_bwd_v /
    rsp_conv3d_wgrad_v treat the padding as gamma = beta = 0 and drop its gradients (no staging copies in the engine)."""
    N, D, H, W, Cin, Cv, Cp = 2, 4, 10, 10, 64, 83, 84
    g = ConvGeom(N, D, H, W, Cin, Cp, (1, 3, 3), (1, 1, 1), (0, 1, 1))
    x = rnd(N, D, H, W, Cin, seed=1)
    w = rnd(Cv, Cin, 1, 3, 3, seed=2, scale=0.1)
    wp = torch.zeros(Cp, Cin, 1, 3, 3)
    wp[:Cv] = w
    y_ref, st_ref = CPU.conv_fwd(g, x, wp, None, True)
    y, st = hip.conv_fwd(g, x.to(DEV), hip.conv_pack_fwd(g, wp.to(DEV)), None, True)
    close(y, y_ref, 2e-5, "padded conv")
    gamma, beta = rnd(Cv, seed=3) + 1.5, rnd(Cv, seed=4)
    rm, rv = rnd(Cv, seed=5), rnd(Cv, seed=6) + 1.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    mi_ref, ss_ref = CPU.bn_finalize(st_ref, g.rows, None, gamma, beta, 1e-5, 0.1, rm_ref, rv_ref)
    rm_d, rv_d = rm.to(DEV), rv.to(DEV)
    mi, ss = hip.bn_finalize(st, g.rows, None, gamma.to(DEV), beta.to(DEV), 1e-5, 0.1, rm_d, rv_d)
    close(ss, ss_ref, 1e-5, "scale/shift (padding: 0)")
    assert float(ss[:, Cv:].abs().max()) == 0.0
    close(rm_d, rm_ref, 1e-5, "running_mean")
    close(rv_d, rv_ref, 1e-5, "running_var")
    pg = PoolGeom(N, D, H, W, Cp)
    out = hip.bn_act_pool_fwd(pg, y, ss, None, True)
    assert float(out[..., Cv:].abs().max()) == 0.0
    dout = rnd(N, D, H, W, Cp, seed=7)
    dg_ref, db_ref = torch.empty(Cv), torch.empty(Cv)
    dy_ref, _ = CPU.bn_act_pool_bwd(pg, y_ref, None, dout, gamma, mi_ref, ss_ref, True, False, dg_ref, db_ref)
    dg, db = torch.empty(Cv, device=DEV), torch.empty(Cv, device=DEV)
    dy, _ = hip.bn_act_pool_bwd(pg, y, None, dout.to(DEV), gamma.to(DEV), mi, ss, True, False, dg, db)
    close(dy, dy_ref, 2e-5, "dy")
    close(dg, dg_ref, 2e-5, "dgamma")
    close(db, db_ref, 2e-5, "dbeta")
    assert float(dy[..., Cv:].abs().max()) == 0.0
    # weight gradient straight into the (83, 64, ...) parameter gradient; and the temporal partner whose INPUT channels are padded
    dw_ref = torch.empty_like(w)
    CPU.conv_wgrad(g, x, dy_ref, dw_ref)
    dw = torch.empty(Cv, Cin, 1, 3, 3, device=DEV)
    hip.conv_wgrad(g, x.to(DEV), dy, dw)
    close(dw, dw_ref, 2e-5, "dw (output channels padded)")
    g2 = ConvGeom(N, D, H, W, Cp, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0))
    dy2 = rnd(N, D, H, W, 64, seed=8)
    dw2_ref = torch.empty(64, Cv, 3, 1, 1)
    CPU.conv_wgrad(g2, out.cpu(), dy2, dw2_ref)
    dw2 = torch.empty(64, Cv, 3, 1, 1, device=DEV)
    hip.conv_wgrad(g2, out, dy2.to(DEV), dw2)
    close(dw2, dw2_ref, 2e-5, "dw (input channels padded)")


@pytest.mark.parametrize("C,Cv,tiles", [(64, 64, 5), (84, 83, 3), (1152, 1152, 2), (128, 128, 3000)])
def test_deferred_running_statistics_update(hip, C, Cv, tiles):
    """rsp_bn_finalize_x with batch_stats_out reports the pass's batch moments and leaves the running statistics alone;
    rsp_bn_running_update (one launch for a list of layers) then moves them exactly as the in-place finalize does."""
    rows = tiles * 128 - 17
    part = (torch.rand(tiles, C, 2) + 0.5) * 100
    part[:, :, 1] = part[:, :, 0] ** 2 / 128 * 1.3 + 5.0
    if Cv < C:
        part[:, Cv:] = 0
    gamma, beta = rnd(Cv, seed=3) + 1.5, rnd(Cv, seed=4)
    rm, rv = rnd(Cv, seed=5), rnd(Cv, seed=6) + 1.5
    other_rm, other_rv = rnd(40, seed=7), rnd(40, seed=8) + 1.5                 # a second layer in the same set
    d = lambda t: t.to(DEV)
    rm_a, rv_a = d(rm), d(rv)
    mi_a, ss_a = hip.bn_finalize(d(part), rows, None, d(gamma), d(beta), 1e-5, 0.1, rm_a, rv_a)
    rm_b, rv_b, orm, orv = d(rm), d(rv), d(other_rm), d(other_rv)
    ema = hip.bn_ema_set([(rm_b, rv_b, 0.1), (orm, orv, 0.25)])
    ema.stats[1].copy_(torch.stack([torch.full((40,), 2.0), torch.full((40,), 3.0)]))
    mi_b, ss_b = hip.bn_finalize(d(part), rows, None, d(gamma), d(beta), 1e-5, 0.1, None, None, batch_stats_out=ema.stats[0])
    assert torch.equal(rm_b.cpu(), rm) and torch.equal(rv_b.cpu(), rv)              # untouched so far
    assert torch.equal(mi_a, mi_b) and torch.equal(ss_a, ss_b)
    ema.run()
    close(rm_b, rm_a, 1e-6, "running_mean after the deferred update")
    close(rv_b, rv_a, 1e-6, "running_var after the deferred update")
    close(orm, 0.75 * other_rm + 0.25 * 2.0, 1e-6, "second layer mean")
    close(orv, 0.75 * other_rv + 0.25 * 3.0, 1e-6, "second layer var")


def test_virtual_pixel_stem_equals_padded_stem(hip):
    """R3D-18's and R(2+1)D's 3-channel stride-2 stems on virtual pixels (engine.VirtualStem: two kernel-width-6 stride-3
    convolutions over the packed 3-channel row) against the same unit as a 4-channel-padded convolution, forward, backward and
    after a weight update."""
    from virtual_stem_util import CASES, check_case
    for case in CASES + [(64, 0, (7, 7, 7), (1, 2, 2), (3, 3, 3), (2, 8, 56, 56))]:
        check_case(DEV, case, 3e-5)


@pytest.mark.parametrize("N,D,H,W,C,pool", [(2, 4, 12, 12, 64, None), (3, 2, 7, 7, 48, None), (2, 3, 9, 10, 6, None),
                                              (2, 4, 14, 14, 64, ((1, 3, 3), (1, 2, 2), (0, 1, 1))),
                                              (2, 4, 9, 9, 10, ((1, 3, 3), (1, 2, 2), (0, 1, 1)))])
def test_bn_act_gate_fused_is_bit_identical_to_the_three_ops(hip, N, D, H, W, C, pool):
    """ops.bn_act_gate_fwd (BatchNorm-apply + ReLU + S3D-G self-gating, optionally through the max-pool behind it) against
    bn_act_pool_fwd -> gate_fwd (-> maxpool_fwd): same bits, with and without the activation kept, into a channel slice too."""
    y = rnd(N, D, H, W, C, seed=11).to(DEV)
    ss = torch.stack([rnd(C, seed=12).abs() + 0.5, rnd(C, seed=13) * 0.3]).contiguous().to(DEV)
    w, b = (rnd(C, C, 1, 1, 1, seed=14) * 0.2).to(DEV), (rnd(C, seed=15) * 0.1).to(DEV)
    pg = PoolGeom(N, D, H, W, C)
    a_ref = hip.bn_act_pool_fwd(pg, y, ss, None, True)
    o_ref, mean_ref, gate_ref = hip.gate_fwd(a_ref, w, b)
    o, a, mean, gate = hip.bn_act_gate_fwd(pg, y, ss, True, w, b, True)
    assert torch.equal(a, a_ref) and torch.equal(mean, mean_ref) and torch.equal(gate, gate_ref) and torch.equal(o, o_ref)
    o2, a2, mean2, gate2 = hip.bn_act_gate_fwd(pg, y, ss, True, w, b, False)
    assert a2 is None and torch.equal(mean2, mean_ref) and torch.equal(gate2, gate_ref) and torch.equal(o2, o_ref)
    # into a channel slice of a wider (concat) tensor
    if C % 2 == 0:
        cat = torch.zeros(N, D, H, W, C + 8, device=DEV)
        hip.bn_act_gate_fwd(pg, y, ss, True, w, b, False, out=cat[..., 8:])
        assert torch.equal(cat[..., 8:], o_ref) and float(cat[..., :8].abs().max()) == 0.0
    if pool is not None:
        pp = PoolGeom(N, D, H, W, C, *pool)
        p_ref, _ = hip.maxpool_fwd(pp, o_ref, False)
        p3, _, _, _ = hip.bn_act_gate_fwd(pg, y, ss, True, w, b, False, pool=pp)
        assert torch.equal(p3, p_ref)
        # ... and for a forward a backward follows: the pool's arg-max from the same pass (first maximum in scan order)
        if hip.bn_act_gate_pool_idx_ok(pp, y, ss):
            assert C % 4 == 0
            p_k, i_k = hip.maxpool_fwd(pp, o_ref, True)
            p4, a4, m4, g4, i4 = hip.bn_act_gate_fwd(pg, y, ss, True, w, b, False, pool=pp, pool_idx=True)
            assert a4 is None and torch.equal(p4, p_k) and torch.equal(i4, i_k) and torch.equal(m4, mean_ref) and torch.equal(g4, gate_ref)
        else:
            assert C % 4 != 0
    # and the checker agrees to rounding
    oc, ac, mc, gc = CPU.bn_act_gate_fwd(pg, y.cpu(), ss.cpu(), True, w.cpu(), b.cpu(), True)
    close(o, oc, 2e-5, "gated output")
    close(gate, gc, 2e-5, "gate")


@pytest.mark.parametrize("N,D,H,W,C,k,s,p", [(2, 4, 12, 12, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1)), (1, 5, 9, 11, 20, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                             (2, 3, 8, 8, 6, (1, 3, 3), (1, 2, 2), (0, 1, 1))])
def test_bn_act_pool_fwd_overlapping_window_equals_apply_then_maxpool(hip, N, D, H, W, C, k, s, p):
    """The ResNet stems' bn1 -> relu -> MaxPool3d(3, 2, 1) (models/resnet.py:203-207) as ONE pass over the convolution output
    (engine._pool_fusion, forwards that keep nothing): bn_act_pool_fwd with the overlapping, padded window against the unit-window
    apply followed by maxpool_fwd — same bits; and against the checker."""
    y = rnd(N, D, H, W, C, seed=61).to(DEV)
    ss = torch.stack([rnd(C, seed=62).abs() + 0.5, rnd(C, seed=63) * 0.3]).contiguous().to(DEV)
    a = hip.bn_act_pool_fwd(PoolGeom(N, D, H, W, C), y, ss, None, True)
    pg = PoolGeom(N, D, H, W, C, k, s, p)
    want, _ = hip.maxpool_fwd(pg, a, False)
    got = hip.bn_act_pool_fwd(pg, y, ss, None, True)
    assert got.shape == want.shape and torch.equal(got, want)
    # ... and with the arg-max a backward needs (the kept forward): same values, same first-maximum indices
    want_k, idx_k = hip.maxpool_fwd(pg, a, True)
    fused = hip.bn_act_maxpool_fwd(pg, y, ss, True, True)
    if C % 4 == 0:
        assert fused is not None and torch.equal(fused[0], want_k) and torch.equal(fused[1], idx_k)
        assert hip.bn_act_maxpool_fwd(pg, y, ss, True, False)[1] is None
    else:
        assert fused is None                      # (not covered: the caller runs the two ops)
    close(got, CPU.bn_act_pool_fwd(pg, y.cpu(), ss.cpu(), None, True), 2e-6, "fused apply + overlapping max-pool")


@pytest.mark.parametrize("N,D,H,W,C,sliced", [(2, 4, 12, 12, 64, False), (3, 2, 7, 7, 48, True), (2, 3, 9, 10, 6, False)])
def test_bn_act_gate_bwd_fused_matches_the_two_ops(hip, N, D, H, W, C, sliced):
    """ops.bn_act_gate_bwd (activation recomputed from y, the gate's data gradient formed inside the BatchNorm backward kernels)
    against gate_bwd on the stored activation followed by bn_act_pool_bwd — on the device and against the checker."""
    y = rnd(N, D, H, W, C, seed=21).to(DEV)
    gamma = (rnd(C, seed=22).abs() + 0.5).to(DEV)
    mean_c, var_c = y.mean(dim=(0, 1, 2, 3)), y.var(dim=(0, 1, 2, 3), unbiased=False)
    invstd = 1.0 / torch.sqrt(var_c + 1e-3)
    beta = (rnd(C, seed=23) * 0.3).to(DEV)
    mi = torch.stack([mean_c, invstd]).contiguous()
    ss = torch.stack([gamma * invstd, beta - mean_c * gamma * invstd]).contiguous()
    w, b = (rnd(C, C, 1, 1, 1, seed=24) * 0.2).to(DEV), (rnd(C, seed=25) * 0.1).to(DEV)
    pg = PoolGeom(N, D, H, W, C)
    o, a, mean, gate = hip.bn_act_gate_fwd(pg, y, ss, True, w, b, True)
    if sliced:      # gradient of the gated output as a channel slice of a wider concat gradient
        wide = rnd(N, D, H, W, C + 16, seed=26).to(DEV)
        dout = wide[..., 8:8 + C]
    else:
        dout = rnd(N, D, H, W, C, seed=26).to(DEV)
    dw_r, db_r, dg_r, dbt_r = (torch.empty_like(t) for t in (w, b, gamma, beta))
    dx = hip.gate_bwd(a, dout, w, mean, gate, dw_r, db_r)
    dy_r, _ = hip.bn_act_pool_bwd(pg, y, None, dx, gamma, mi, ss, True, False, dg_r, dbt_r)
    dw, db, dg, dbt = (torch.empty_like(t) for t in (w, b, gamma, beta))
    dy = hip.bn_act_gate_bwd(pg, y, dout, gamma, mi, ss, True, w, mean, gate, dg, dbt, dw, db)
    for name, got, ref in (("dy", dy, dy_r), ("dw", dw, dw_r), ("db", db, db_r), ("dgamma", dg, dg_r), ("dbeta", dbt, dbt_r)):
        close(got, ref.cpu(), 1e-6, name)
    dwc, dbc, dgc, dbtc = (torch.empty(t.shape) for t in (w, b, gamma, beta))
    dyc = CPU.bn_act_gate_bwd(pg, y.cpu(), dout.cpu(), gamma.cpu(), mi.cpu(), ss.cpu(), True, w.cpu(), mean.cpu(), gate.cpu(), dgc, dbtc,
                              dwc, dbc)
    for name, got, ref in (("dy", dy, dyc), ("dw", dw, dwc), ("dgamma", dg, dgc), ("dbeta", dbt, dbtc)):
        close(got, ref, 5e-5, name + " vs checker")


def test_conv_cases_also_pass_on_the_direct_kernel():
    """igemm_direct_kernel (operands straight from L2, no LDS tiles) is off by default since the 64-wide tiles took over the small
    launches (conv_igemm.hip: direct_applies); it stays selectable for re-measurements, so it stays correct: the convolution cases
    in a child interpreter with RSP_DIRECT_MAX_TILES=448 — every launch of at most 448 tiles with Cin % 16 == 0 then runs on it (the
    switch is read once per process)."""
    import os
    import subprocess
    import sys
    if os.environ.get("RSP_DIRECT_MAX_TILES"):
        pytest.skip("already the direct-kernel run")
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, RSP_DIRECT_MAX_TILES="448")
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(here, "test_kernels_gpu.py"), "-x", "-q", "-m", "gpu", "-k",
                        "conv_fwd_dgrad_wgrad or conv_fuzz or deterministic or reported_kernel_name"], capture_output=True, text=True,
                       timeout=1800, env=env, cwd=os.path.dirname(here))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-1000:]


def test_rowgeom_cache_is_bounded_and_ordered(hip):
    """ops.HipOps keeps the weight-gradient kernels' row-geometry tables per geometry: a table filled on one stream is
    handed to a weight gradient on ANOTHER stream only behind the fill's event, and the cache is bounded — least recently used
    tables go, results stay right."""
    saved = (hip.ROWGEOM_MAX_TABLES, dict(hip._rowgeom), hip._rowgeom_bytes)
    hip._rowgeom.clear()
    hip._rowgeom_bytes = 0
    hip.ROWGEOM_MAX_TABLES = 2
    try:
        side = torch.cuda.Stream()
        geoms = [ConvGeom(2, 4, 8 + i, 8, 16, 32, (3, 3, 3), (1, 1, 1), (1, 1, 1)) for i in range(4)]
        for rnd_i, g in enumerate(geoms + geoms[:2]):
            x = rnd(g.N, g.Di, g.Hi, g.Wi, g.Cin, seed=20 + rnd_i)
            do, ho, wo = g.out_dims
            dy = rnd(g.N, do, ho, wo, g.Cout, seed=40 + rnd_i)
            dw_ref = torch.empty(g.Cout, g.Cin, *g.k)
            CPU.conv_wgrad(g, x, dy, dw_ref, None)
            xd, dyd = x.to(DEV), dy.to(DEV)
            # first use on the side stream (where the engine's weight-gradient tasks run), then at once on the main stream with
            # another Cout: the cached table must not be read before its fill has finished
            dw_side = torch.empty_like(dw_ref, device=DEV)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                hip.conv_wgrad(g, xd, dyd, dw_side)
            g2 = ConvGeom(g.N, g.Di, g.Hi, g.Wi, g.Cin, 16, g.k, g.s, g.p)
            dw2 = torch.empty(16, g.Cin, *g.k, device=DEV)
            hip.conv_wgrad(g2, xd, dyd[..., :16].contiguous(), dw2)
            torch.cuda.current_stream().wait_stream(side)
            close(dw_side, dw_ref, 2e-5, "wgrad on the side stream")
            close(dw2, dw_ref[:16], 2e-5, "wgrad through the cached table on the main stream")
            assert len(hip._rowgeom) <= 2
    finally:
        torch.cuda.synchronize()
        hip._rowgeom.clear()
        hip.ROWGEOM_MAX_TABLES = saved[0]
        hip._rowgeom.update(saved[1])
        hip._rowgeom_bytes = saved[2]
