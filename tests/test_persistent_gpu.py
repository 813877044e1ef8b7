"""GPU: the persistent implicit-GEMM kernels (round 4: resident workgroups walking (tile, K-slice) units, next unit's first chunk copied
under the last MFMAs, raw-buffer epilogue stores) produce BIT-IDENTICAL outputs to the per-tile kernels they replace — same tiles, same
K walk, same accumulation order — for forward (+ BatchNorm partials) and input gradient over the tile widths, both K orders, strided
classes, depth-major rows, K tails, column segments and K-split tails.  The per-tile kernels run in a child interpreter with
RSP_NO_PERSIST=1 (the switch is read once per process).  Reference call sites: nn.Conv3d forward / backward-input of every backbone
(models/c3d.py:21-52, models/resnet.py:48-77, models/r2plus1d_vcop.py:49-67, models/s3dg.py:36-52)."""
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CASES = {
    "bn32_tm": (2, 4, 12, 12, 32, 24, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "bn64_ks_depth_major": (4, 8, 28, 28, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "bn64_strided": (2, 4, 12, 12, 64, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
    "bn128_segment_ktail": (3, 3, 18, 21, 8, 130, (3, 1, 7), (1, 1, 2), (1, 0, 2)),
    "bn96_strided_ktail": (2, 3, 18, 10, 20, 96, (3, 3, 1), (2, 2, 1), (1, 1, 0)),
    "piecewise_short_frames": (16, 4, 19, 11, 128, 32, (3, 1, 3), (1, 1, 1), (1, 0, 0)),
    "one_chunk": (8, 3, 7, 21, 4, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    "bn160": (4, 4, 14, 14, 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "bn160_dgrad": (4, 4, 14, 14, 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    "bn160_ks_depth_major": (3, 4, 7, 7, 128, 140, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "ksplit_tail": (32, 2, 7, 7, 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "many_units_per_workgroup": (32, 8, 56, 56, 32, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "ksplit_r3d_layer2": (32, 4, 14, 14, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    "ksplit_s3dg_pointwise": (16, 4, 14, 14, 480, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    "ksplit_s3dg_sep": (16, 4, 14, 14, 160, 320, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "ksplit_c3d_conv5": (32, 2, 7, 7, 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # 129..144 columns at sizes above the tiny-launch plan (round 6: launches of <= 256 units of 128 x 64 run on the 32-wide tile)
    "bn144_big": (8, 4, 28, 28, 64, 144, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "bn144_dgrad_big": (8, 4, 28, 28, 144, 64, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    "bn144_ks_depth_major_big": (16, 4, 14, 14, 128, 140, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # the 256 x 64 instance (round 6) against the per-tile 128 x 64 kernel: K short enough that neither plan splits it
    "tall_fwd_short_k": (4, 8, 112, 112, 16, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    "tall_dgrad_short_k": (4, 8, 112, 112, 64, 8, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
}


def _child(path):
    sys.path.insert(0, ROOT)
    from rspnet_amd import ops
    from rspnet_amd.ops import ConvGeom
    be = ops.backend()
    dev = torch.device("cuda", 0)
    out = {}
    for name, (N, D, H, W, cin, cout, k, s, p) in CASES.items():
        g = ConvGeom(N, D, H, W, cin, cout, k, s, p)
        gen = torch.Generator(device=dev).manual_seed(7)
        x = torch.randn(N, D, H, W, cin, device=dev, generator=gen)
        w = torch.randn(cout, cin, *k, device=dev, generator=gen) * 0.05
        b = torch.randn(cout, device=dev, generator=gen)
        dy = torch.randn(N, *g.out_dims, cout, device=dev, generator=gen)
        torch.full((1 << 22,), 7.0, device=dev)                     # stale memory is recognisable
        y, st = be.conv_fwd(g, x, be.conv_pack_fwd(g, w), b, True)
        kf = be.lib.rsp_last_conv_kernel().decode()
        dx = be.conv_dgrad(g, dy, w)
        out[name] = (y.cpu(), st.cpu(), dx.cpu(), kf, be.lib.rsp_last_conv_kernel().decode())
    torch.save(out, path)


def test_persistent_kernels_equal_the_per_tile_kernels_bit_for_bit(tmp_path):
    """... and the 144-wide instance (segments of 129..144 columns: a 16-wide fifth column block on the 16x16x4 MFMA, round 5) equals the
    160-wide one bit for bit in its four 32-wide blocks and to rounding in the half block, whose k are summed in another order."""
    res = {}
    for mode in ("persistent", "per_tile", "half"):
        env = dict(os.environ)
        env.pop("RSP_NO_PERSIST", None)
        env.pop("RSP_NO_HALF_BLOCK", None)
        env["RSP_TALL_MIN_TILES"] = "768"           # (the 256 x 64 instance is off by default: selected here for its bit-identity cases)
        if mode == "per_tile":
            env["RSP_NO_PERSIST"] = "1"
        if mode == "persistent":
            env["RSP_NO_HALF_BLOCK"] = "1"      # the per-tile kernels' tiles: 160-wide
        f = str(tmp_path / f"{mode}.pt")
        subprocess.run([sys.executable, "-c", f"import sys; sys.path.insert(0, {ROOT!r}); sys.path.insert(0, {os.path.join(ROOT, 'tests')!r}); "
                                              f"import test_persistent_gpu as t; t._child({f!r})"], env=env, check=True, cwd=ROOT)
        res[mode] = torch.load(f)
    ran_persistent = 0
    for name in CASES:
        yp, sp, dp, kfp, kdp = res["persistent"][name]
        yc, sc, dc, kfc, kdc = res["per_tile"][name]
        assert "persist" not in kfc and "persist" not in kdc, (name, kfc, kdc)
        ran_persistent += ("persist" in kfp) + ("persist" in kdp)
        if name == "many_units_per_workgroup":
            # the 256-row instance cuts ITS ragged last round along K (its own plan): same products, another order in those tiles
            assert "<256, 64," in kfp, kfp
            assert float((yp - yc).abs().max()) <= 2e-6 * float(yc.abs().max()), (name, "forward", kfp)
            assert torch.equal(dp, dc), (name, "input gradient", kdp)
            continue
        assert torch.equal(yp, yc), (name, "forward", kfp, float((yp - yc).abs().max()))
        assert torch.equal(sp, sc), (name, "BatchNorm partials", kfp)
        assert torch.equal(dp, dc), (name, "input gradient", kdp, float((dp - dc).abs().max()))
    assert "<256, 64," in res["persistent"]["tall_fwd_short_k"][3] and "<256, 64," in res["persistent"]["tall_dgrad_short_k"][4]
    assert ran_persistent >= 16, ran_persistent      # (the long tap-major 128-wide launches and multi-class dgrads stay per-tile)
    ran_half = 0
    for name in CASES:
        yp, sp, dp, kfp, kdp = res["persistent"][name]
        yh, sh, dh, kfh, kdh = res["half"][name]
        for what, a, b, kern in (("forward", yp, yh, kfh), ("BatchNorm partials", sp.transpose(1, 2), sh.transpose(1, 2), kfh), ("input gradient", dp, dh, kdh)):
            if "<128, 144," not in kern:
                assert torch.equal(a, b), (name, what, kern)
                continue
            ran_half += 1
            assert a.shape[-1] > 128 and torch.equal(a[..., :128], b[..., :128]), (name, what, "32-wide blocks")
            tol = 2e-6 * float(a.abs().max())
            assert float((a[..., 128:] - b[..., 128:]).abs().max()) <= tol, (name, what, "half block")
    assert ran_half >= 7, ran_half


def test_k_split_layers_are_run_to_run_identical():
    """K-split tail tiles (their slices summed in fixed order by splitk_reduce_vec_kernel) through the persistent kernels: 60 launches of
    R3D-18 layers 2 / 3, an S3D-G separable unit and C3D conv5 must give the same bits every time, forward and input gradient."""
    sys.path.insert(0, ROOT)
    from rspnet_amd import ops
    from rspnet_amd.ops import ConvGeom
    be = ops.backend()
    dev = torch.device("cuda", 0)
    for name in ("ksplit_tail", "ksplit_s3dg_sep", "ksplit_c3d_conv5", "ksplit_r3d_layer2"):
        N, D, H, W, cin, cout, k, s, p = CASES[name]
        g = ConvGeom(N, D, H, W, cin, cout, k, s, p)
        gen = torch.Generator(device=dev).manual_seed(11)
        x = torch.randn(N, D, H, W, cin, device=dev, generator=gen)
        w = torch.randn(cout, cin, *k, device=dev, generator=gen) * 0.05
        dy = torch.randn(N, *g.out_dims, cout, device=dev, generator=gen)
        wp = be.conv_pack_fwd(g, w)
        y0, st0 = be.conv_fwd(g, x, wp, None, True)
        dx0 = be.conv_dgrad(g, dy, w)
        for it in range(60):
            torch.full((1 << 20,), float(it), device=dev)            # churn the allocator: fresh workspace contents
            y, st = be.conv_fwd(g, x, wp, None, True)
            dx = be.conv_dgrad(g, dy, w)
            assert torch.equal(y, y0) and torch.equal(st, st0) and torch.equal(dx, dx0), (name, it)
