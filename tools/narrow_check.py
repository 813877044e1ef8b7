#!/usr/bin/env python
"""Do the 64-wide tiles (narrow_tiles) change a convolution's output by more than summation-order noise?  The same forward (+ BN
statistic partials) and input gradient in two child interpreters — default and RSP_NARROW_MAX_TILES=0 — compared against an fp64
reference on the CPU: both must sit at the same distance from it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

CASES = [(2, 4, 10, 10, 128, 288, (1, 3, 3)), (4, 2, 7, 7, 256, 256, (3, 3, 3)), (2, 4, 14, 14, 480, 400, (1, 1, 1)),
         (4, 4, 8, 8, 64, 128, (3, 3, 3)), (4, 1, 4, 4, 512, 512, (3, 3, 3))]
if len(sys.argv) > 1 and sys.argv[1] == "child":
    from rspnet_amd import ops
    from rspnet_amd.ops import ConvGeom
    be = ops.backend()
    dev = torch.device("cuda", 0)
    out = {}
    for i, (N, D, H, W, cin, cout, k) in enumerate(CASES):
        g0 = torch.Generator().manual_seed(100 + i)
        x = torch.randn(N, D, H, W, cin, generator=g0)
        w = torch.randn(cout, cin, *k, generator=g0) * (cin * k[0] * k[1] * k[2]) ** -0.5
        p = tuple(a // 2 for a in k)
        g = ConvGeom(N, D, H, W, cin, cout, k, (1, 1, 1), p)
        y, st = be.conv_fwd(g, x.to(dev), be.conv_pack_fwd(g, w.to(dev)), None, True)
        dy = torch.randn(y.shape, generator=g0)
        dx = be.conv_dgrad(g, dy.to(dev), w.to(dev))
        out[i] = (y.cpu(), st.double().sum(0).cpu(), dx.cpu(), be.lib.rsp_last_conv_kernel().decode())
    torch.save(out, sys.argv[2])
    sys.exit(0)
res = {}
for tag, env in (("narrow", {}), ("wide", {"RSP_NARROW_MAX_TILES": "0"})):
    path = f"/tmp/narrow_check_{tag}.pt"
    subprocess.run([sys.executable, os.path.abspath(__file__), "child", path], check=True, env=dict(os.environ, **env))
    res[tag] = torch.load(path)
for i, (N, D, H, W, cin, cout, k) in enumerate(CASES):
    g0 = torch.Generator().manual_seed(100 + i)
    x = torch.randn(N, D, H, W, cin, generator=g0)
    w = torch.randn(cout, cin, *k, generator=g0) * (cin * k[0] * k[1] * k[2]) ** -0.5
    p = tuple(a // 2 for a in k)
    y64 = torch.nn.functional.conv3d(x.permute(0, 4, 1, 2, 3).double(), w.double(), padding=p).permute(0, 2, 3, 4, 1)
    s64 = torch.stack([y64.reshape(-1, cout).sum(0), (y64 ** 2).reshape(-1, cout).sum(0)], 1)

    def rms(a, b):
        return float(((a.double() - b) ** 2).mean().sqrt() / (b ** 2).mean().sqrt())

    line = f"{N}x{D}x{H}x{W}x{cin}->{cout} k{k}:"
    for tag in ("narrow", "wide"):
        y, st, dx, kern = res[tag][i]
        line += f"  {tag}: y {rms(y, y64):.2e} sum {float(((st[:, 0] - s64[:, 0]).abs() / s64[:, 1].sqrt()).max()):.2e} sumsq {rms(st[:, 1], s64[:, 1]):.2e} [{kern[-24:]}]"
    yn, yw = res["narrow"][i][0], res["wide"][i][0]
    line += f"  | narrow vs wide: y {rms(yn, yw.double()):.2e} dx {rms(res['narrow'][i][2], res['wide'][i][2].double()):.2e}"
    print(line)
