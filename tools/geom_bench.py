#!/usr/bin/env python
"""Forward time of a list of convolution geometries on the library RSPNET_HIP_LIB points at (default: the product build): 60
back-to-back launches each (sustained clocks), HIP events.  For A/B runs of variant builds (tools/build_variant.sh) on the layers a
change is aimed at:  python tools/geom_bench.py [s3dg14|r3d|all] [fwd|dgrad|wgrad] [--opt NAME=v1,v2,...]
--opt: every geometry once per value of a launcher planning option (rsp_conv3d_set_option: narrow_max_tiles, narrow32_max_units,
tall_min_tiles), side by side in one process."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom

SETS = {
    "s3dg14": [(16, 4, 14, 14, 528, 448, (1, 1, 1)), (16, 4, 14, 14, 512, 288, (1, 1, 1)), (16, 4, 14, 14, 480, 304, (1, 1, 1)),
               (16, 4, 14, 14, 160, 320, (1, 3, 3)), (16, 4, 14, 14, 144, 288, (1, 3, 3)), (16, 4, 14, 14, 320, 320, (3, 1, 1)),
               (16, 4, 14, 14, 288, 288, (3, 1, 1)), (16, 4, 14, 14, 128, 256, (1, 3, 3)), (16, 4, 14, 14, 256, 256, (3, 1, 1)),
               (16, 4, 14, 14, 112, 224, (1, 3, 3)), (16, 4, 14, 14, 224, 224, (3, 1, 1)), (16, 4, 14, 14, 96, 208, (1, 3, 3)),
               (16, 4, 14, 14, 208, 208, (3, 1, 1)), (16, 4, 14, 14, 512, 64, (1, 1, 1))],
    "s3dg28": [(16, 8, 28, 28, 256, 288, (1, 1, 1)), (16, 8, 28, 28, 192, 176, (1, 1, 1)), (16, 8, 28, 28, 128, 192, (1, 3, 3)),
               (16, 8, 28, 28, 192, 192, (3, 1, 1)), (16, 8, 28, 28, 96, 128, (1, 3, 3)), (16, 8, 28, 28, 128, 128, (3, 1, 1))],
    "small": [(16, 2, 7, 7, 832, 448, (1, 1, 1)), (16, 2, 7, 7, 832, 128, (1, 1, 1)), (16, 2, 7, 7, 320, 320, (3, 1, 1)),
              (16, 2, 7, 7, 160, 320, (1, 3, 3)), (16, 2, 7, 7, 128, 128, (3, 1, 1)), (16, 2, 7, 7, 384, 384, (3, 1, 1)),
              (16, 2, 7, 7, 192, 384, (1, 3, 3)), (16, 2, 7, 7, 832, 624, (1, 1, 1)),
              (32, 8, 28, 28, 64, 128, (1, 1, 1), (2, 2, 2)), (32, 4, 14, 14, 128, 256, (1, 1, 1), (2, 2, 2)),
              (32, 2, 7, 7, 256, 512, (1, 1, 1), (2, 2, 2))],
    "r3d": [(32, 4, 14, 14, 128, 128, (3, 3, 3)), (32, 2, 7, 7, 256, 256, (3, 3, 3)), (32, 1, 4, 4, 512, 512, (3, 3, 3))],
    "wg": [(32, 16, 56, 56, 64, 144, (1, 3, 3)), (32, 16, 56, 56, 64, 128, (3, 3, 3)), (32, 8, 28, 28, 128, 288, (1, 3, 3)),
           (32, 16, 56, 56, 144, 64, (3, 1, 1)), (32, 8, 28, 28, 128, 256, (3, 3, 3)), (32, 4, 14, 14, 256, 576, (1, 3, 3)),
           (16, 8, 56, 56, 64, 192, (1, 3, 3)), (16, 8, 56, 56, 192, 192, (3, 1, 1)), (16, 8, 28, 28, 128, 192, (1, 3, 3)),
           (32, 16, 56, 56, 64, 232, (1, 3, 3), (1, 2, 2)), (32, 8, 28, 28, 64, 64, (3, 3, 3))],
    "stem": [(32, 16, 112, 112, 4, 64, (3, 3, 3))],
    # 33..64-column launches of at least a round of 256-row tiles (tall_tiles): R3D-18 layer1 and its virtual-pixel stem class,
    # R(2+1)D's temporal halves (144 -> 64, 84 -> 64), S3D-G's (7,1,1) and 56 x 56 pointwise layers
    "tall": [(32, 8, 28, 28, 64, 64, (3, 3, 3)), (32, 16, 112, 89, 4, 64, (7, 7, 6), (1, 2, 3), (3, 3, 0)), (32, 16, 56, 56, 144, 64, (3, 1, 1)),
             (32, 16, 56, 56, 84, 64, (3, 1, 1)), (16, 8, 112, 112, 64, 64, (7, 1, 1)), (16, 8, 56, 56, 64, 64, (1, 1, 1)),
             (32, 8, 28, 28, 64, 128, (3, 3, 3), (2, 2, 2))],
    # less than one 64-wide unit per CU (narrow_bn -> 32-wide tiles instead of a K split)
    "tiny": [(16, 4, 14, 14, 512, 64, (1, 1, 1)), (16, 4, 14, 14, 64, 64, (3, 1, 1)), (16, 4, 14, 14, 480, 64, (1, 1, 1)), (16, 4, 14, 14, 16, 48, (1, 3, 3)),
             (16, 4, 14, 14, 48, 48, (3, 1, 1)), (16, 4, 14, 14, 24, 64, (1, 3, 3)), (16, 4, 14, 14, 512, 128, (1, 1, 1)), (16, 4, 14, 14, 528, 128, (1, 1, 1)),
             (16, 2, 7, 7, 832, 128, (1, 1, 1)), (16, 2, 7, 7, 128, 128, (3, 1, 1)), (16, 2, 7, 7, 832, 448, (1, 1, 1)), (16, 2, 7, 7, 320, 320, (3, 1, 1)),
             (16, 2, 7, 7, 384, 384, (3, 1, 1)), (16, 2, 7, 7, 192, 384, (1, 3, 3)), (16, 2, 7, 7, 832, 624, (1, 1, 1)),
             (32, 2, 7, 7, 256, 256, (3, 3, 3)), (32, 1, 4, 4, 512, 512, (3, 3, 3)), (32, 4, 14, 14, 128, 256, (1, 1, 1), (2, 2, 2)),
             (32, 2, 7, 7, 256, 512, (1, 1, 1), (2, 2, 2))],
    "r21d": [(32, 4, 14, 14, 256, 576, (1, 3, 3)), (32, 4, 14, 14, 576, 256, (3, 1, 1)), (32, 2, 7, 7, 512, 1152, (1, 3, 3)),
             (32, 2, 7, 7, 1152, 512, (3, 1, 1))],
}
argv = [a for a in sys.argv[1:] if not a.startswith("--opt")]
opt = None
for i, a in enumerate(sys.argv):
    if a == "--opt":
        name, vals = sys.argv[i + 1].split("=")
        opt = (name, [int(v) for v in vals.split(",")])
        argv = [x for x in argv if x != sys.argv[i + 1]]
which = argv[0] if len(argv) > 0 else "all"
mode = argv[1] if len(argv) > 1 else "fwd"      # fwd | wgrad | dgrad
cases = sum(SETS.values(), []) if which == "all" else SETS[which]
be = ops.backend()
dev = torch.device("cuda", 0)
print("lib:", os.environ.get("RSPNET_HIP_LIB", "product"), "mode:", mode, "option:", opt)
tot = {}
for case in cases:
    N, D, H, W, cin, cout, k = case[:7]
    st = case[7] if len(case) > 7 else (1, 1, 1)
    p = case[8] if len(case) > 8 else tuple(x // 2 for x in k)
    g = ConvGeom(N, D, H, W, cin, cout, k, st, p)
    x = torch.randn(N, D, H, W, cin, device=dev)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    wp = be.conv_pack_fwd(g, w)
    if mode == "fwd":
        run = lambda: be.conv_fwd(g, x, wp, None, True)
    else:
        do, ho, wo = g.out_dims
        dy = torch.randn(N, do, ho, wo, cout, device=dev)
        dw = torch.empty_like(w)
        run = (lambda: be.conv_wgrad(g, x, dy, dw, None)) if mode == "wgrad" else (lambda: be.conv_dgrad(g, dy, w))
    line = f"{N}x{D}x{H}x{W}x{cin}->{cout} k{k} s{st}:"
    for val in (opt[1] if opt else [None]):
        prev = be.set_option(opt[0], val) if opt else None
        try:
            for _ in range(10):
                run()
            kern = be.lib.rsp_last_conv_kernel().decode().replace("igemm_persist_kernel", "P").replace("igemm_kernel", "T")
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(60):
                run()
            e1.record()
            torch.cuda.synchronize()
        finally:
            if opt:
                be.set_option(opt[0], -1)
        us = e0.elapsed_time(e1) / 60 * 1e3
        tot[val] = tot.get(val, 0.0) + us
        line += f"  [{val}] {us:8.1f} us {g.flops / us / 1e6:6.1f} TF {kern}"
    print(line)
print("sum us:", {k: round(v, 1) for k, v in tot.items()})
