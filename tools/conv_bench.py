#!/usr/bin/env python
"""Per-layer conv micro-benchmark (C3D shapes at B=32 by default): TFLOP/s of fwd / dgrad / wgrad launches.
Usage: python tools/conv_bench.py [--r21d | --s3dg | --r3d | --stems] [--layers a,b] [--what fwd,dgrad,wgrad]."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--what", default="fwd,dgrad,wgrad")
ap.add_argument("--stems", action="store_true", help="the four backbone stems (Cin padded to 4) instead of the C3D stack")
ap.add_argument("--r3d", action="store_true", help="R3D-18 residual-stage conv shapes")
ap.add_argument("--r21d", action="store_true", help="R(2+1)D factored conv shapes (mid channels zero-padded to a multiple of 4)")
ap.add_argument("--s3dg", action="store_true", help="S3D-G conv shapes (B=16 by default for this list)")
ap.add_argument("--layers", default="", help="comma-separated layer names to keep")
args = ap.parse_args()

import torch
from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom

B = args.batch
LAYERS = [("conv1", 16, 112, 3, 64), ("conv1p4", 16, 112, 4, 64), ("conv2", 16, 56, 64, 128), ("conv3a", 8, 28, 128, 256), ("conv3b", 8, 28, 256, 256),
          ("conv4a", 4, 14, 256, 512), ("conv4b", 4, 14, 512, 512), ("conv5a", 2, 7, 512, 512)]
be = ops.backend()
dev = torch.device("cuda", 0)


def timeit(fn):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / args.iters


STEMS = [("c3d-stem", 16, 112, 4, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("r3d-stem", 16, 112, 4, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),
         ("r21d-stem", 16, 112, 4, 45, (1, 7, 7), (1, 2, 2), (0, 3, 3)), ("s3dg-stem", 16, 224, 4, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3))]
R3D = [("r3d-l1", 8, 28, 64, 64), ("r3d-l2", 4, 14, 128, 128), ("r3d-l3", 2, 7, 256, 256), ("r3d-l4", 1, 4, 512, 512),
       ("r3d-l2s2", 8, 28, 64, 128, (3, 3, 3), (2, 2, 2), (1, 1, 1)), ("r3d-l4s2", 2, 7, 256, 512, (3, 3, 3), (2, 2, 2), (1, 1, 1))]
S_ = lambda k: ((1, k, k), (1, 1, 1), (0, k // 2, k // 2))
T_ = lambda k, st=1: ((k, 1, 1), (st, 1, 1), (k // 2, 0, 0))
R21D = [("c1.sp", 16, 112, 4, 84, (1, 7, 7), (1, 2, 2), (0, 3, 3)), ("c1.tm", 16, 56, 84, 64, *T_(3)),
        ("c2.sp", 16, 56, 64, 144, *S_(3)), ("c2.tm", 16, 56, 144, 64, *T_(3)),
        ("c3a.sp", 16, 56, 64, 232, (1, 3, 3), (1, 2, 2), (0, 1, 1)), ("c3a.tm", 16, 28, 232, 128, *T_(3, 2)),
        ("c3b.sp", 8, 28, 128, 288, *S_(3)), ("c3b.tm", 8, 28, 288, 128, *T_(3)),
        ("c4a.sp", 8, 28, 128, 460, (1, 3, 3), (1, 2, 2), (0, 1, 1)), ("c4a.tm", 8, 14, 460, 256, *T_(3, 2)),
        ("c4b.sp", 4, 14, 256, 576, *S_(3)), ("c4b.tm", 4, 14, 576, 256, *T_(3)),
        ("c5a.sp", 4, 14, 256, 924, (1, 3, 3), (1, 2, 2), (0, 1, 1)), ("c5a.tm", 4, 7, 924, 512, *T_(3, 2)),
        ("c5b.sp", 2, 7, 512, 1152, *S_(3)), ("c5b.tm", 2, 7, 1152, 512, *T_(3))]
P_ = ((1, 1, 1), (1, 1, 1), (0, 0, 0))
S3DG = [("stem.tm", 8, 112, 64, 64, *T_(7)), ("basic", 8, 56, 64, 64, *P_), ("sc2.sp", 8, 56, 64, 192, *S_(3)), ("sc2.tm", 8, 56, 192, 192, *T_(3)),
        ("3b.b0", 8, 28, 192, 64, *P_), ("3b.b1", 8, 28, 192, 96, *P_), ("3b.b1s", 8, 28, 96, 128, *S_(3)), ("3b.b1t", 8, 28, 128, 128, *T_(3)),
        ("3c.b1", 8, 28, 256, 128, *P_), ("3c.b1s", 8, 28, 128, 192, *S_(3)), ("3c.b1t", 8, 28, 192, 192, *T_(3)),
        ("4b.b0", 4, 14, 480, 192, *P_), ("4b.b1s", 4, 14, 96, 208, *S_(3)), ("4b.b1t", 4, 14, 208, 208, *T_(3)), ("4b.b2", 4, 14, 480, 16, *P_),
        ("4f.b0", 4, 14, 528, 256, *P_), ("4f.b1s", 4, 14, 160, 320, *S_(3)), ("4f.b1t", 4, 14, 320, 320, *T_(3)),
        ("5c.b0", 2, 7, 832, 384, *P_), ("5c.b1s", 2, 7, 192, 384, *S_(3)), ("5c.b1t", 2, 7, 384, 384, *T_(3))]
if args.stems:
    LAYERS = STEMS
if args.r21d:
    LAYERS = R21D
if args.s3dg:
    LAYERS = S3DG
    if B == 32:
        B = 16
if args.r3d:
    LAYERS = R3D
tot = {}
if args.layers:
    LAYERS = [L for L in LAYERS if L[0] in args.layers.split(",")]
for L in LAYERS:
    name, T, HW, cin, cout = L[:5]
    k, s, p = L[5:] if len(L) > 5 else ((3, 3, 3), (1, 1, 1), (1, 1, 1))
    g = ConvGeom(B, T, HW, HW, cin, cout, k, s, p)
    x = torch.randn(B, T, HW, HW, cin, device=dev)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    dy = torch.randn(B, *g.out_dims, cout, device=dev)
    wp = be.conv_pack_fwd(g, w)
    dw = torch.empty_like(w)
    line = f"{name:7s} {g.flops / 1e9:8.1f} GF"
    for what in args.what.split(","):
        if what == "fwd":
            ms = timeit(lambda: be.conv_fwd(g, x, wp, None, True))
        elif what == "dgrad":
            if cin <= 4:
                continue
            ms = timeit(lambda: be.conv_dgrad(g, dy, w))
        else:
            ms = timeit(lambda: be.conv_wgrad(g, x, dy, dw))
        tot[what] = tot.get(what, 0) + ms
        line += f" | {what} {ms:7.3f} ms {g.flops / ms / 1e9:6.1f} TF"
    print(line, flush=True)
print("total ms:", {k: round(v, 2) for k, v in tot.items()})
