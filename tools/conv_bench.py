#!/usr/bin/env python
"""Per-layer conv micro-benchmark (C3D shapes at B=32 by default): TFLOP/s of fwd / dgrad / wgrad launches.
Usage: python tools/conv_bench.py [--tune BITS] ;  with --tune it loads tools/librspnet_hip_tune.so (ablation build)."""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

ap = argparse.ArgumentParser()
ap.add_argument("--tune", type=int, default=None)
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--iters", type=int, default=5)
ap.add_argument("--what", default="fwd,dgrad,wgrad")
ap.add_argument("--stems", action="store_true", help="the four backbone stems (Cin padded to 4) instead of the C3D stack")
ap.add_argument("--r3d", action="store_true", help="R3D-18 residual-stage conv shapes")
ap.add_argument("--no-stem-kernel", action="store_true", help="ablation build only: route stems through the implicit-GEMM kernel")
args = ap.parse_args()
if args.tune is not None:
    os.environ["RSPNET_HIP_LIB"] = os.path.join(ROOT, "tools", "librspnet_hip_tune.so")
    os.environ["RSP_TUNE"] = str(args.tune)
if args.no_stem_kernel:
    os.environ["RSP_NO_STEM"] = "1"

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
if args.stems:
    LAYERS = STEMS
if args.r3d:
    LAYERS = R3D
tot = {}
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
