#!/usr/bin/env python
"""igemm forward efficiency vs GEMM K at fixed M (16x8x56x56 rows) and N=128: how much of a tile's time is prologue/epilogue."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--cout", type=int, default=128)
ap.add_argument("--rows-scale", type=int, default=1, help="multiply the clip count (16) by this")
args = ap.parse_args()
import torch
from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom
be = ops.backend(); dev = torch.device("cuda", 0)
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for T, HW in ((8, 56), (4, 14)):
    for cin in (32, 64, 128, 256, 512, 1024):
        g = ConvGeom(16 * args.rows_scale, T, HW, HW, cin, args.cout, (1, 3, 3), (1, 1, 1), (0, 1, 1))
        x = torch.randn(16 * args.rows_scale, T, HW, HW, cin, device=dev); w = torch.randn(args.cout, cin, 1, 3, 3, device=dev) * 0.05
        wp = be.conv_pack_fwd(g, w)
        ms = timeit(lambda: be.conv_fwd(g, x, wp, None, True))
        print(f"rows {g.rows:7d} K {cin*9:5d} ({cin*9//32:3d} chunks): {ms*1e3:8.1f} us  {g.flops/ms/1e9:6.1f} TF", flush=True)
