#!/usr/bin/env python
"""Where a tile's time goes in igemm_body: per-wave timestamps (s_memtime) at entry / first chunk landed / K loop done / stores issued /
end, from the -DRSP_PHASE_PROBE build (tools/build_probe.sh).  Prints per-shape phase means and, per SIMD, how much of the launch at
least one resident wave spent inside its K loop (tools/phase_analyze.py reads the dumps).  Usage (GPU box): python tools/phase_probe.py [--shapes ksweep,r21d,s3dg]"""
import argparse
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["RSPNET_HIP_LIB"] = os.path.join(ROOT, "tools", "librspnet_hip_probe.so")
ap = argparse.ArgumentParser()
ap.add_argument("--shapes", default="ksweep")
args = ap.parse_args()

import numpy as np
import torch
from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom

be = ops.backend()
lib = be.lib
lib.rsp_phase_probe_set.argtypes = [C.c_void_p]
lib.rsp_phase_probe_set.restype = None
dev = torch.device("cuda", 0)
S_ = lambda k: ((1, k, k), (1, 1, 1), (0, k // 2, k // 2))
T_ = lambda k, st=1: ((k, 1, 1), (st, 1, 1), (k // 2, 0, 0))
P_ = ((1, 1, 1), (1, 1, 1), (0, 0, 0))
SHAPES = {
    "ksweep": [(f"K{c * 9}", 16, 8, 56, c, 128, *S_(3)) for c in (32, 64, 128, 512)],
    "r21d": [("c2.sp", 32, 16, 56, 64, 144, *S_(3)), ("c2.tm", 32, 16, 56, 144, 64, *T_(3)), ("c3b.sp", 32, 8, 28, 128, 288, *S_(3)),
             ("c3b.tm", 32, 8, 28, 288, 128, *T_(3)), ("c4b.sp", 32, 4, 14, 256, 576, *S_(3))],
    "s3dg": [("sc2.sp", 16, 8, 56, 64, 192, *S_(3)), ("3b.b1s", 16, 8, 28, 96, 128, *S_(3)), ("4b.b0", 16, 4, 14, 480, 192, *P_),
             ("4f.b1s", 16, 4, 14, 160, 320, *S_(3)), ("5c.b1t", 16, 2, 7, 384, 384, *T_(3))],
    "r3d": [("l1", 32, 8, 28, 64, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("l2", 32, 4, 14, 128, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1))],
}
for group in args.shapes.split(","):
    for name, B, T, HW, cin, cout, k, s, p in SHAPES[group]:
        g = ConvGeom(B, T, HW, HW, cin, cout, k, s, p)
        x = torch.randn(B, T, HW, HW, cin, device=dev)
        w = torch.randn(cout, cin, *k, device=dev) * 0.05
        wp = be.conv_pack_fwd(g, w)
        fn = lambda: be.conv_fwd(g, x, wp, None, True)
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        nwg = 1 << 16
        dbg = torch.zeros(nwg * 4 * 8, dtype=torch.int64, device=dev)
        lib.rsp_phase_probe_set(C.c_void_p(dbg.data_ptr()))
        fn()
        torch.cuda.synchronize()
        lib.rsp_phase_probe_set(None)
        d = dbg.cpu().numpy().reshape(-1, 8).astype(np.uint64)
        d = d[d[:, 0] != 0]
        out = os.environ.get("PROBE_RAW", os.path.join(ROOT, "gpurun_out", "phase_raw"))
        os.makedirs(out, exist_ok=True)
        np.save(os.path.join(out, f"{group}_{name}.npy"), d)
        with open(os.path.join(out, f"{group}_{name}.txt"), "w") as f:
            f.write(f"{ms} {g.flops} {g.rows} {cin * k[0] * k[1] * k[2]} {cout}\n")
        print(f"{group}/{name}: {ms * 1e3:8.1f} us {g.flops / ms / 1e9:6.1f} TF {lib.rsp_last_conv_kernel().decode()} waves {len(d)}", flush=True)
