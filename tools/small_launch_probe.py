"""GPU box: the small late-layer convolutions of S3D-G / R3D-18 launched (a) back to back, 300 in a row (sustained clocks: what a
training step sees) and (b) one at a time behind an idle gap (what an isolated micro-benchmark sees).  us per launch and TFLOP/s."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rspnet_amd import ops  # noqa: E402
from rspnet_amd.ops import ConvGeom  # noqa: E402

be = ops.backend()
dev = torch.device("cuda:0")
CASES = [   # name, N, D, H, W, Cin, Cout, k, s, p
    ("s3dg_4b.b0 1x1", 16, 4, 14, 14, 480, 192, (1, 1, 1), (1, 1, 1), (0, 0, 0)),
    ("s3dg_4f.b1 sp", 16, 4, 14, 14, 160, 320, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("s3dg_4f.b1 tm", 16, 4, 14, 14, 320, 320, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s3dg_5c.b1 tm", 16, 2, 7, 7, 384, 384, (3, 1, 1), (1, 1, 1), (1, 0, 0)),
    ("s3dg_3b.b1 sp", 16, 8, 28, 28, 96, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("r3d_l3 3x3x3", 32, 4, 14, 14, 256, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    ("r3d_l4 3x3x3", 32, 2, 7, 7, 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
    # exactly 1 / 2 / 3 / 6 tiles of 128x128 per CU, K = 1152 (36 chunks): time per chunk against the workgroups sharing a CU
    ("256 tiles", 16, 4, 16, 32, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("512 tiles", 16, 8, 16, 32, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("768 tiles", 16, 12, 16, 32, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
    ("1536 tiles", 16, 24, 16, 32, 128, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
]
for name, N, D, H, W, ci, co, k, s, p in CASES:
    g = ConvGeom(N, D, H, W, ci, co, k, s, p)
    x = torch.randn(N, D, H, W, ci, device=dev)
    w = torch.randn(co, ci, *k, device=dev) * 0.05
    ps = be.pack_set([(g, 0, w)])
    ps.run()
    wp = ps.packed[0]
    for _ in range(20):
        be.conv_fwd(g, x, wp, None, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 300
    e0.record()
    for _ in range(n):
        be.conv_fwd(g, x, wp, None, True)
    e1.record()
    torch.cuda.synchronize()
    sustained = e0.elapsed_time(e1) / n * 1e3
    iso = []
    for _ in range(10):
        time.sleep(0.005)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        be.conv_fwd(g, x, wp, None, True)
        b.record()
        torch.cuda.synchronize()
        iso.append(a.elapsed_time(b) * 1e3)
    iso.sort()
    name_k = be.lib.rsp_last_conv_kernel().decode()
    print(f"{name:16s} rows {g.rows:7d} K {ci * k[0] * k[1] * k[2]:5d} N {co:4d}  back-to-back {sustained:7.1f} us {g.flops / sustained / 1e6:6.1f} TF | "
          f"isolated {iso[len(iso) // 2]:7.1f} us {g.flops / iso[len(iso) // 2] / 1e6:6.1f} TF  {name_k}")
