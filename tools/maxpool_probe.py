#!/usr/bin/env python
"""Effective bandwidth of the stand-alone MaxPool3d kernels on S3D-G's / R3D-18's pools: algorithmic bytes (input once, output once,
arg-max once when kept; backward: dout + arg-max in, dx out) / time."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd import ops
from rspnet_amd.ops import PoolGeom

be = ops.backend()
dev = torch.device("cuda", 0)
K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)
SHAPES = [("s3dg maxPool1", 16, 8, 112, 112, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)), ("s3dg maxPool2", 16, 8, 56, 56, 192, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
          ("s3dg 3b branch3", 16, 8, 28, 28, 192, K3, S1, P1), ("s3dg 3c branch3", 16, 8, 28, 28, 256, K3, S1, P1),
          ("s3dg maxPool3", 16, 8, 28, 28, 480, K3, (2, 2, 2), P1), ("s3dg 4b branch3", 16, 4, 14, 14, 480, K3, S1, P1),
          ("s3dg 4f branch3", 16, 4, 14, 14, 528, K3, S1, P1), ("s3dg maxpool4", 16, 4, 14, 14, 832, (2, 2, 2), (2, 2, 2), (0, 0, 0)),
          ("s3dg 5b branch3", 16, 2, 7, 7, 832, K3, S1, P1), ("r3d maxpool", 32, 16, 56, 56, 64, K3, (2, 2, 2), P1)]


def timeit(fn, it=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it


tot = [0.0, 0.0, 0.0]
for name, N, D, H, W, Cc, k, s, p in SHAPES:
    pg = PoolGeom(N, D, H, W, Cc, k, s, p)
    x = torch.relu(torch.randn(N, D, H, W, Cc, device=dev))
    out, idx = be.maxpool_fwd(pg, x, True)
    dout = torch.randn_like(out)
    t0 = timeit(lambda: be.maxpool_fwd(pg, x, False))
    t1 = timeit(lambda: be.maxpool_fwd(pg, x, True))
    t2 = timeit(lambda: be.maxpool_bwd(pg, dout, idx))
    b0, b1, b2 = 4 * (x.numel() + out.numel()), 4 * (x.numel() + 2 * out.numel()), 4 * (x.numel() + 2 * out.numel())
    tot[0] += t0; tot[1] += t1; tot[2] += t2
    print(f"{name:18s} fwd {t0 * 1e3:7.1f} us {b0 / t0 / 1e9:5.2f} TB/s | fwd+argmax {t1 * 1e3:7.1f} us {b1 / t1 / 1e9:5.2f} TB/s | "
          f"bwd {t2 * 1e3:7.1f} us {b2 / t2 / 1e9:5.2f} TB/s", flush=True)
print(f"sum: fwd {tot[0] * 1e3:.0f} us, fwd+argmax {tot[1] * 1e3:.0f} us, bwd {tot[2] * 1e3:.0f} us")
