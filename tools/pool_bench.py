#!/usr/bin/env python
"""Stand-alone MaxPool3d micro-benchmark (S3D-G / R3D-18 shapes): us and algorithmic GB/s (x read once + out written once;
backward: dout + idx read, dx written)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd import ops
from rspnet_amd.ops import PoolGeom
be = ops.backend(); dev = torch.device("cuda", 0)
def timeit(fn, it=10):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
SHAPES = [("s3dg maxPool1", 16, 8, 112, 112, 64, (1, 3, 3), (1, 2, 2), (0, 1, 1)), ("s3dg maxPool2", 16, 8, 56, 56, 192, (1, 3, 3), (1, 2, 2), (0, 1, 1)),
          ("s3dg 3b pool", 16, 8, 28, 28, 192, (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("s3dg 3c pool", 16, 8, 28, 28, 256, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
          ("s3dg maxPool3", 16, 8, 28, 28, 480, (3, 3, 3), (2, 2, 2), (1, 1, 1)), ("s3dg 4b pool", 16, 4, 14, 14, 480, (3, 3, 3), (1, 1, 1), (1, 1, 1)),
          ("s3dg 4f pool", 16, 4, 14, 14, 528, (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("s3dg maxpool4", 16, 4, 14, 14, 832, (2, 2, 2), (2, 2, 2), (0, 0, 0)),
          ("s3dg 5b pool", 16, 2, 7, 7, 832, (3, 3, 3), (1, 1, 1), (1, 1, 1)), ("r3d stem pool", 32, 16, 56, 56, 64, (3, 3, 3), (2, 2, 2), (1, 1, 1))]
for name, N, D, H, W, C, k, s, p in SHAPES:
    pg = PoolGeom(N, D, H, W, C, k, s, p)
    x = torch.randn(N, D, H, W, C, device=dev)
    out, idx = be.maxpool_fwd(pg, x, True)
    dout = torch.randn_like(out)
    f = timeit(lambda: be.maxpool_fwd(pg, x, True))
    f0 = timeit(lambda: be.maxpool_fwd(pg, x, False))
    b = timeit(lambda: be.maxpool_bwd(pg, dout, idx))
    xb, ob = x.numel() * 4, out.numel() * 4
    print(f"{name:16s} fwd(keep) {f*1e3:7.1f} us {(xb+2*ob)/f/1e6:6.0f} GB/s | fwd {f0*1e3:7.1f} us {(xb+ob)/f0/1e6:6.0f} GB/s | bwd {b*1e3:7.1f} us {(xb+2*ob)/b/1e6:6.0f} GB/s", flush=True)
