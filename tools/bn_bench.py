#!/usr/bin/env python
"""BatchNorm forward-apply / backward micro-benchmark: achieved HBM GB/s (algorithmic bytes) per shape.
fwd: read y, write out;  bwd: reduce reads y + dout, apply reads y + dout and writes dy."""
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
SHAPES = [("c3d conv1 pool122", 32, 16, 112, 112, 64, (1, 2, 2)), ("c3d conv2 pool222", 32, 16, 56, 56, 128, (2, 2, 2)),
          ("r21d 144ch", 32, 16, 56, 56, 144, (1, 1, 1)), ("r21d 64ch", 32, 16, 56, 56, 64, (1, 1, 1)), ("r21d 288ch", 32, 8, 28, 28, 288, (1, 1, 1)),
          ("r3d 256ch", 32, 4, 14, 14, 256, (1, 1, 1)), ("s3dg 192ch", 16, 16, 56, 56, 192, (1, 1, 1)), ("s3dg 4x 208ch", 16, 8, 14, 14, 208, (1, 1, 1)),
          ("s3dg 5x 384ch", 16, 4, 7, 7, 384, (1, 1, 1))]
for name, N, D, H, W, C, k in SHAPES:
    pg = PoolGeom(N, D, H, W, C, k, k, (0, 0, 0))
    y = torch.randn(N, D, H, W, C, device=dev)
    ss = torch.stack([torch.rand(C, device=dev) + 0.5, torch.randn(C, device=dev) * 0.1])
    mi = torch.stack([torch.randn(C, device=dev) * 0.1, torch.rand(C, device=dev) + 0.5])
    gamma = torch.rand(C, device=dev) + 0.5
    out = be.bn_act_pool_fwd(pg, y, ss, None, True)
    dout = torch.randn_like(out)
    dg, db = torch.empty(C, device=dev), torch.empty(C, device=dev)
    dy = torch.empty_like(y)
    f = timeit(lambda: be.bn_act_pool_fwd(pg, y, ss, None, True, out=out))
    b = timeit(lambda: be.bn_act_pool_bwd(pg, y, None, dout, gamma, mi, ss, True, False, dg, db, dy_out=dy))
    yb, ob = y.numel() * 4, out.numel() * 4
    print(f"{name:20s} fwd {f*1e3:8.1f} us {(yb+ob)/f/1e6:7.0f} GB/s | bwd (reduce+finalize+apply) {b*1e3:8.1f} us {(3*yb+2*ob)/b/1e6:7.0f} GB/s", flush=True)
