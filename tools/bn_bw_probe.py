#!/usr/bin/env python
"""Effective HBM bandwidth of the BatchNorm kernels (apply+ReLU(+pool) forward, backward reduce + apply) on the big activation
shapes of each backbone: algorithmic bytes (one read of every input, one write of every output) / time.  Cold = a 1 GB
scratch fill between calls (the tensors do not come from the Infinity Cache), warm = back to back."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd import ops
from rspnet_amd.ops import PoolGeom

be = ops.backend()
dev = torch.device("cuda", 0)
SHAPES = [("r2+1d conv2 mid", 32, 16, 56, 56, 144, None), ("r2+1d conv2 out", 32, 16, 56, 56, 64, None),
          ("c3d conv1+pool", 32, 16, 112, 112, 64, ((1, 2, 2), (1, 2, 2))), ("c3d conv2+pool", 32, 16, 56, 56, 128, ((2, 2, 2), (2, 2, 2))),
          ("c3d conv3a", 32, 8, 28, 28, 256, None), ("s3dg conv1 7x1x1", 16, 8, 112, 112, 64, None),
          ("s3dg 3b in", 16, 8, 28, 28, 192, None), ("r3d stem", 32, 16, 56, 56, 64, None),
          ("s3dg 28x28x96", 16, 8, 28, 28, 96, None), ("s3dg 14x14x320", 16, 4, 14, 14, 320, None), ("s3dg 14x14x64", 16, 4, 14, 14, 64, None),
          ("s3dg 7x7x384", 16, 2, 7, 7, 384, None), ("s3dg 7x7x128", 16, 2, 7, 7, 128, None), ("r3d 14x14x128", 32, 4, 14, 14, 128, None),
          ("r3d 4x4x512", 32, 1, 4, 4, 512, None)]
scratch = torch.empty(256 << 20, device=dev)


def timeit(fn, cold, it=10):
    fn()
    torch.cuda.synchronize()
    tot = 0.0
    for _ in range(it):
        if cold:
            scratch.fill_(1.0)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        tot += e0.elapsed_time(e1)
    return tot / it


for name, N, D, H, W, Cc, pool in SHAPES:
    pk, ps = pool if pool else ((1, 1, 1), (1, 1, 1))
    pg = PoolGeom(N, D, H, W, Cc, pk, ps, (0, 0, 0))
    y = torch.randn(N, D, H, W, Cc, device=dev)
    ss = torch.stack([torch.rand(Cc, device=dev) + 0.5, torch.randn(Cc, device=dev) * 0.1]).contiguous()
    mi = torch.stack([torch.randn(Cc, device=dev) * 0.1, torch.rand(Cc, device=dev) + 0.5]).contiguous()
    gamma = torch.rand(Cc, device=dev) + 0.5
    out = be.bn_act_pool_fwd(pg, y, ss, None, True)
    dout = torch.randn_like(out)
    dg, db = torch.empty(Cc, device=dev), torch.empty(Cc, device=dev)
    dy = torch.empty_like(y)
    nb_f = 4 * (y.numel() + out.numel())
    nb_b = 4 * (2 * y.numel() + 2 * dout.numel() + dy.numel())     # reduce reads y, dout; apply reads y, dout, writes dy
    for cold in (False, True):
        tf = timeit(lambda: be.bn_act_pool_fwd(pg, y, ss, None, True, out=out), cold)
        tb = timeit(lambda: be.bn_act_pool_bwd(pg, y, None, dout, gamma, mi, ss, True, False, dg, db, dy_out=dy), cold)
        print(f"{name:18s} {'cold' if cold else 'warm'}  fwd {tf * 1e3:7.1f} us {nb_f / tf / 1e9:6.2f} TB/s ({nb_f / 1e6:7.0f} MB) | "
              f"bwd {tb * 1e3:7.1f} us {nb_b / tb / 1e9:6.2f} TB/s ({nb_b / 1e6:7.0f} MB)", flush=True)
