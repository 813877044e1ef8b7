#!/usr/bin/env python
"""Throughput of the fused augmentation (rsp_augment_batch) at the shipped pretext geometry: 32 samples x 2 clips, T=32,
256x340 uint8 crops -> 112x112 float32.  Prints kernel time (HIP events around the launch group, inputs resident in HBM),
algorithmic bytes (crop read once per pass + output written once) vs the ~6.3 TB/s achievable HBM rate, the end-to-end collate
call (staging + H2D + kernels) and the CPU restatement of the reference pipeline on a 4-clip sample beside it."""
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import ctypes as C

import torch

from rspnet_amd import _lib, ops
from rspnet_amd.augment import FusedGPUCollateFn

B, NC, T, H, W, S = 32, 2, 32, 256, 340, 112
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]
dev = torch.device("cuda", 0)
g = torch.Generator().manual_seed(0)
batch = [([torch.randint(0, 256, (T, H, W, 3), dtype=torch.uint8, generator=g) for _ in range(NC)], b) for b in range(B)]
fn = FusedGPUCollateFn(S, MEAN, STD, device=dev)
random.seed(0)
fn(batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(3):
    fn(batch)
torch.cuda.synchronize()
e2e = (time.perf_counter() - t0) / 3

# kernels only: descriptors + crops resident on the device
n = B * NC
src = torch.randint(0, 256, (n, T, H, W, 3), dtype=torch.uint8, device=dev)
descs = (_lib.AugmentClipDesc * n)()
random.seed(1)
passes = 0
for i, d in enumerate(descs):
    gray, flip, op_list = fn.draw()
    d.src = src[i].data_ptr()
    d.frame_pitch, d.row_pitch, d.h, d.w = H * W * 3, W * 3, H, W
    d.gray, d.flip, d.n_ops = int(gray), int(flip), len(op_list)
    for k, (op, f) in enumerate(op_list):
        d.op[k], d.factor[k], d.one_minus[k] = op, f, 1.0 - f
    passes += 2 if any(op == 1 for op, _ in op_list) else 1
ddev = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).to(dev)
out = torch.empty((n, 3, T, S, S), dtype=torch.float32, device=dev)
be = ops.backend()
be.augment_batch(ddev, n, T, S, MEAN, STD, out)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    be.augment_batch(ddev, n, T, S, MEAN, STD, out)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 10
# bilinear down-sampling touches at most 4 source pixels per output pixel: bytes actually needed <= min(crop, 4 taps x 3 B x pixels)
need = min(T * H * W * 3, T * S * S * 12)
alg = passes * need + n * 3 * T * S * S * 4
print(f"kernels: {ms:.3f} ms per batch of {n} clips ({n / ms * 1e3:.0f} clips/s); algorithmic bytes {alg / 1e9:.3f} GB -> {alg / ms / 1e6:.0f} GB/s "
      f"({alg / ms / 1e6 / 6300:.2f} of 6.3 TB/s)")
print(f"collate call (pinned staging + H2D of {n * T * H * W * 3 / 1e9:.2f} GB uint8 + kernels): {e2e * 1e3:.1f} ms per batch")

from oracle import augment as A
random.seed(2)
t0 = time.perf_counter()
for i in range(4):
    A.augment_clip(batch[i][0][0], S, A.draw_params(), MEAN, STD)
dt = (time.perf_counter() - t0) / 4
print(f"cpu restatement of the reference pipeline: {dt * 1e3:.1f} ms per clip ({torch.get_num_threads()} threads) -> {dt * n * 1e3:.0f} ms per batch")
