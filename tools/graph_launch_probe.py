#!/usr/bin/env python
"""Is a replayed step bound by the GPU or by the host's graph submission?  Times, per replay of the captured pretext step:
host time of the replay call with the GPU idle (synchronised before), the GPU's own time for that replay, and the back-to-back
step interval."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch

arch = sys.argv[1] if len(sys.argv) > 1 else "s3dg"
from model_util import make_cfg
from rspnet_amd.graph_step import GraphedPretextStep
from rspnet_amd.moco import ModelFactory
from rspnet_amd.moco.builder_diffspeed_diffloss import Loss

dev = torch.device("cuda", 0)
hw, B = (224, 16) if arch == "s3dg" else (112, 32)
model = ModelFactory(make_cfg(arch, 16384)).build_moco_diffloss(device=dev)
crit = Loss(margin=2.0, A=1.0, M=1.0)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05, momentum=0.9, weight_decay=1e-4)
im_q = torch.randn(B, 3, 32, hw, hw, device=dev)
im_k = torch.randn(B, 3, 32, hw, hw, device=dev)
stepper = GraphedPretextStep(model, crit, opt, issue="graph")
for _ in range(6):
    stepper(im_q, im_k)
torch.cuda.synchronize()
print("graphs:", len(stepper.graphs), "fallback:", stepper.fallback_reason)
host, gpu = [], []
for _ in range(10):
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    stepper(im_q, im_k)
    e1.record()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    host.append((t1 - t0) * 1e3)
    gpu.append(e0.elapsed_time(e1))
print(f"{arch}: GPU idle before each replay: host call {sorted(host)[5]:.2f} ms (min {min(host):.2f}), GPU time {sorted(gpu)[5]:.2f} ms")
torch.cuda.synchronize()
t0 = time.perf_counter()
hs = []
for _ in range(30):
    a = time.perf_counter()
    stepper(im_q, im_k)
    hs.append((time.perf_counter() - a) * 1e3)
torch.cuda.synchronize()
print(f"{arch}: back to back: {(time.perf_counter() - t0) / 30 * 1e3:.2f} ms per step, host call p50 {sorted(hs)[15]:.2f} ms")
