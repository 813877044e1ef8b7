#!/usr/bin/env python
"""Fine-tune step throughput: MultiTaskWrapper(finetune=True) train step (forward + CrossEntropyLoss + backward
+ torch SGD) and eval-mode forward at the pretext geometry, B=32 clips of 3x16x112x112, 101 classes (UCF-101), one MI355X.
Conv FLOPs per clip: train 3F - F_first, eval F (C3D F = 76.99 GF, first conv 2.08)."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rspnet_amd.finetune import train_step, validate_step
from rspnet_amd.models import ModelFactory

GF = {"c3d": (76.99, 2.08), "resnet18": (16.62, 6.61), "r2plus1d-vcop": (42.72, 1.22), "s3dg": (34.07, 1.89)}
ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="c3d", choices=sorted(GF))
ap.add_argument("--batch", type=int, default=32)
ap.add_argument("--steps", type=int, default=10)
args = ap.parse_args()
dev = torch.device("cuda", 0)
hw = 224 if args.arch == "s3dg" else 112
model = ModelFactory({"model": {"arch": args.arch}, "dataset": {"num_classes": 101}}).build_multitask_wrapper(0)
crit = torch.nn.CrossEntropyLoss()
opt = torch.optim.SGD(model.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
g = torch.Generator(device=dev).manual_seed(0)
x = torch.randn(args.batch, 3, 16, hw, hw, device=dev, generator=g)
y = torch.randint(0, 101, (args.batch,), device=dev, generator=g)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / args.steps


model.train()
t_train = timed(lambda: train_step(model, crit, opt, x, y))
model.eval()
t_eval = timed(lambda: validate_step(model, crit, x, y))
F, F1 = GF[args.arch]
print(f"{args.arch} B={args.batch} {hw}x{hw}: train step {t_train * 1e3:.1f} ms = {args.batch / t_train:.0f} clips/s "
      f"({args.batch * (3 * F - F1) / t_train / 1e3:.1f} conv TFLOP/s); eval forward {t_eval * 1e3:.1f} ms = {args.batch / t_eval:.0f} clips/s "
      f"({args.batch * F / t_eval / 1e3:.1f} conv TFLOP/s)")
