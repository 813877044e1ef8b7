#!/usr/bin/env python
"""Dump the parameter gradients of the seeded first bench step (tiny or full size) to a file: two runs under different environment
switches are then compared tensor by tensor (tools/grad_dump.py --compare a.pt b.pt)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

if sys.argv[1] == "--compare":
    a, b = torch.load(sys.argv[2]), torch.load(sys.argv[3])
    for k in a:
        d = float((a[k].double() - b[k].double()).norm() / b[k].double().norm().clamp_min(1e-30))
        if d > 1e-5:
            print(f"{k:50s} rel l2 {d:.3e}  shape {tuple(a[k].shape)}")
    sys.exit(0)
import bench

out, arch, B, hw, K = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
args = bench.parse_args(["--arch", arch, "--steps", "1", "--warmup", "0", "--graph", "off", "--batch", str(B), "--hw", str(hw), "--queue", str(K),
                         "--no-other-workloads"])
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
m, cap = bench.measure(args, arch, B, hw, bench.ARCHS[arch][2], 1, 0, dev, 0, 1, want_parity=True)
torch.save({k: v.clone() for k, v in cap["grads"].items()}, out)
print("saved", out, len(cap["grads"]))
