#!/usr/bin/env python
"""Debug helper: run one golden case on the HIP backend and on the CPU checker backend, print per-tensor gradient distances."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from cpu_ops import CpuOps
from golden_util import build_inputs, cases_for, load_case
from model_util import run_model_step
from rspnet_amd import ops
arch = sys.argv[1]
a, ws, seed = cases_for(arch, 1)[0]
z, meta = load_case(a, ws, seed)
spec, inputs = build_inputs(a, meta)
hip = ops.backend()
r1, p1, m1, g1 = run_model_step(a, meta, inputs, 0, torch.device("cuda", 0), "fused")
ops.set_backend(CpuOps())
r2, p2, m2, g2 = run_model_step(a, meta, inputs, 0, torch.device("cpu"), "fused")
print("loss", r1["loss"], r2["loss"], "logits diff", np.abs(r1["logits1"] - r2["logits1"]).max())
errs = sorted(((float(np.linalg.norm(g1[k] - g2[k]) / max(np.linalg.norm(g2[k]), 1e-12)), float(np.linalg.norm(g2[k])), k) for k in g1 if g1[k] is not None), reverse=True)
for e in errs[:12]:
    print(f"{e[0]:.3e}  norm {e[1]:.3e}  {e[2]}")
print("median", np.median([e[0] for e in errs]))
