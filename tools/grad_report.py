#!/usr/bin/env python
"""Runs every single-rank golden case on the HIP backend and prints the whole-step error figures (no assertions):
forward worst, post-state / momentum / gradient relative-L2 estimates, next to the gate tests/golden_util.py would apply."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
from golden_util import ALL_CASES, build_inputs, grad_tol, load_case, rel_err, tensor_err, worst_grad_err
from model_util import run_model_step
for a, ws, seed in ALL_CASES:
    if ws != 1:
        continue
    z, meta = load_case(a, ws, seed)
    spec, inputs = build_inputs(a, meta)
    res, post, mom, grads = run_model_step(a, meta, inputs, 0, torch.device("cuda", 0), "fused")
    fwd = max(rel_err(res[k], z["r0." + k]) for k in ("loss", "logits1", "logits2", "l_pos_M", "l_neg_M", "q_A", "q_M", "k_A_shuf", "kneg_A_shuf"))
    wk, wg = worst_grad_err(z, 0, grads)
    wm = max((tensor_err(z, 0, "mom", n[len("r0.momsum."):], mom[n[len("r0.momsum."):]]) for n in z.files
              if n.startswith("r0.momsum.") and z["r0.gradsum." + n[len("r0.momsum."):]].size), default=0.0)
    print(f"{a:16s} s{seed}: fwd {fwd:.1e}  grad {wg:.2e} ({wk.split('.')[-3:]})  mom {wm:.2e}  gate {grad_tol(a):.1e}", flush=True)
