"""How well-conditioned are the fixtures' gradients?  TEST INFRASTRUCTURE ONLY.

``python -m oracle.gen_conditioning [arch | arch@ws2 ...]`` runs the oracle restatement of every fixture case (multi-rank ones under the key `arch@wsN`) three times
on the same inputs with the same code — fp32 as the reference does (oneDNN convolutions, fused batch_norm), fp64, fp32 with
oneDNN switched off (ATen's native convolution) and fp32 with BatchNorm evaluated in its folded scale/shift form (what a fused
conv+BN kernel computes): the same function in other summation / evaluation orders, which is all that separates any two
correct fp32 implementations — plus, as a fourth order, the product's host logic on the torch checker backend compared with
the gradients stored in the fixture — and records, per case, the relative L2 distance of the variants' parameter gradients (max and
median over tensors) and logits from the default run in tests/golden/conditioning.json; `grad_rel_l2_max` is the larger one.

Why: in a ReLU / max-pool network the backward pass is discontinuous in the forward values.  Two correct fp32
implementations whose forward activations differ by a relative delta (summation order; delta ~ 1e-6 for C3D, ~3e-5 for the
deep S3D-G / Bottleneck stacks) decide a fraction ~delta of the ReLU masks / pool arg-maxes differently, and each such element
changes its gradient contribution by O(1): the whole-gradient distance is ~sqrt(delta) (3e-3 .. 2e-2), independent of the
fixture size (more elements: more flips, each weighing less).  kappa = |g_fp32 - g_fp64| / |g_fp64| of the ORACLE ITSELF is
therefore the floor under any whole-step gradient comparison with the reference's fp32 numbers, and tests/golden_util.py
derives the per-backbone whole-step gradient gate from it (the exact, unit-by-unit check of the backward composition is the
teacher-forced replay, tests/test_teacher_forced_gpu.py, at 2e-5)."""
from __future__ import annotations

import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import gen_golden as G      # noqa: E402
from oracle import restatement as S     # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def measure(tag, seed, meta=None, spec=None, fast=False, ws=1):
    """fast=True: skip the fp64 variant (gen_golden's seed screen).  ws > 1: a multi-rank fixture — the same variants over `ws`
    simulated ranks, distances of the DDP-averaged gradients (the checker variant is single-rank only)."""
    if spec is None:
        with open(os.path.join(GOLDEN, f"state_spec_{G.tag_file(tag)}.json")) as f:
            spec = {k: (tuple(s), d) for k, (s, d) in json.load(f).items()}
    z = None
    if meta is None:
        z = np.load(os.path.join(GOLDEN, G.case_name(tag, ws, seed) + ".npz"))
        meta = json.loads(str(z["meta"]))
        meta["nudges"] = G.nudges_from_npz(z)
    state, mom, clips, perms_B, sh = G.case_inputs(spec, tag, meta["B"], meta["HW"], meta["K"], ws, seed, meta.get("nudges"))

    import contextlib

    def run(dt, mkldnn=True, folded_bn=False):
        torch.set_default_dtype(dt)
        try:
            with torch.backends.mkldnn.flags(enabled=mkldnn), (S.bn_scale_shift() if folded_bn else contextlib.nullcontext()):
                return _run_once(dt)
        finally:
            torch.set_default_dtype(torch.float32)

    def _run_once(dt):
        if True:
            sts = [{k: (torch.from_numpy(v.copy()).to(dt) if v.dtype == np.float32 else torch.from_numpy(v.copy())) for k, v in state.items()}
                   for _ in range(ws)]
            return S.moco_step(meta["arch"], sts, [torch.from_numpy(clips[r][0]).to(dt) for r in range(ws)],
                               [torch.from_numpy(clips[r][1]).to(dt) for r in range(ws)],
                               [torch.from_numpy(perms_B[r]) for r in range(ws)], (torch.from_numpy(sh[0]), torch.from_numpy(sh[1])),
                               meta["speed"], K=meta["K"], lr=meta["lr"], fc_type=meta.get("fc_type", "linear"),
                               momentum_buffers=[{} for _ in range(ws)])[0]

    o32 = run(torch.float32)
    out = {}
    variants = [("native_conv", lambda: run(torch.float32, mkldnn=False)), ("folded_bn", lambda: run(torch.float32, folded_bn=True))]
    if not fast:
        variants.insert(0, ("fp64", lambda: run(torch.float64)))
    for tag_v, fn in variants:
        other = fn()
        errs = []
        for k, g in o32["grads"].items():
            if g is None or float(g.norm()) < 1e-4:
                continue
            errs.append(float((other["grads"][k].double() - g.double()).norm() / g.double().norm()))
        out[f"grad_rel_l2_max_{tag_v}"] = max(errs)
        out[f"grad_rel_l2_median_{tag_v}"] = float(np.median(errs))
        out[f"logits_rel_{tag_v}"] = float((other["logits1"].double() - o32["logits1"].double()).abs().max() / o32["logits1"].abs().max())
    if ws == 1 and z is not None and "r0.gradproj." + next(k for k, g in o32["grads"].items() if g is not None) in z.files:
        # fourth evaluation order: the product's host logic on the torch checker backend (channels-last convolutions, folded
        # BatchNorm, fused sibling GEMMs) against the reference's own gradients stored in the fixture (sketch estimate)
        out["grad_rel_l2_max_checker"] = G.checker_grad_error(tag, meta, spec, z)
    out["grad_rel_l2_max"] = max(v for k, v in out.items() if k.startswith("grad_rel_l2_max_"))
    return out


def main():
    only = sys.argv[1:]
    path = os.path.join(GOLDEN, "conditioning.json")
    out = json.load(open(path)) if (only and os.path.exists(path)) else {}      # a full run starts from scratch
    for tag in only:
        out.pop(tag, None)
        if "@" not in tag:
            for k in [k for k in out if k.startswith(tag + "@ws")]:
                out.pop(k)
    with open(os.path.join(GOLDEN, "index.json")) as f:
        index = json.load(f)
    for tag, ws, seed in index:
        key = tag if ws == 1 else f"{tag}@ws{ws}"        # multi-rank fixtures have floors of their own (tests/golden_util.py:grad_tol)
        if only and tag not in only and key not in only:
            continue
        m = measure(tag, seed, ws=ws)
        prev = out.get(key)
        out[key] = m if prev is None else {k: max(m[k], prev.get(k, 0.0)) for k in m}       # worst over the arch's seeds
        print(key, seed, m, flush=True)
        with open(path, "w") as f:
            json.dump(out, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
