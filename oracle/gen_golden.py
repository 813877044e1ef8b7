"""Generate tests/golden/*.npz from the REAL reference.  TEST INFRASTRUCTURE ONLY.

Run in the build container only:  ``python -m oracle.gen_golden``  (needs /root/reference).
Every fixture is *data*: inputs are re-derived from ``oracle/portable.py`` seeds (plus the fixture's guard band: the BatchNorm
bias values oracle/guard.py moved away from ReLU knife edges, kept in the file), the file keeps the
reference's outputs for one teacher-forced pretext step (free-running trajectories
are chaotic, so each "step" is its own seeded pre-step state with non-zero SGD momentum buffers,
non-zero queue_ptr and non-trivial BN running stats).
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from typing import Dict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
GOLDEN = os.path.join(ROOT, "tests", "golden")
sys.path.insert(0, ROOT)

from oracle import portable as P  # noqa: E402

# (arch, B per rank, H=W, K, world sizes, seeds)
CASES = [
    ("c3d", 4, 32, 64, (1, 2), 2),
    ("resnet18", 8, 64, 64, (1, 2), 1),
    ("r2plus1d-vcop", 4, 32, 64, (1, 2), 1),
    ("s3dg", 4, 64, 64, (1, 2), 1),
    ("c3d:mlp", 4, 32, 64, (1,), 1),      # fc_type='mlp' heads (moco/split_wrapper.py:171-179)
    ("resnet50", 8, 64, 64, (1,), 1),     # Bottleneck blocks (models/resnet.py:80-116): 1x1x1 convs, 2048-d head
    ("resnet34", 8, 64, 64, (1,), 1),     # [3,4,6,3] BasicBlocks (models/resnet.py:223-228)
    # the other projection heads (moco/split_wrapper.py:113-126) ...
    ("c3d:conv", 4, 32, 64, (1,), 1),
    ("c3d:convbn", 4, 32, 64, (1,), 1),
    ("c3d:speednet", 4, 32, 64, (1,), 1),
    # ... and the other entries of diff_speed=[4,2,1] (builder_diffspeed_diffloss.py:428-431): T_real = 8 / 32 frames
    ("c3d:linear:4", 4, 32, 64, (1, 2), 1),
    ("c3d:linear:1", 4, 32, 64, (1,), 1),
]
LR = 0.05
T_IN = 32


def split_arch(tag):
    """'c3d:mlp' -> ('c3d', 'mlp'); 'c3d:linear:4' additionally fixes the drawn speed; plain arch names use 'linear' heads."""
    parts = tag.split(":")
    return parts[0], (parts[1] if len(parts) > 1 else "linear")


def tag_speed(tag):
    """The entry of diff_speed the case replays for random.choice (default 2, the shipped configs' only entry)."""
    parts = tag.split(":")
    return int(parts[2]) if len(parts) > 2 else 2


def tag_file(tag):
    return tag.replace("-", "_").replace(":", "_")


def case_name(arch, ws, seed):
    return f"{tag_file(arch)}_ws{ws}_s{seed}"


def case_inputs(spec, arch, B, HW, K, ws, seed, nudges=None):
    """Everything a consumer needs to replay the step: state, momentum, clips, permutations.  `nudges`: the fixture's guard band
    (oracle/guard.py: {bias key: (channel indices, values)} written over fill_state's draw; tests/golden_util.py:load_case puts
    the fixture's into meta["nudges"])."""
    state = P.fill_state(spec, seed)
    if nudges:
        for key, (idx, val) in nudges.items():
            state[key][np.asarray(idx, dtype=np.int64)] = np.asarray(val, dtype=np.float32)
    state["queue_ptr"][:] = (B * ws * (seed % 3 + 1)) % K
    tk = [k for k in spec if k.startswith("encoder_q.") and not k.endswith(
        ("running_mean", "running_var", "num_batches_tracked"))]
    mom = P.fill_momentum([(k, spec[k][0]) for k in tk], seed)
    clips = [P.clips(seed, r, (B, 3, T_IN, HW, HW)) for r in range(ws)]
    perms_B = [P.permutation(f"perm:{r}", seed, B) for r in range(ws)]
    sh = (P.permutation("shuffle1", seed, B * ws), P.permutation("shuffle2", seed, B * ws))
    return state, mom, clips, perms_B, sh


def pack(res: Dict, rank: int, out: Dict[str, np.ndarray]):
    pre = f"r{rank}."
    for k in ("loss", "loss_A", "loss_M", "logits1", "logits2", "l_pos_M", "l_neg_M", "q_A", "q_M",
              "k_A_shuf", "k_M_shuf", "kneg_A_shuf", "kneg_M_shuf"):
        out[pre + k] = np.asarray(res[k])
    out[pre + "post.queue"] = res["post_state"]["queue"]
    out[pre + "post.queue_ptr"] = res["post_state"]["queue_ptr"]
    for k, v in res["post_state"].items():
        if k in ("queue", "queue_ptr"):
            continue
        if k.endswith("num_batches_tracked"):
            out[pre + "post." + k] = np.asarray(v)
        else:
            out[pre + "postsum." + k] = P.summarise(k, v)
            if k.startswith("encoder_q.") and not k.endswith(("running_mean", "running_var")):
                out[pre + "postproj." + k] = P.projections(k, v)
    for k, g in res["grads"].items():
        out[pre + "gradsum." + k] = np.zeros(0) if g is None else P.summarise(k, g)
        if g is not None:
            out[pre + "gradproj." + k] = P.projections(k, g)
    for k, v in res["momentum_post"].items():
        out[pre + "momsum." + k] = P.summarise(k, v)
        out[pre + "momproj." + k] = P.projections(k, v)


def nudges_from_npz(z):
    """{bias key: (indices, values)} of a fixture file's guard band (oracle/guard.py), or {}."""
    pre = "nudge.idx."
    return {name[len(pre):]: (np.asarray(z[name]), np.asarray(z["nudge.val." + name[len(pre):]])) for name in z.files if name.startswith(pre)}


def _load_nudges(tmpdir):
    path = os.path.join(tmpdir, "nudges.npz")
    if not os.path.exists(path):
        return None
    with np.load(path) as z:
        return nudges_from_npz(z)


def _worker(rank, ws, arch, B, HW, K, seed, port, tmpdir):
    import torch
    from oracle import ref_harness as R
    torch.set_num_threads(max(1, 8 // ws))
    R.ensure_process_group(rank, ws, port)
    model = R.build_reference_model(split_arch(arch)[0], K=K, fc_type=split_arch(arch)[1])
    spec = R.state_spec(model)
    state, mom, clips, perms_B, sh = case_inputs(spec, arch, B, HW, K, ws, seed, _load_nudges(tmpdir))
    res = R.run_reference_step(model, state, clips[rank][0], clips[rank][1], [perms_B[rank], sh[0], sh[1]],
                               tag_speed(arch), lr=LR, momentum_buffers=mom, ddp=(ws > 1))
    out: Dict[str, np.ndarray] = {}
    pack(res, rank, out)
    out[f"r{rank}.relu_margin"] = np.array(res["relu_margin"])
    out[f"r{rank}.pool_margin"] = np.array(res["pool_margin"])
    np.savez(os.path.join(tmpdir, f"r{rank}.npz"), **out)
    if rank == 0:
        with open(os.path.join(tmpdir, "spec.json"), "w") as f:
            json.dump({k: [list(s), d] for k, (s, d) in spec.items()}, f)


def reference_spec(arch, K):
    from oracle import ref_harness as R
    model = R.build_reference_model(split_arch(arch)[0], K=K, fc_type=split_arch(arch)[1])
    return dict(R.state_spec(model))


def guard_for(arch, spec, B, HW, K, ws, seed, verbose=False):
    """The guard band of one case (oracle/guard.py): settled on the restatement's fp64 query pass of every rank."""
    import torch
    from oracle import guard
    from oracle import restatement as S
    state, _, clips, perms_B, _ = case_inputs(spec, arch, B, HW, K, ws, seed)
    q = [S.diff_speed(torch.from_numpy(clips[r][0]), torch.from_numpy(clips[r][1]), torch.from_numpy(perms_B[r]), tag_speed(arch))[0].numpy()
         for r in range(ws)]
    return guard.guard_band(split_arch(arch)[0], split_arch(arch)[1], state, q, verbose=verbose)


def checker_grad_error(arch, meta, spec, out):
    """Worst relative L2 gradient distance (random-projection estimate) between the reference's gradients and the product's
    host logic run on the torch checker backend (tests/cpu_ops.py: channels-last convolutions, folded BatchNorm) — a third
    independent fp32 evaluation order, used like in gen_golden_finetune.py only to reject ill-conditioned seeds."""
    import torch
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_ops import CpuOps
    from model_util import run_model_step
    from rspnet_amd import ops
    inputs = case_inputs(spec, arch, meta["B"], meta["HW"], meta["K"], 1, meta["seed"], meta.get("nudges"))
    prev = ops.set_backend(CpuOps())
    try:
        _, _, _, grads = run_model_step(arch, meta, inputs, 0, torch.device("cpu"), "fused")
    finally:
        ops.set_backend(prev)
    worst = 0.0
    for k, g in grads.items():
        ref = out["r0.gradproj." + k] if "r0.gradproj." + k in out else None
        if g is not None and ref is not None and float(out["r0.gradsum." + k][0]) >= 1e-4:
            worst = max(worst, P.proj_rel_err(k, g, ref))
    return worst


def run_case(arch, B, HW, K, ws, seed, nudges=None):
    import torch.multiprocessing as mp
    from oracle.ref_harness import _free_port
    with tempfile.TemporaryDirectory() as tmp:
        port = _free_port()
        if nudges:
            np.savez(os.path.join(tmp, "nudges.npz"), **{f"nudge.idx.{k}": v[0] for k, v in nudges.items()},
                     **{f"nudge.val.{k}": v[1] for k, v in nudges.items()})
        mp.spawn(_worker, args=(ws, arch, B, HW, K, seed, port, tmp), nprocs=ws, join=True)
        out: Dict[str, np.ndarray] = {}
        for r in range(ws):
            with np.load(os.path.join(tmp, f"r{r}.npz")) as z:
                out.update({k: z[k] for k in z.files})
        with open(os.path.join(tmp, "spec.json")) as f:
            spec = json.load(f)
    for k, (idx, val) in (nudges or {}).items():
        out[f"nudge.idx.{k}"] = np.asarray(idx, dtype=np.int32)
        out[f"nudge.val.{k}"] = np.asarray(val, dtype=np.float32)
    out["meta"] = np.array(json.dumps(dict(arch=split_arch(arch)[0], fc_type=split_arch(arch)[1], B=B, HW=HW, K=K, ws=ws, seed=seed, lr=LR, speed=tag_speed(arch),
                                           T_in=T_IN, m=0.999, T=0.07, sgd_momentum=0.9, weight_decay=1e-4,
                                           margin=2.0, A=1.0, M=1.0)))
    return out, spec


# Seed acceptance (on top of the guard band, which every seed gets):
#   * the band holds in fp32: smallest |ReLU input| / channel sigma in the restatement's fp32 forward, in units of each ReLU's
#     band, and in the REFERENCE's own forward, in units of the band's floor (oracle/guard.py:band_eps), both >= RELU_CHECK
#     (asserted: the band was settled on the restatement in fp64; 1.0 = at the band's edge);
#   * max-pool arg-max of the small disjoint-window layers: top-2 gap >= POOL_MARGIN (a bias cannot open it: seed selection);
#   * the ranking hinge max(0, margin - (p - n)) is not within HINGE_MARGIN of its kink for any sample;
#   * conditioning screen: the oracle restatement evaluated in two other fp32 orders (native convolution; folded scale/shift
#     BatchNorm with fp32 statistics partials) and the product's host logic on the torch checker backend reproduce the
#     fixture's gradients to SCREEN_TOL (worst tensor, relative L2).  What is left after the guard band are arg-max decisions
#     of the overlapping / large-layer pools; the screen keeps the seeds on which none of them sits within rounding of a tie.
#     Only CPU evaluations of the oracle and of the reference take part, never a GPU result.
# The first seed that meets all of them is kept; if none of MAX_SEEDS does, the one with the smallest screen value among those
# that meet the hard limits (POOL_FLOOR, HINGE_MARGIN).
RELU_CHECK = 0.6
POOL_MARGIN = 1.5e-5
POOL_FLOOR = 3e-6
HINGE_MARGIN = 1e-3
SCREEN_TOL = 3e-4
MAX_SEEDS = {"c3d": 10, "s3dg": 2}      # (an S3D-G seed takes ~10 minutes: 77 units to settle, 4 evaluations of the screen)
MAX_SEEDS_DEFAULT = 6


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    only = sys.argv[1:] or None
    index_path = os.path.join(GOLDEN, "index.json")
    index = json.load(open(index_path)) if os.path.exists(index_path) else []
    for arch, B, HW, K, wss, nseeds in CASES:
        if only and arch not in only:
            continue
        for e in [e for e in (json.load(open(index_path)) if os.path.exists(index_path) else []) if e[0] == arch]:
            stale = os.path.join(GOLDEN, case_name(*e) + ".npz")
            if os.path.exists(stale):
                os.remove(stale)
        index = [e for e in index if e[0] != arch and (not only or e[0] in only)]
        spec0 = {k: (tuple(s_), d) for k, (s_, d) in reference_spec(arch, K).items()}
        max_seeds = MAX_SEEDS.get(split_arch(arch)[0], MAX_SEEDS_DEFAULT) + nseeds - 1
        for ws in wss:
            seed, kept = 0, 0
            fallback = []          # (screen value, seed, out, spec): seeds inside the hard limits that missed a soft one
            while kept < nseeds:
                seed += 1
                if seed > max_seeds:
                    assert len(fallback) >= nseeds - kept, f"{arch} ws{ws}: no acceptable seed among {max_seeds}"
                    fallback.sort(key=lambda t: t[0])
                    for worst, sd_, out, spec in fallback[:nseeds - kept]:
                        print(f"{arch} ws{ws}: keeping seed {sd_}, the best of {max_seeds}: screen value {worst:.2e}", flush=True)
                        _write(arch, ws, sd_, out, spec, index, index_path)
                        kept += 1
                    break
                nudges, rep = guard_for(arch, spec0, B, HW, K, ws, seed)
                out, spec = run_case(arch, B, HW, K, ws, seed, nudges)
                relu = min(float(out[f"r{r}.relu_margin"]) for r in range(ws))
                pool = min(float(out[f"r{r}.pool_margin"]) for r in range(ws))
                hinge = min(float(np.abs(2.0 - (out[f"r{r}.l_pos_M"] - out[f"r{r}.l_neg_M"])).min()) for r in range(ws))
                print(f"{arch} ws{ws} seed {seed}: guard {rep}; reference: relu margin {relu:.2f} bands, pool gap {pool:.2e}, hinge {hinge:.2e}",
                      flush=True)
                assert relu >= RELU_CHECK and rep["fp32_margin_in_bands"] >= RELU_CHECK, "the guard band settled on the restatement does not hold in fp32"
                if pool < POOL_FLOOR or hinge < HINGE_MARGIN:
                    print(f"skip {arch} ws{ws} seed {seed}: arg-max / hinge knife edge", flush=True)
                    continue
                from oracle.gen_conditioning import measure
                meta = json.loads(str(out["meta"]))
                meta["nudges"] = nudges
                spec_t = {k: (tuple(s_), d) for k, (s_, d) in spec.items()}
                m = measure(arch, seed, meta=meta, spec=spec_t, fast=True, ws=ws)
                worst = m["grad_rel_l2_max"]
                if ws == 1:
                    worst = max(worst, checker_grad_error(arch, meta, spec_t, out))
                print(f"     screen: worst tensor {worst:.2e} between fp32 evaluation orders", flush=True)
                if worst > SCREEN_TOL or pool < POOL_MARGIN:
                    fallback.append((worst * (1.0 if pool >= POOL_MARGIN else 3.0), seed, out, spec))
                    continue
                _write(arch, ws, seed, out, spec, index, index_path)
                kept += 1


def extra(arch, ws, seeds):
    """``python -m oracle.gen_golden --extra s3dg 2 2,3``: ADDITIONAL candidate fixtures of one (family, world size) for the given
    seeds, next to the ones on disk (same guard band, no screening: the caller decides which to keep — e.g. by how the HIP path
    lands on them, which only a GPU box can tell)."""
    index_path = os.path.join(GOLDEN, "index.json")
    case = next(c for c in CASES if c[0] == arch)
    _, B, HW, K, _, _ = case
    spec0 = {k: (tuple(s_), d) for k, (s_, d) in reference_spec(arch, K).items()}
    for seed in seeds:
        index = json.load(open(index_path))
        if [arch, ws, seed] in index:
            continue
        nudges, rep = guard_for(arch, spec0, B, HW, K, ws, seed)
        out, spec = run_case(arch, B, HW, K, ws, seed, nudges)
        relu = min(float(out[f"r{r}.relu_margin"]) for r in range(ws))
        pool = min(float(out[f"r{r}.pool_margin"]) for r in range(ws))
        print(f"{arch} ws{ws} seed {seed}: guard {rep}; reference: relu margin {relu:.2f} bands, pool gap {pool:.2e}", flush=True)
        assert relu >= RELU_CHECK and rep["fp32_margin_in_bands"] >= RELU_CHECK
        _write(arch, ws, seed, out, spec, index, index_path)


def _write(arch, ws, seed, out, spec, index, index_path):
    name = case_name(arch, ws, seed)
    np.savez_compressed(os.path.join(GOLDEN, name + ".npz"), **out)
    with open(os.path.join(GOLDEN, f"state_spec_{tag_file(arch)}.json"), "w") as f:
        json.dump(spec, f, indent=0)
    index.append([arch, ws, seed])
    print("wrote", name, "loss", out["r0.loss"], flush=True)
    # (two generator processes may work on different architectures side by side: merge with what is on disk)
    on_disk = json.load(open(index_path)) if os.path.exists(index_path) else []
    mine = {e[0] for e in index}
    merged = [e for e in on_disk if e[0] not in mine] + [e for e in index if e[0] in mine and os.path.exists(os.path.join(GOLDEN, case_name(*e) + ".npz"))]
    with open(index_path, "w") as f:
        json.dump(sorted(merged), f)


if __name__ == "__main__" and len(sys.argv) > 1 and sys.argv[1] == "--extra":
    extra(sys.argv[2], int(sys.argv[3]), [int(x) for x in sys.argv[4].split(",")])
    sys.exit(0)
if __name__ == "__main__":
    main()
