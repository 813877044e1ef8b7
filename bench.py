#!/usr/bin/env python
"""bench.py — clips/sec of the RSPNet pretext step (BASELINE.json metric) on N MI355X of one node.

A step = momentum update + diff-speed gather + 2 key-encoder passes (shuffle-BN) + query forward/backward +
InfoNCE/ranking losses + enqueue + gradient all-reduce + SGD, on synthetic clips already resident in HBM.
Workload at every N: BASELINE configs[1] — C3D, B=32 clips per GPU, model input (32,3,32,112,112) -> encoder input
3x16x112x112, K=16384, dim=128, T=0.07, m=0.999, fp32 (weak scaling).  One process per GPU (torch.distributed/RCCL).

Launching.  `python bench.py --gpus N` starts the N ranks itself: the parent — which never touches a GPU — picks a free
port and starts N child interpreters of this file with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / MASTER_PORT
set (what /root/reference/pretrain.py:278-283,335-336 does with mp.spawn + a tcp://127.0.0.1 rendezvous), waits for them and
exits non-zero if any rank failed.  Under `python -m torch.distributed.run ... bench.py --gpus N` (WORLD_SIZE already set)
the process is a rank and runs directly.

Prints ONE JSON line on rank 0 (see the driver contract): value = whole-job clips/s; plus
  roofline     — the conv MFMA launches grouped by the kernel template the library dispatched (rsp_conv3d_kernel_name):
                 algorithmic FLOPs / HIP-event time on the launch stream against the fp32-input MFMA peak (157.3 TFLOP/s,
                 MI355X_MICROARCH.md); the block's headline is the kernel with the largest share of the step for THIS arch;
  cpu_baseline — the oracle restatement (oracle/restatement.py, proven equal to the reference) timed on this host's
                 physical cores on BASELINE config 1 (32 clips; 1 warm-up + 2 timed steps) — rank 0, N=1 only;
  parity       — the CPU leg's warm-up step REPLAYS the GPU run's first step (same pre-step state, clips, permutations):
                 relative errors of loss / logits / features / queue slab / whole gradient at the measured size;
  steps_ms     — p50 / min / max / p90 of the GPU-side per-step intervals + the host's enqueue time per step;
  comm_ms      — N > 1: per-step stall of this rank's compute stream behind each collective;
  other_workloads — N = 1 default run: BASELINE configs 3-5 (R3D-18, R(2+1)D B=32; S3D-G B=16 at 224^2), 10 + 30 steps each.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_F32_MFMA_TFLOPS = 157.3
ARCHS = {
    # arch: (per-GPU batch, H=W, base lr from config/pretrain/*.jsonnet)
    "c3d": (32, 112, 0.1),
    "resnet18": (32, 112, 0.1),
    "r2plus1d-vcop": (32, 112, 0.05),
    "s3dg": (16, 224, 0.05),
}


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--arch", default="c3d", choices=sorted(ARCHS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--hw", type=int, default=None, help="clip height=width (default: the BASELINE config of --arch)")
    ap.add_argument("--queue", type=int, default=16384, help="MoCo K before the world-size trim")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=32, help="clips per CPU-baseline step (BASELINE config 1: 32)")
    ap.add_argument("--cpu-steps", type=int, default=2, help="timed CPU-baseline steps after one warm-up step")
    ap.add_argument("--graph", choices=("auto", "on", "off"), default="auto",
                    help="the step as one replayed HIP graph (rspnet_amd/graph_step.py; N=1 only).  auto: only when the host cannot "
                         "issue the step fast enough (measured in the warm-up: S3D-G's 1 800 launches) — replayed, a step the host "
                         "keeps ahead of is 0.5-2.4 %% slower than issued eagerly with its side streams; on: always; off: never")
    ap.add_argument("--no-other-workloads", dest="other_workloads", action="store_false",
                    help="skip BASELINE configs 3-5 (R3D-18, R(2+1)D, S3D-G) that the default N=1 C3D run appends")
    ap.add_argument("--other-steps", type=int, default=30)
    ap.add_argument("--other-warmup", type=int, default=10)
    ap.add_argument("--deadline", type=float, default=float(os.environ.get("RSP_BENCH_DEADLINE", 1500)),
                    help="self-launcher only: seconds after which a still-running job is killed and rc=124 returned")
    ap.add_argument("--collective-timeout", type=float, default=300.0,
                    help="process-group timeout (rendezvous and every collective), seconds")
    ap.add_argument("--force-dp", action="store_true",
                    help="--gpus 1 only: initialise the real nccl (RCCL) process group with ONE rank and run the data-parallel step — "
                         "clip all-to-all, fused key all-gather, bucketed gradient all-reduce, gloo side group — as N > 1 runs it "
                         "(MoCoDiffLossTwoFc(force_collectives=True)); the line carries comm_ms")
    ap.add_argument("--grad-floor", choices=("static", "live"), default="static",
                    help="parity block: conditioning floor of the whole gradient on the replayed state = the oracle's own fp32 gradient "
                         "against its fp64 gradient.  live: recompute it (a 150 s fp64 replay on the CPU); static: take the committed value "
                         "for this (seeded, hence identical) state from profiles/grad_floor.json after checking an input fingerprint")
    ap.add_argument("--eager-steps", type=int, default=10,
                    help="N=1 graphed runs: also time this many eagerly issued steps (the way N > 1 issues them) after the timed region")
    ap.add_argument("--seed", type=int, default=1234, help="seed of the synthetic state and clips (the default run's is 1234)")
    ap.add_argument("--selftest-hang-rank", type=int, default=-1, help=argparse.SUPPRESS)
    ap.add_argument("--selftest-parity", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--selftest-cpu", action="store_true",
                    help="launcher / host-logic self-test without a GPU: gloo + the tests' checker op backend on tiny clips; "
                         "the printed line says data='selftest-cpu' and is not a measurement")
    return ap.parse_args(argv)


# ----------------------------------------------------------------------------------------------------------------------
# self-launcher (parent process: no GPU call anywhere on this path)
# ----------------------------------------------------------------------------------------------------------------------
def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch(n: int, argv, deadline_s: float) -> int:
    """Start n rank processes of this file, wait, return the job's exit code: non-zero if any rank failed, or if the job is
    still running `deadline_s` seconds after the start (a rank stuck in a collective, a dead peer) — every rank is killed then,
    so the caller gets a diagnosable failure instead of a node held until its own limit."""
    port = _free_port()
    t_start = time.monotonic()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(n), "LOCAL_WORLD_SIZE": str(n),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port)})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC (RCCL across processes on this driver)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env))
    rc = 0
    try:
        live = list(procs)
        while live:
            time.sleep(0.05)
            if time.monotonic() - t_start > deadline_s:
                print(f"bench.py: job exceeded its deadline of {deadline_s:.0f} s with ranks "
                      f"{[procs.index(p) for p in live]} still running; killing all ranks", file=sys.stderr)
                rc = 124
                break
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code if code > 0 else 1
                    print(f"bench.py: rank {procs.index(p)} exited with {code}; stopping the other ranks", file=sys.stderr)
                    for q in live:                                  # the survivors would wait in a collective forever
                        q.terminate()
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    return rc


# ----------------------------------------------------------------------------------------------------------------------
# CPU baseline (checker code, never the product path)
# ----------------------------------------------------------------------------------------------------------------------
def host_cpu():
    """(model string, physical cores usable by this process)."""
    model, pairs, phys, core = "unknown", set(), None, None
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and model == "unknown":
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("physical id"):
                    phys = line.split(":", 1)[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":", 1)[1].strip()
                elif not line.strip():
                    if phys is not None and core is not None:
                        pairs.add((phys, core))
                    phys = core = None
    except OSError:
        pass
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return model, max(1, min(len(pairs) or avail, avail))


def rank_cpu_set(local_rank: int, local_ws: int):
    """The logical CPUs rank `local_rank` of `local_ws` ranks on this node keeps to itself: the CPUs this process may use, ordered
    by (package, physical core, hyper-thread), cut into `local_ws` contiguous slices of whole physical cores — eight interpreters
    plus their RCCL proxy threads on shared cores is the first-run hazard of a multi-GPU job (pretrain.py:278-283 spawns one
    process per GPU and leaves placement to the OS).  Ranks 0..ws/2-1 land on the first package, the rest on the second, which is
    how the GPUs of an 8 x MI355X node hang off its two sockets."""
    avail = sorted(os.sched_getaffinity(0))
    if local_ws <= 1 or len(avail) < local_ws:
        return None

    def topo(cpu):
        base = f"/sys/devices/system/cpu/cpu{cpu}/topology/"
        try:
            with open(base + "physical_package_id") as f:
                pkg = int(f.read())
            with open(base + "core_id") as f:
                core = int(f.read())
            return pkg, core
        except (OSError, ValueError):
            return 0, cpu

    cores = {}
    for c in avail:
        cores.setdefault(topo(c), []).append(c)
    phys = [cores[k] for k in sorted(cores)]
    per = len(phys) // local_ws
    if per < 1:
        return None
    mine = phys[local_rank * per:(local_rank + 1) * per]
    return sorted(c for grp in mine for c in grp)


def _fingerprint(state, im_q, im_k, first):
    """Cheap identity of what the replayed step consumes (the bench's state and clips are seeded: the same numbers on every box)."""
    import torch
    acc = float(im_q.double().sum()) * 3.0 + float(im_k.double().abs().sum())
    for k in sorted(state):
        if state[k].dtype == torch.float32:
            acc += float(state[k].double().abs().sum())
    perm = first[0].tolist() + first[1][0].tolist() + first[1][1].tolist() + [int(first[2])]
    h = 0
    for i, v in enumerate(perm):
        h = (h * 1000003 + int(v) + i) % (1 << 61)
    return {"sum": acc, "draws": h}


def cpu_baseline(arch, hw, sample_b, steps, K, lr, parity=None, grad_floor="static"):
    """Oracle restatement timed on this host (reported baseline).  With `parity` = what the GPU's first step consumed and
    produced (bench.py:parity_capture), the untimed warm-up step replays THAT step — same pre-step state, clips, diff-speed
    permutation and shuffle permutations — and the line gets a "parity" object: BASELINE config 2's "loss match vs
    reference" re-proved on every bench run at the measured size (loss / logits within 1e-3 relative is the north-star bar)."""
    import numpy as np
    import torch
    from oracle import portable as P
    from oracle import restatement as S
    model, cores = host_cpu()
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(0)
    par = None
    if parity is not None:
        state = parity["state"]
        im_q, im_k = parity["im_q"], parity["im_k"]
        sample_b = im_q.shape[0]
        first = (parity["perm"], (parity["sh1"], parity["sh2"]), parity["speed"])
    else:
        with open(os.path.join(ROOT, "tests", "golden", f"state_spec_{arch.replace('-', '_')}.json")) as f:
            spec = {k: (tuple(s), d) for k, (s, d) in json.load(f).items()}
        spec["queue"] = ((128, K), "float32")
        state = {k: torch.from_numpy(v) for k, v in P.fill_state(spec, 1).items()}
        im_q = torch.randn(sample_b, 3, 32, hw, hw, generator=g)
        im_k = torch.randn(sample_b, 3, 32, hw, hw, generator=g)
        first = None
    moms = [{}]
    times = []
    state0_fp = {k: v.clone() for k, v in state.items()} if first is not None else None
    g64 = None
    static_floor = None
    if first is not None and grad_floor == "static":
        fpath = os.path.join(ROOT, "profiles", "grad_floor.json")
        key = f"{arch}_b{sample_b}_hw{hw}_k{K}"
        if os.path.exists(fpath):
            with open(fpath) as f:
                ent = json.load(f).get(key)
            fp = _fingerprint(state, im_q, im_k, first)
            if ent is not None and ent["fingerprint"]["draws"] == fp["draws"] and \
                    abs(ent["fingerprint"]["sum"] - fp["sum"]) <= 1e-6 * abs(fp["sum"]):
                static_floor = ent
    t64 = 0.0
    if first is not None and static_floor is None:
        # the same replay in fp64 first (on a copy of the pre-step state): the whole gradient has a conditioning floor — ReLU / max-pool
        # decisions flip under fp32 rounding — that is a property of the STATE, not of the kernels; it is measured here as the distance
        # of the oracle's own fp32 gradient from its fp64 gradient, and the GPU's gradient is held against the same fp64 gradient
        t64 = time.perf_counter()
        torch.set_default_dtype(torch.float64)
        try:
            st64 = {k: (v.clone().double() if v.dtype == torch.float32 else v.clone()) for k, v in state.items()}
            o64 = S.moco_step(arch, [st64], [im_q.double()], [im_k.double()], [first[0]], first[1], first[2], K=K, lr=lr,
                              momentum_buffers=[{}])[0]
            g64 = {k: v.detach().clone() for k, v in o64["grads"].items() if v is not None}
            del st64, o64
        finally:
            torch.set_default_dtype(torch.float32)
        t64 = time.perf_counter() - t64
    for it in range(1 + steps):                      # step 0 = warm-up (allocator, oneDNN primitive caches)
        if it == 0 and first is not None:
            perm, sh, speed = first
        else:
            perm = torch.randperm(sample_b, generator=g)
            sh = (torch.randperm(sample_b, generator=g), torch.randperm(sample_b, generator=g))
            speed = 2
        t0 = time.perf_counter()
        outs = S.moco_step(arch, [state], [im_q], [im_k], [perm], sh, speed, K=K, lr=lr, momentum_buffers=moms)
        times.append(time.perf_counter() - t0)
        if it == 0 and first is not None:
            o, got = outs[0], parity["out"]

            def rel(a, b):
                a, b = a.double(), b.double()
                return float((a - b).abs().max() / b.abs().max().clamp_min(1e-5))

            def whole_l2(ga, gb):      # relative L2 distance of two whole gradients (one vector over all tensors)
                num = den = 0.0
                for k, gr in gb.items():
                    if gr is not None and ga.get(k) is not None:
                        dd = ga[k].double() - gr.double()
                        num, den = num + float((dd * dd).sum()), den + float((gr.double() ** 2).sum())
                return (num / den) ** 0.5 if den > 0 else None

            grad_vs_oracle = whole_l2(parity["grads"], o["grads"])
            floor = whole_l2(o["grads"], g64) if g64 is not None else None
            grad_vs_fp64 = whole_l2(parity["grads"], g64) if g64 is not None else None
            floor_src = "live: fp64 replay of this step on the oracle"
            if static_floor is not None:
                floor, floor_src = static_floor["grad_floor_rel_l2"], static_floor["source"]
            elif g64 is not None:
                print("bench.py: grad floor entry for profiles/grad_floor.json: " + json.dumps(
                    {f"{arch}_b{sample_b}_hw{hw}_k{K}": {"grad_floor_rel_l2": floor, "grad_vs_fp64_rel_l2": grad_vs_fp64,
                                                        "fingerprint": _fingerprint(state0_fp, im_q, im_k, first)}}), file=sys.stderr)
            ptr0 = parity["ptr0"]
            par = {"vs": "oracle/restatement.py:moco_step (pinned to the reference) replaying the GPU run's first step: same "
                         "pre-step state, clips, diff-speed and shuffle permutations",
                   "loss_rel": rel(got["loss"], o["loss"]), "loss_A_rel": rel(got["loss_A"], o["loss_A"]),
                   "loss_M_rel": rel(got["loss_M"], o["loss_M"]),
                   "logits_rel": max(rel(got["logits1"], o["logits1"]), rel(got["logits2"], o["logits2"])),
                   "ranking_logits_rel": max(rel(got["l_pos_M"], o["l_pos_M"]), rel(got["l_neg_M"], o["l_neg_M"])),
                   "features_rel": max(rel(got["q_A"], o["q_A"]), rel(got["q_M"], o["q_M"])),
                   "queue_slab_rel": rel(got["queue_slab"], state["queue"][:, ptr0:ptr0 + sample_b]),
                   "grad_rel_l2": grad_vs_oracle,
                   # conditioning floor of this state: oracle fp32 vs oracle fp64; the GPU gradient against the same fp64 gradient
                   "grad_floor_rel_l2": floor, "grad_floor_source": floor_src, "grad_floor_static": static_floor is not None,
                   "grad_vs_fp64_rel_l2": grad_vs_fp64,
                   "grad_gate": "grad_rel_l2 <= max(3 x grad_floor_rel_l2, 1e-4) — the rule of tests/golden_util.py:grad_tol: two correct "
                                "fp32 evaluations of this state differ by the floor each from the exact gradient (ReLU / max-pool "
                                "decisions that flip under rounding), so by up to ~2 floors from one another; 1e-4 is the summation-order "
                                "noise of fp32 under which a floor (5e-6 on tiny well-conditioned states) stops being a scale",
                   "fp64_replay_s": round(t64, 1) if g64 is not None else None,
                   "loss_gpu": float(got["loss"]), "loss_oracle": float(o["loss"]), "tolerance": 1e-3}
            par = {k: (float(f"{v:.3e}") if isinstance(v, float) and k.endswith(("_rel", "_l2")) else v) for k, v in par.items()}
            par["forward_ok"] = bool(all(par[k] <= 1e-3 for k in ("loss_rel", "loss_A_rel", "loss_M_rel", "logits_rel",
                                                                  "ranking_logits_rel", "features_rel", "queue_slab_rel")))
            par["grad_ok"] = None if floor is None else bool(grad_vs_oracle <= max(3.0 * floor, 1e-4))
            par["ok"] = par["forward_ok"] and par["grad_ok"] is not False
    timed = times[1:] or times
    dt = sum(timed) / len(timed)
    res = {"value": round(sample_b / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port", "cpu_model": model,
           "s_per_step": round(dt, 2),
           "sample": f"{arch} full pretext step (2 key passes + query fwd/bwd + losses + SGD) on {sample_b} synthetic clips "
                     f"3x32x{hw}x{hw}, K={K}; 1 warm-up step ({times[0]:.1f} s"
                     f"{', the replay of the GPU run first step' if first is not None else ''}) + mean of {len(timed)} timed; torch "
                     f"{torch.__version__} CPU ops, {cores} threads = physical cores of {model}"}
    return res, par


# ----------------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------------
def load_traffic(arch, B, kernel):
    """(L2-miss GB per launch, matrix-pipe busy share, source file) of `kernel` from the committed PMC passes
    (profiles/traffic.json, written by tools/summarize_profiles.py) — STATIC: counters cannot be collected inside a bench run."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None, None, None
    with open(tpath) as f:
        t = json.load(f)
    arch_ent = t.get(f"{arch}_b{B}", {})
    ent = arch_ent.get(kernel)
    if ent is None:
        return None, None, None, None
    # the counters belong to the build they were collected on: stale when the kernel sources have changed since
    from rspnet_amd import _lib
    built = (arch_ent.get("_build") or {}).get("csrc_sha256")
    stale = {"traffic_stale": built != _lib.source_hash(), "profiled_csrc_sha256": built, "profiled_commit": (arch_ent.get("_build") or {}).get("commit")}
    return round(ent["hbm_bytes_per_launch"] / 1e9, 4), ent.get("mfma_busy"), ent.get("source"), stale


def _pct(xs, q):
    xs = sorted(xs)
    return xs[min(len(xs) - 1, int(round(q * (len(xs) - 1))))] if xs else None


def measure(args, arch, B, hw, base_lr, steps, warmup, dev, rank, ws, want_parity=False):
    """Build the pretext model of `arch`, run `warmup` untimed + `steps` timed steps, return (result dict, parity capture)."""
    import torch
    import torch.distributed as dist
    from rspnet_amd import ops
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD

    cuda = dev.type == "cuda"
    K = args.queue // (B * ws) * (B * ws)                  # utils/moco.py:8-10 trim
    cfg = {"model": {"arch": arch},
           "moco": {"dim": 128, "k": K, "m": 0.999, "t": 0.07, "fc_type": "linear", "diff_speed": [2]}}
    torch.manual_seed(args.seed)
    model = ModelFactory(cfg).build_moco_diffloss(device=dev, force_collectives=True if args.force_dp else None)
    model.train()
    coll = bool(model.module._dp()[2])
    crit = Loss(margin=2.0, A=1.0, M=1.0)
    lr = base_lr * ws * B / 64                            # framework/utils/environment.py:13-16
    opt = SGD(model.parameters(), lr=lr, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)  # pretrain.py:65-72
    g = torch.Generator(device=dev).manual_seed(args.seed + rank)
    im_q = torch.randn(B, 3, 32, hw, hw, device=dev, generator=g)
    im_k = torch.randn(B, 3, 32, hw, hw, device=dev, generator=g)

    # with the collectives on (N > 1, --force-dp) the stepper replays the step as HIP-graph SEGMENTS between its collective points
    use_graph = cuda and args.graph in ("on", "auto")
    stepper = None
    if use_graph:
        from rspnet_amd.graph_step import GraphedPretextStep
        stepper = GraphedPretextStep(model, crit, opt, warmup=2, issue="graph" if args.graph == "on" else "auto")
        # the synthetic clips live in the stepper's static clip buffers (resident in HBM before the timed region, as everywhere in
        # this file): a replayed step then reads them in place instead of copying 2 x B clips per step
        bq, bk = stepper.clip_buffers(im_q, im_k)
        bq.copy_(im_q)
        bk.copy_(im_k)
        im_q, im_k = bq, bk

    def eager_step():
        out, tgt, rl, rt = model(im_q, im_k)
        loss, loss_A, loss_M = crit(out, tgt, rl, rt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss, loss_A, loss_M, out, rl

    def step():
        # the reference's loop body (pretrain.py:157-165) — eagerly, or as ONE replayed HIP graph (rspnet_amd/graph_step.py)
        return stepper(im_q, im_k) if stepper is not None else eager_step()

    def fence():
        if ws > 1:
            dist.barrier()
        if cuda:
            torch.cuda.synchronize()

    capture = None
    if want_parity:
        # one extra untimed step in front of the warm-up, replayed afterwards on the CPU oracle (cpu_baseline)
        inner = model.module
        state = {k: v.detach().cpu().clone() for k, v in inner.state_dict().items()}
        ptr0 = int(state["queue_ptr"])
        loss, loss_A, loss_M, out, rl = step()
        names = {id(p): n for n, p in inner.named_parameters()}
        perm, speed, sh1, sh2 = inner._last_draw
        q_A, q_M = inner._last_q
        capture = {"state": state, "im_q": im_q.cpu(), "im_k": im_k.cpu(), "perm": perm.cpu(), "speed": int(speed),
                   "sh1": torch.from_numpy(sh1.copy()), "sh2": torch.from_numpy(sh2.copy()), "ptr0": ptr0, "K": K, "lr": lr,
                   "grads": {names[id(p)]: p.grad.detach().cpu().clone() for p in inner.encoder_q.parameters()
                             if p.grad is not None},
                   "out": {"loss": loss.detach().cpu(), "loss_A": loss_A.cpu(), "loss_M": loss_M.cpu(),
                           "logits1": out[0].detach().cpu(), "logits2": out[1].detach().cpu(), "l_pos_M": rl[0].detach().cpu(),
                           "l_neg_M": rl[1].detach().cpu(), "q_A": q_A.cpu(), "q_M": q_M.cpu(),
                           "queue_slab": inner.queue[:, ptr0:ptr0 + B].cpu()}}
        del loss, loss_A, loss_M, out, rl

    for _ in range(warmup):
        step()
    be = ops.backend()
    inner = model.module
    # everything alive now (modules, plans, descriptor caches) is long-lived: move it out of the collector's reach so that a
    # full collection in the timed region — seen as one 40-75 ms host pause per run — has nothing to scan
    import gc
    gc.collect()
    gc.freeze()
    fence()
    graphed = stepper is not None and not stepper.disabled and len(stepper.graphs) > 0
    if graphed and cuda:
        # where the GPU's time of a replayed step goes: a few steps with an event pair around every graph (outside the timed region)
        stepper.segment_gpu_events = []
        for _ in range(3):
            step()
        fence()
        gev, stepper.segment_gpu_events = stepper.segment_gpu_events, None
        seg_gpu = {}
        for name, e0, e1 in gev:
            seg_gpu.setdefault(name, []).append(e0.elapsed_time(e1))
        # ... and between one graph's end and the next graph's start (consecutive replays in issue order; graphs of different lanes
        # overlap, so a negative value means "started before the previous one ended")
        for (n0, _, e1), (n1, e0, _) in zip(gev[:-1], gev[1:]):
            if not n1.startswith("main:top"):
                seg_gpu.setdefault(f"gap {n0} -> {n1}", []).append(e1.elapsed_time(e0))
        seg_gpu = {k: round(_pct(v, 0.5), 3) for k, v in seg_gpu.items()}
    else:
        seg_gpu = None
        for _ in range(3):      # (the same number of steps in every issue mode: `final_loss` of a seeded run is then comparable
            step()              #  between --graph off / on / auto — the replayed step is the eager step, bit for bit)
        fence()
    # (per-launch events are not recorded in the timed region: a replayed graph has none, and the eager step runs its independent
    #  passes on side streams, where a launch's interval also holds its neighbours' time — see the roofline pass below)
    if coll:
        inner.comm_log = {}
    marks, host, hbm, waits = [], [], [], []
    if graphed and stepper.mode != "whole":
        stepper.segment_host_ms = {}
    if cuda:
        marks.append(torch.cuda.Event(enable_timing=True))
        marks[0].record()
    t0 = time.perf_counter()
    for _ in range(steps):
        h0 = time.perf_counter()
        loss = step()[0]
        host.append((time.perf_counter() - h0) * 1e3)       # time the host needs to ENQUEUE a step (it runs ahead of the GPU)
        waits.append(getattr(stepper, "last_wait_s", 0.0) * 1e3 if stepper is not None else 0.0)
        if cuda:
            marks.append(torch.cuda.Event(enable_timing=True))
            marks[-1].record()
    fence()
    dt = time.perf_counter() - t0
    seg_host = None
    if stepper is not None and stepper.segment_host_ms is not None:
        seg_host, stepper.segment_host_ms = stepper.segment_host_ms, None
    log = []
    comm, inner.comm_log = inner.comm_log, None
    final_loss = float(loss.detach())
    # what issuing ONE step costs the host when nothing holds it up: the queue is empty at the start of each sample (inside the
    # timed loop a call also waits whenever the hardware queue is full — GPU time, not submission cost)
    idle_issue = []
    if cuda:
        for _ in range(5):
            fence()
            h0 = time.perf_counter()
            step()
            idle_issue.append((time.perf_counter() - h0) * 1e3)
        fence()
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    step_ms = dt / steps * 1e3
    res = {"clips_per_s": ws * B * steps / dt, "ms_per_step": step_ms, "final_loss": final_loss, "K": K, "lr": lr, "B": B,
           "hw": hw, "graph": bool(graphed), "collectives": coll,
           "issue_mode": {"whole": "graph", "segments": "graph_segments", "lanes": "graph_lanes"}[stepper.mode] if graphed else "eager"}
    if cuda:
        # worst rank's host time to issue one step (Python + launches + collective calls, without the stepper's back-pressure wait):
        # the number that says whether a rank is host-bound
        hs = torch.tensor([_pct([h - w for h, w in zip(host, waits)], 0.5), _pct(idle_issue, 0.5)], dtype=torch.float64, device=dev)
        if ws > 1:
            dist.all_reduce(hs, op=dist.ReduceOp.MAX)
        res["host_submit_p50_max_over_ranks"] = round(float(hs[0].item()), 3)
        res["host_issue_idle_gpu_p50_max_over_ranks"] = round(float(hs[1].item()), 3)
    if graphed and args.eager_steps > 0:
        # the same step issued eagerly with its side streams — how a run with more than one rank issues it (RCCL collectives are
        # not captured): the N = 1 point of a scaling curve in the N > 1 issue mode
        eager_step()
        fence()
        te = time.perf_counter()
        for _ in range(args.eager_steps):
            eager_step()
        fence()
        res["eager_ms_per_step"] = (time.perf_counter() - te) / args.eager_steps * 1e3
    if cuda:
        # do the side lanes (query pass / second key pass / weight gradients) really run beside the main stream?  (HIP multiplexes
        # streams onto a few hardware queues: rspnet_amd/streams.py measures it)
        from rspnet_amd import streams as _streams
        res["lanes_overlap_main"] = _streams.lanes_overlap(dev)
    if stepper is not None and not graphed:
        why = stepper.fallback_reason or "not captured within the warm-up steps"
        res["issue_policy" if why.startswith("issued eagerly by policy") else "graph_fallback"] = why
    roof_steps = steps
    if cuda:
        # roofline pass: the same step issued eagerly ON ONE STREAM with a HIP-event pair around every convolution launch — the
        # kernels' own durations, as rocprofv3's serialised trace reports them; outside the timed region, continuing the same
        # training state (the step's results do not depend on how it is issued)
        from rspnet_amd.engine import BranchStreams
        roof_steps = max(2, min(10, steps))
        saved = (inner.overlap_query_eager, BranchStreams.EAGER_TASKS)
        inner.overlap_query_eager, BranchStreams.EAGER_TASKS = False, False
        try:
            eager_step()
            fence()
            be.event_log, be.hbm_log = [], []
            for _ in range(roof_steps):
                eager_step()
            fence()
            log, be.event_log = be.event_log, None
            hbm, be.hbm_log = be.hbm_log, None
        finally:
            inner.overlap_query_eager, BranchStreams.EAGER_TASKS = saved
    if marks:
        per = [marks[i].elapsed_time(marks[i + 1]) for i in range(steps)]
        res["steps_ms"] = {"p50": round(_pct(per, 0.5), 3), "min": round(min(per), 3), "max": round(max(per), 3),
                           "p90": round(_pct(per, 0.9), 3), "first": round(per[0], 3),
                           "host_enqueue_p50": round(_pct(host, 0.5), 3), "host_enqueue_max": round(max(host), 3),
                           "host_submit_p50": round(_pct([h - w for h, w in zip(host, waits)], 0.5), 3),
                           "host_backpressure_p50": round(_pct(waits, 0.5), 3),
                           "host_issue_idle_gpu_p50": round(_pct(idle_issue, 0.5), 3) if idle_issue else None,
                           "note": "GPU-side step intervals (HIP events on the launch stream at step boundaries); host_enqueue = host "
                                   "time per step() call = host_submit (Python + launches, INCLUDING waits inside the runtime when the "
                                   "hardware queue is full) + host_backpressure (the stepper lets the host run at most 8 steps ahead: "
                                   "waiting there is GPU time, not submission cost); host_issue_idle_gpu = the same call with an empty "
                                   "queue (device synchronised before each of 5 samples): what the host itself needs per step"}
        if seg_host:
            res["steps_ms"]["segment_host_p50"] = {k: round(_pct(v, 0.5), 3) for k, v in seg_host.items()}
        if seg_gpu:
            res["steps_ms"]["segment_gpu_p50"] = seg_gpu      # lane:graph -> ms on its own stream (graphs of different lanes overlap)
    if comm:
        # per-rank stall of the compute stream behind each collective, ms per step (this rank)
        cm = {}
        for name, items in comm.items():
            tot = sum((a.elapsed_time(b) if not isinstance(a, float) else a) for a, b in items) if cuda else sum(items)
            cm[name] = round(tot / steps, 3)
        res["comm_ms"] = cm
    if log:
        per_kernel, per_kind = {}, {}
        for kind, f, e0, e1, kernel, nbytes, _geom, fx in log:
            ms = e0.elapsed_time(e1)
            for table, key in ((per_kernel, kernel), (per_kind, kind)):
                a = table.setdefault(key, [0.0, 0.0, 0, 0.0, 0.0])
                a[0] += f
                a[1] += ms
                a[2] += 1
                a[3] += nbytes
                a[4] += fx
        flops = sum(v[0] for v in per_kernel.values())
        ms = sum(v[1] for v in per_kernel.values())
        all_tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # dominant kernel of THIS backbone = the template instance with the largest share of the timed step.  Each timed
        # group also holds the small helpers launched with it (split-K reduce, dgrad weight re-pack, wgrad slab reduce).
        dom = max(per_kernel, key=lambda k: per_kernel[k][1])
        dflops, dms, dn, dbytes, dexec = per_kernel[dom]
        achieved = dflops / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        executed = dexec / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        fx_all = sum(v[4] for v in per_kernel.values())
        traffic, busy, traffic_src, stale = load_traffic(arch, B, dom)
        alg_gb = dbytes / max(dn, 1) / 1e9
        whole = flops / roof_steps / (step_ms * 1e-3) / 1e12
        res["roofline"] = {
            "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4),
            # what the matrix pipe is asked to do: the same launches priced with the multiply-adds the kernels EXECUTE (the K / row
            # chunks that are zero padding for a whole tile are skipped: rsp_conv3d_executed_fraction, the kernels' own planning code)
            "executed_achieved": round(executed, 2), "executed_frac": round(executed / PEAK_F32_MFMA_TFLOPS, 4),
            "executed_over_algorithmic": round(dexec / dflops, 4) if dflops > 0 else None,
            "mfma_busy": busy, "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of this kernel in the committed "
                                                 "PMC pass (static, see traffic_source)",
            "traffic": traffic, "traffic_static": True,
            # a static look-up keyed on the kernel name: stale when rspnet_amd/csrc has changed since the PMC passes were collected
            "traffic_stale": None if stale is None else stale["traffic_stale"],
            "traffic_build": stale,
            "traffic_unit": "GB of L2-miss (fabric) traffic per launch: PMC FETCH_SIZE x2 + WRITE_SIZE; includes Infinity-Cache "
                            "hits (MI355X_MICROARCH.md), so an upper bound of the HBM bytes",
            "flop_count": "algorithmic: 2 x MACs of the convolution INCLUDING the taps that fall into the zero padding (SURVEY.md 8d); "
                          "the kernels skip the chunks of taps that are padding for a whole tile (DESIGN.md 5d), so the matrix pipe "
                          "executes fewer — up to a third fewer on the 2-frame layers",
            "algorithmic_gb_per_launch": round(alg_gb, 4),
            "traffic_over_algorithmic": None if traffic is None or alg_gb <= 0 else round(traffic / alg_gb, 2),
            "traffic_source": traffic_src,
            "kernel": dom, "launches": dn, "avg_launch_ms": round(dms / max(dn, 1), 4),
            "share_of_step": round(dms / roof_steps / step_ms, 4),
            "share_of_conv_time": round(dms / ms, 4) if ms > 0 else None,
            "share_note": "share_of_step = this kernel's SERIALISED time (one-stream roofline pass) over the TIMED step, whose passes overlap "
                          "on side streams / lanes: an upper bound of the share; share_of_conv_time = over all convolution launches of the same pass",
            "algorithmic_gflop_per_launch": round(dflops / max(dn, 1) / 1e9, 2),
            "whole_step": {"algorithmic_conv_gflop_per_clip": round(flops / roof_steps / B / 1e9, 2),
                           "achieved": round(whole, 2), "frac": round(whole / PEAK_F32_MFMA_TFLOPS, 4),
                           "executed_frac": round(whole * (fx_all / flops) / PEAK_F32_MFMA_TFLOPS, 4) if flops > 0 else None},
            "all_conv_launches": {"achieved": round(all_tf, 2), "frac": round(all_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                  "executed_frac": round(all_tf * (fx_all / flops) / PEAK_F32_MFMA_TFLOPS, 4) if flops > 0 else None,
                                  "ms_per_step": round(ms / roof_steps, 3), "launches": len(log)},
            "per_kernel": {k: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2), "executed_tflops": round(v[4] / (v[1] * 1e-3) / 1e12, 2),
                               "ms_per_step": round(v[1] / roof_steps, 3),
                               "launches_per_step": round(v[2] / roof_steps, 2),
                               "avg_launch_ms": round(v[1] / v[2], 4)}
                           for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
            "per_kind_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 2) for k, v in per_kind.items()},
            "per_kind_ms_per_step": {k: round(v[1] / roof_steps, 3) for k, v in per_kind.items()}}
    if hbm:
        # the streaming kernels of the step against the HBM roofline: algorithmic bytes (every operand tensor moved once) / HIP-event
        # time of the launch group, from the same one-stream pass as the matrix kernels
        PEAK_HBM_TBS = 8.0
        groups = {}
        for kind, nbytes, e0, e1 in hbm:
            a = groups.setdefault(kind, [0, 0.0, 0])
            a[0] += nbytes
            a[1] += e0.elapsed_time(e1)
            a[2] += 1
        res["hbm_kernels"] = {
            "peak_tb_s": PEAK_HBM_TBS, "bound": "hbm",
            "note": "algorithmic bytes / HIP-event time per launch group in the one-stream roofline pass; small launches are latency-, "
                    "not bandwidth-bound, and tensors a producer just wrote may come from the Infinity Cache",
            "groups": {k: {"tb_s": round(v[0] / (v[1] * 1e-3) / 1e12, 3), "frac": round(v[0] / (v[1] * 1e-3) / 1e12 / PEAK_HBM_TBS, 4),
                           "ms_per_step": round(v[1] / roof_steps, 3), "launches_per_step": round(v[2] / roof_steps, 1),
                           "gb_per_step": round(v[0] / roof_steps / 1e9, 3)}
                       for k, v in sorted(groups.items(), key=lambda kv: -kv[1][1]) if v[1] > 0}}
    gc.unfreeze()
    # let the next workload start from an empty device
    del model, opt, im_q, im_k, inner
    if cuda:
        torch.cuda.synchronize()
        be._ws.clear()
        torch.cuda.empty_cache()
    return res, capture


def run_rank(args):
    import datetime

    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}")
    cpu_selftest = args.selftest_cpu
    if cpu_selftest and args.selftest_hang_rank == rank:
        time.sleep(3600)                                    # launcher-deadline self-test: this rank never joins the group
    # each rank's host threads (this interpreter, RCCL's proxy threads, gloo) stay on the rank's own cores: set BEFORE the first
    # GPU / process-group call, which is where those threads are created
    cpus = None
    if ws > 1 and hasattr(os, "sched_setaffinity") and not os.environ.get("RSP_NO_PIN"):
        try:
            cpus = rank_cpu_set(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", ws)))
            if cpus:
                os.sched_setaffinity(0, cpus)
        except (OSError, ValueError) as e:      # a restricted cgroup / unusual topology: run unpinned rather than not at all
            print(f"bench.py: rank {rank}: CPU pinning skipped ({e})", file=sys.stderr)
            cpus = None
    if cpu_selftest:
        dev = torch.device("cpu")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(ws, 1) // 2))
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if args.force_dp and ws != 1:
        raise SystemExit("bench.py: --force-dp is a one-rank run (--gpus 1); with more ranks the collectives are on anyway")
    if args.force_dp and not cpu_selftest:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", str(_free_port()))
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if ws > 1 or args.force_dp:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # a rank that never arrives / a collective that never completes ends the job with rc != 0 (watchdog abort) instead of
        # holding the node until the driver's own limit
        tmo = datetime.timedelta(seconds=args.collective_timeout)
        if cpu_selftest:
            dist.init_process_group("gloo", rank=rank, world_size=ws, timeout=tmo)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=dev, timeout=tmo)

    from rspnet_amd import ops
    if cpu_selftest:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from cpu_ops import CpuOps                     # TEST backend: exercises launcher + host logic only
        ops.set_backend(CpuOps())

    # who is in the job: world size AND the number of distinct devices behind the ranks (two ranks on one GPU would still "scale")
    rccl_ranks = None
    if ws > 1 or args.force_dp:
        if cpu_selftest:
            ident = f"cpu-process-{os.getpid()}"
        else:
            pr = torch.cuda.get_device_properties(dev)
            ident = str(getattr(pr, "uuid", None) or (getattr(pr, "pci_domain_id", 0), getattr(pr, "pci_bus_id", local_rank),
                                                      getattr(pr, "pci_device_id", 0)))
        idents = [None] * ws
        dist.all_gather_object(idents, (ident, len(cpus) if cpus else len(os.sched_getaffinity(0))))
        rccl_ranks = {"world_size": ws, "distinct_devices": len({i for i, _ in idents}), "backend": dist.get_backend(),
                      "host_cpus_per_rank": [n for _, n in idents], "pinned": bool(cpus)}

    B, hw, base_lr = ARCHS[args.arch]
    B = args.batch or B
    hw = args.hw or hw
    want_cpu = ws == 1 and not args.no_cpu_baseline and (not cpu_selftest or args.selftest_parity)
    m, capture = measure(args, args.arch, B, hw, base_lr, args.steps, args.warmup, dev, rank, ws,
                         want_parity=want_cpu and args.cpu_sample == B)
    if rank == 0:
        K, lr = m["K"], m["lr"]
        res = {
            "metric": f"clips/sec pretext step ({args.arch} 16x{hw}x{hw}, B={B}/GPU)",
            "value": round(m["clips_per_s"], 3), "unit": "clips/s", "n_gpus": ws, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(m["ms_per_step"], 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "selftest-cpu" if cpu_selftest else "synthetic",
            "config": {"workload": f"{args.arch} pretext step, {B} synthetic clips/GPU, model input {B}x3x32x{hw}x{hw} "
                                   f"(encoder 3x16x{hw}x{hw}), K={K}, dim=128, T=0.07, m=0.999, SGD lr={lr:g}",
                       "global_batch": B * ws, "parallelism": f"dp{ws}"},
            "final_loss": round(m["final_loss"], 5),
        }
        res["config"]["step_issue"] = {
            "graph": "one replayed HIP graph (rspnet_amd/graph_step.py: the host cannot issue this step fast enough)",
            "graph_segments": "four replayed HIP-graph segments between the step's collective points, RCCL calls issued eagerly in "
                              "between (rspnet_amd/graph_step.py: the host cannot issue this step fast enough)",
            "graph_lanes": "replayed LINEAR HIP graphs — the three forward passes side by side on three streams, the backward in pieces beside a weight-gradient lane (steps_ms.segment_gpu_p50 lists them) — with the step's "
                           "collective points between graphs (rspnet_amd/graph_step.py: the host cannot issue this step fast enough)",
            "eager": "eager launches (independent passes on side streams)"}[m["issue_mode"]] + \
            "; roofline numbers from a one-stream eager pass of the same step outside the timed region"
        res["step_issue_mode"] = m["issue_mode"]
        if rccl_ranks is not None:
            res["rccl_ranks"] = rccl_ranks
        for k in ("host_submit_p50_max_over_ranks", "host_issue_idle_gpu_p50_max_over_ranks"):
            if k in m:
                res[k] = m[k]
        if m["collectives"]:
            res["config"]["collectives"] = ("RCCL (nccl backend): clip all-to-all x2, fused key all-gather x1, bucketed gradient "
                                            "all-reduce from inside backward, gloo side group for the step's random draws"
                                            + (" — forced in a world of one rank (--force-dp)" if args.force_dp else ""))
        if "eager_ms_per_step" in m:
            res["issued_eagerly"] = {"ms_per_step": round(m["eager_ms_per_step"], 3),
                                     "clips_per_s": round(ws * B / m["eager_ms_per_step"] * 1e3, 2), "steps": args.eager_steps,
                                     "note": "same step, eager launches with the same side streams: how N > 1 ranks issue it"}
        for k in ("steps_ms", "comm_ms", "roofline", "hbm_kernels", "graph_fallback", "issue_policy", "lanes_overlap_main"):
            if k in m:
                res[k] = m[k]
    # BASELINE.json configs 3-5 on the same box (N=1, default run only): the other three backbones at their own batch / clip
    # size, a shorter run each — whole-step and dominant-kernel roofline fractions next to the headline.  Each runs in a CHILD
    # interpreter of this file (fresh process, started and waited for — never exec'ed into): whatever happens there — an
    # out-of-memory kill, a runtime crash inside a graph capture — cannot take the headline line down with it.
    if ws == 1 and not cpu_selftest and args.other_workloads and args.arch == "c3d" and not args.batch and not args.hw:
        def child(arch, extra, steps, warmup):
            cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--arch", arch, "--steps", str(steps), "--warmup", str(warmup),
                   "--queue", str(args.queue), "--graph", args.graph, "--no-cpu-baseline", "--no-other-workloads"] + extra
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
            lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
            if r.returncode != 0 or not lines:
                raise RuntimeError(f"rc {r.returncode}: {r.stderr.strip()[-300:]}")
            return json.loads(lines[-1])

        def dp_child(arch, plain_value):
            """The data-parallel path itself on this one GPU: a child run with the real RCCL group of one rank and every collective
            of the N > 1 step forced on (what the 2/4/8-GPU runs execute, minus the wires), in the issue mode N > 1 ranks use."""
            try:
                od = child(arch, ["--force-dp", "--eager-steps", "0"], args.other_steps, args.other_warmup)
                return {"clips_per_s": round(od["value"], 2), "ms_per_step": od["ms_per_step"], "steps": od["steps"],
                        "comm_ms": od.get("comm_ms"), "collectives": od["config"].get("collectives"),
                        "step_issue_mode": od.get("step_issue_mode"), "step_issue": od["config"]["step_issue"],
                        "host_submit_p50": (od.get("steps_ms") or {}).get("host_submit_p50"),
                        "host_issue_idle_gpu_p50": (od.get("steps_ms") or {}).get("host_issue_idle_gpu_p50"),
                        "segment_host_p50": (od.get("steps_ms") or {}).get("segment_host_p50"), "rccl_ranks": od.get("rccl_ranks"),
                        "final_loss": od["final_loss"], "vs_this_line": round(od["value"] / plain_value, 4)}
            except Exception as e:      # noqa: BLE001
                return {"error": f"{type(e).__name__}: {e}"[:400]}

        others = {}
        for arch in ("resnet18", "r2plus1d-vcop", "s3dg"):
            try:
                od = child(arch, [], args.other_steps, args.other_warmup)
                rf = od["roofline"]
                others[arch] = {"workload": od["config"]["workload"], "clips_per_s": round(od["value"], 2),
                                "ms_per_step": od["ms_per_step"], "steps": od["steps"], "warmup": od["warmup"],
                                "steps_ms": od.get("steps_ms"), "whole_step_frac": rf["whole_step"]["frac"],
                                "algorithmic_conv_gflop_per_clip": rf["whole_step"]["algorithmic_conv_gflop_per_clip"],
                                "conv_launches_frac": rf["all_conv_launches"]["frac"],
                                "conv_ms_per_step": rf["all_conv_launches"]["ms_per_step"], "dominant_kernel": rf["kernel"],
                                "dominant_kernel_frac": rf["frac"], "dominant_kernel_share_of_step": rf["share_of_step"],
                                "final_loss": od["final_loss"], "step_issue_mode": od.get("step_issue_mode"),
                                "step_issue": od["config"]["step_issue"]
                                + (f" (graph capture fell back: {od['graph_fallback']})" if "graph_fallback" in od else "")
                                + (f" ({od['issue_policy']})" if "issue_policy" in od else "")}
                # ... and the same backbone the way N > 1 ranks run it (the N = 1 point of ITS scaling curve, same issue mode)
                others[arch]["dp_path_at_one_rank"] = dp_child(arch, od["value"])
            except Exception as e:      # noqa: BLE001 - reported in the line, never fatal for the headline
                others[arch] = {"error": f"{type(e).__name__}: {e}"[:400]}
        res["other_workloads"] = others
        # BASELINE configs 3-5 as FLAT scalars as well: top-level keys, and once more inside the roofline object (a reader that keeps
        # only the contract's keys and objects still holds the three numbers per workload)
        flat = {}
        for arch, key in (("resnet18", "resnet18"), ("r2plus1d-vcop", "r2plus1d"), ("s3dg", "s3dg")):
            o = others.get(arch) or {}
            if "error" in o or "clips_per_s" not in o:
                continue
            flat[f"{key}_clips_per_s"] = o["clips_per_s"]
            flat[f"{key}_ms_per_step"] = o["ms_per_step"]
            flat[f"{key}_whole_step_frac"] = o["whole_step_frac"]
            flat[f"{key}_dominant_kernel_frac"] = o["dominant_kernel_frac"]
        res.update(flat)
        if "roofline" in res:
            res["roofline"]["other_workloads"] = dict(flat)
        res["dp_path_at_one_rank"] = dp_child("c3d", res["value"])
        # the N = 1 number in the issue mode the N > 1 points of a scaling curve use: efficiency is never computed across modes
        dp1 = res["dp_path_at_one_rank"]
        if "error" not in dp1:
            res["n1_same_mode"] = {"clips_per_s": dp1["clips_per_s"], "step_issue_mode": dp1["step_issue_mode"],
                                   "note": "this workload at one rank with every data-parallel collective on (RCCL group of one rank), "
                                           "issued the way ranks of an N > 1 job issue it: the N = 1 point to divide N > 1 values by"}
    if rank == 0:
        if want_cpu:
            try:
                cb, par = cpu_baseline(args.arch, hw, args.cpu_sample, args.cpu_steps, m["K"], m["lr"], parity=capture,
                                       grad_floor=args.grad_floor)
                res["cpu_baseline"] = cb
                res["vs_cpu_baseline"] = round(res["value"] / cb["value"], 1) if cb["value"] > 0 else None
                if par is not None:
                    res["parity"] = par
            except Exception as e:      # noqa: BLE001 - the measured line still goes out
                res["cpu_baseline"] = {"error": f"{type(e).__name__}: {e}"[:400]}
        print(json.dumps(res), flush=True)
    if ws > 1 or args.force_dp:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args.gpus, argv, args.deadline))
    run_rank(args)


if __name__ == "__main__":
    main()
