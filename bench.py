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
                 physical cores on BASELINE config 1 (32 clips; 1 warm-up + 2 timed steps) — rank 0, N=1 only.
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


def launch(n: int, argv) -> int:
    """Start n rank processes of this file, wait, return the job's exit code (non-zero if any rank failed)."""
    port = _free_port()
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


def cpu_baseline(arch, hw, sample_b, steps, K):
    import torch
    from oracle import portable as P
    from oracle import restatement as S
    with open(os.path.join(ROOT, "tests", "golden", f"state_spec_{arch.replace('-', '_')}.json")) as f:
        spec = {k: (tuple(s), d) for k, (s, d) in json.load(f).items()}
    spec["queue"] = ((128, K), "float32")
    model, cores = host_cpu()
    torch.set_num_threads(cores)
    state = {k: torch.from_numpy(v) for k, v in P.fill_state(spec, 1).items()}
    moms = [{}]
    g = torch.Generator().manual_seed(0)
    im_q = torch.randn(sample_b, 3, 32, hw, hw, generator=g)
    im_k = torch.randn(sample_b, 3, 32, hw, hw, generator=g)
    times = []
    for it in range(1 + steps):                      # step 0 = warm-up (allocator, oneDNN primitive caches)
        perm = torch.randperm(sample_b, generator=g)
        sh = (torch.randperm(sample_b, generator=g), torch.randperm(sample_b, generator=g))
        t0 = time.perf_counter()
        S.moco_step(arch, [state], [im_q], [im_k], [perm], sh, 2, K=K, lr=0.05, momentum_buffers=moms)
        times.append(time.perf_counter() - t0)
    timed = times[1:] or times
    dt = sum(timed) / len(timed)
    return {"value": round(sample_b / dt, 4), "unit": "clips/s", "cores": cores, "kind": "port", "cpu_model": model,
            "s_per_step": round(dt, 2),
            "sample": f"{arch} full pretext step (2 key passes + query fwd/bwd + losses + SGD) on {sample_b} synthetic clips "
                      f"3x32x{hw}x{hw}, K={K}; 1 warm-up step ({times[0]:.1f} s) + mean of {len(timed)} timed; torch "
                      f"{torch.__version__} CPU ops, {cores} threads = physical cores of {model}"}


# ----------------------------------------------------------------------------------------------------------------------
# one rank
# ----------------------------------------------------------------------------------------------------------------------
def load_traffic(arch, B, kernel):
    """Measured HBM bytes per launch of `kernel` (PMC passes, profiles/traffic.json), GB, or None."""
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if not os.path.exists(tpath):
        return None, None
    with open(tpath) as f:
        t = json.load(f)
    ent = t.get(f"{arch}_b{B}", {}).get(kernel)
    if ent is None:
        return None, None
    return round(ent["hbm_bytes_per_launch"] / 1e9, 4), ent.get("source")


def run_rank(args):
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    if ws != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={ws}")
    cpu_selftest = args.selftest_cpu
    if cpu_selftest:
        dev = torch.device("cpu")
        torch.set_num_threads(max(1, (os.cpu_count() or 2) // max(ws, 1) // 2))
    else:
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if cpu_selftest:
            dist.init_process_group("gloo", rank=rank, world_size=ws)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=dev)

    from rspnet_amd import ops
    if cpu_selftest:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        from cpu_ops import CpuOps                     # TEST backend: exercises launcher + host logic only
        ops.set_backend(CpuOps())
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD

    B, hw, base_lr = ARCHS[args.arch]
    B = args.batch or B
    hw = args.hw or hw
    K = args.queue // (B * ws) * (B * ws)                  # utils/moco.py:8-10 trim
    cfg = {"model": {"arch": args.arch},
           "moco": {"dim": 128, "k": K, "m": 0.999, "t": 0.07, "fc_type": "linear", "diff_speed": [2]}}
    torch.manual_seed(1234)
    model = ModelFactory(cfg).build_moco_diffloss(device=dev)
    model.train()
    crit = Loss(margin=2.0, A=1.0, M=1.0)
    lr = base_lr * ws * B / 64                            # framework/utils/environment.py:13-16
    opt = SGD(model.parameters(), lr=lr, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)  # pretrain.py:65-72
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    im_q = torch.randn(B, 3, 32, hw, hw, device=dev, generator=g)
    im_k = torch.randn(B, 3, 32, hw, hw, device=dev, generator=g)

    def step():
        out, tgt, rl, rt = model(im_q, im_k)
        loss, loss_A, loss_M = crit(out, tgt, rl, rt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        return loss

    def fence():
        if ws > 1:
            dist.barrier()
        if dev.type == "cuda":
            torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    be = ops.backend()
    fence()
    if dev.type == "cuda":
        be.event_log = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    log = []
    if dev.type == "cuda":
        log, be.event_log = be.event_log, None
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss.detach())

    if rank == 0:
        per_kernel, per_kind = {}, {}
        for kind, f, e0, e1, kernel in log:
            ms = e0.elapsed_time(e1)
            for table, key in ((per_kernel, kernel), (per_kind, kind)):
                a = table.setdefault(key, [0.0, 0.0, 0])
                a[0] += f
                a[1] += ms
                a[2] += 1
        flops = sum(v[0] for v in per_kernel.values())
        ms = sum(v[1] for v in per_kernel.values())
        all_tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # dominant kernel of THIS backbone = the template instance with the largest share of the timed step.  Each timed
        # group also holds the small helpers launched with it (split-K reduce, dgrad weight re-pack, wgrad slab reduce).
        dom = max(per_kernel, key=lambda k: per_kernel[k][1]) if per_kernel else None
        dflops, dms, dn = per_kernel[dom] if dom else (0.0, 0.0, 0)
        achieved = dflops / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        traffic, traffic_src = load_traffic(args.arch, B, dom) if dom else (None, None)
        clips = ws * B * args.steps / dt
        res = {
            "metric": f"clips/sec pretext step ({args.arch} 16x{hw}x{hw}, B={B}/GPU)",
            "value": round(clips, 3), "unit": "clips/s", "n_gpus": ws, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "selftest-cpu" if cpu_selftest else "synthetic",
            "config": {"workload": f"{args.arch} pretext step, {B} synthetic clips/GPU, model input {B}x3x32x{hw}x{hw} "
                                   f"(encoder 3x16x{hw}x{hw}), K={K}, dim=128, T=0.07, m=0.999, SGD lr={lr:g}",
                       "global_batch": B * ws, "parallelism": f"dp{ws}"},
            "final_loss": round(final_loss, 5),
        }
        if log:
            step_ms = dt / args.steps * 1e3
            res["roofline"] = {
                "bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS, "unit": "TFLOP/s",
                "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                "traffic_unit": "GB HBM per launch (PMC FETCH_SIZE x2 + WRITE_SIZE)", "traffic_source": traffic_src,
                "kernel": dom, "launches": dn, "avg_launch_ms": round(dms / max(dn, 1), 4),
                "share_of_step": round(dms / args.steps / step_ms, 4),
                "algorithmic_gflop_per_launch": round(dflops / max(dn, 1) / 1e9, 2),
                "all_conv_launches": {"achieved": round(all_tf, 2), "frac": round(all_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                      "ms_per_step": round(ms / args.steps, 3), "launches": len(log)},
                "per_kernel": {k: {"tflops": round(v[0] / (v[1] * 1e-3) / 1e12, 2), "ms_per_step": round(v[1] / args.steps, 3),
                                   "launches_per_step": round(v[2] / args.steps, 2),
                                   "avg_launch_ms": round(v[1] / v[2], 4)}
                               for k, v in sorted(per_kernel.items(), key=lambda kv: -kv[1][1])},
                "per_kind_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 2) for k, v in per_kind.items()},
                "per_kind_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in per_kind.items()}}
        if ws == 1 and not args.no_cpu_baseline and not cpu_selftest:
            res["cpu_baseline"] = cpu_baseline(args.arch, hw, args.cpu_sample, args.cpu_steps, 16384)
        print(json.dumps(res), flush=True)
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch(args.gpus, argv))
    run_rank(args)


if __name__ == "__main__":
    main()
