#!/usr/bin/env python
"""bench.py — clips/sec of the RSPNet pretext step (BASELINE.json metric) on N MI355X of one node.

A step = momentum update + diff-speed gather + 2 key-encoder passes (shuffle-BN) + query forward/backward +
InfoNCE/ranking losses + enqueue + gradient all-reduce + SGD, on synthetic clips already resident in HBM.
Workload at every N: BASELINE configs[1] — C3D, B=32 clips per GPU, model input (32,3,32,112,112) -> encoder input
3x16x112x112, K=16384, dim=128, T=0.07, m=0.999, fp32 (weak scaling).  One process per GPU (torch.distributed/RCCL).

Prints ONE JSON line on rank 0 (see the driver contract): value = whole-job clips/s; plus
  roofline     — conv MFMA launches (fwd + dgrad + wgrad of every layer): algorithmic FLOPs / HIP-event time on the
                 launch stream, against the fp32-input MFMA peak (157.3 TFLOP/s, MI355X_MICROARCH.md);
  cpu_baseline — the oracle restatement (oracle/restatement.py, proven equal to the reference) timed on this host's
                 cores on a bounded sample of the same workload (rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch
import torch.distributed as dist

PEAK_F32_MFMA_TFLOPS = 157.3
ARCHS = {
    # arch: (per-GPU batch, H=W, base lr from config/pretrain/*.jsonnet)
    "c3d": (32, 112, 0.1),
    "resnet18": (32, 112, 0.1),
    "r2plus1d-vcop": (32, 112, 0.05),
    "s3dg": (16, 224, 0.05),
}


def cpu_baseline(arch, hw, sample_b):
    """Oracle restatement timed on the host (checker code, never the product path)."""
    import json as _json
    from oracle import portable as P
    from oracle import restatement as S
    with open(os.path.join(ROOT, "tests", "golden", f"state_spec_{arch.replace('-', '_')}.json")) as f:
        spec = {k: (tuple(s), d) for k, (s, d) in _json.load(f).items()}
    K = 16384
    spec["queue"] = ((128, K), "float32")
    state = {k: torch.from_numpy(v) for k, v in P.fill_state(spec, 1).items()}
    g = torch.Generator().manual_seed(0)
    im_q = torch.randn(sample_b, 3, 32, hw, hw, generator=g)
    im_k = torch.randn(sample_b, 3, 32, hw, hw, generator=g)
    perm = torch.randperm(sample_b, generator=g)
    sh = (torch.randperm(sample_b, generator=g), torch.randperm(sample_b, generator=g))
    t0 = time.perf_counter()
    S.moco_step(arch, [state], [im_q], [im_k], [perm], sh, 2, K=K, lr=0.05, momentum_buffers=[{}])
    dt = time.perf_counter() - t0
    return {"value": round(sample_b / dt, 4), "unit": "clips/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"1 full pretext step (fwd+bwd+SGD) on {sample_b} synthetic clips 3x32x{hw}x{hw}, K={K}, "
                      f"torch {torch.__version__} CPU ops, {dt:.1f} s"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--arch", default="c3d", choices=sorted(ARCHS))
    ap.add_argument("--batch", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=8)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    ws = int(os.environ.get("WORLD_SIZE", "1"))
    assert ws == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={ws}: launch with torch.distributed.run"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if ws > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=dev)

    from rspnet_amd import ops
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD

    B, hw, base_lr = ARCHS[args.arch]
    B = args.batch or B
    K = 16384 // (B * ws) * (B * ws)                      # utils/moco.py:8-10 trim
    cfg = {"model": {"arch": args.arch},
           "moco": {"dim": 128, "k": K, "m": 0.999, "t": 0.07, "fc_type": "linear", "diff_speed": [2]}}
    torch.manual_seed(1234)
    model = ModelFactory(cfg).build_moco_diffloss(device=dev)
    model.train()
    crit = Loss(margin=2.0, A=1.0, M=1.0)
    lr = base_lr * ws * B / 64                            # framework/utils/environment.py:13-16
    opt = SGD([p for p in model.parameters() if p.requires_grad], lr=lr, momentum=0.9, dampening=0.0,
              weight_decay=1e-4, nesterov=False)
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
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    be = ops.backend()
    fence()
    be.event_log = []
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    fence()
    dt = time.perf_counter() - t0
    log, be.event_log = be.event_log, None
    t = torch.tensor([dt], dtype=torch.float64, device=dev)
    if ws > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dt = float(t.item())
    final_loss = float(loss.detach())

    if rank == 0:
        flops = sum(e[1] for e in log)
        ms = sum(e[2].elapsed_time(e[3]) for e in log)
        per_kind = {}
        for kind, f, e0, e1, ncols in log:
            a = per_kind.setdefault(kind, [0.0, 0.0, 0])
            a[0] += f
            a[1] += e0.elapsed_time(e1)
            a[2] += 1
        all_tf = flops / (ms * 1e-3) / 1e12 if ms > 0 else 0.0
        # dominant kernel = igemm_kernel<128,128> (forward and input-gradient launches with GEMM N > 64 share it; each timed
        # group also holds that launch's split-K reduce / dgrad weight re-pack, ~3% of it).  The narrow-tile and stem launches
        # and wgrad_dma_kernel are reported beside it.
        dom = [e for e in log if e[0] in ("conv_fwd", "conv_dgrad") and e[4] > 64]
        dflops, dms, dn = sum(e[1] for e in dom), sum(e[2].elapsed_time(e[3]) for e in dom), len(dom)
        achieved = dflops / (dms * 1e-3) / 1e12 if dms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and args.arch == "c3d" and B == 32:
            with open(tpath) as f:
                traffic = round(json.load(f)["hbm_bytes_per_launch"] / 1e9, 3)
        clips = ws * B * args.steps / dt
        res = {
            "metric": f"clips/sec pretext step ({args.arch} 16x{hw}x{hw}, B={B}/GPU)",
            "value": round(clips, 3), "unit": "clips/s", "n_gpus": ws, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{args.arch} pretext step, {B} synthetic clips/GPU, model input {B}x3x32x{hw}x{hw} "
                                   f"(encoder 3x16x{hw}x{hw}), K={K}, dim=128, T=0.07, m=0.999, SGD lr={lr:g}",
                       "global_batch": B * ws, "parallelism": f"dp{ws}"},
            "final_loss": round(final_loss, 5),
            "roofline": {"bound": "mfma", "achieved": round(achieved, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                         "unit": "TFLOP/s", "frac": round(achieved / PEAK_F32_MFMA_TFLOPS, 4), "traffic": traffic,
                         "traffic_unit": "GB HBM per launch (PMC FETCH_SIZE x2 + WRITE_SIZE, profiles/traffic.json)",
                         "kernel": "igemm_kernel<128,128> (implicit-GEMM conv3d forward + dgrad, v_mfma_f32_32x32x2_f32)",
                         "launches": dn, "avg_launch_ms": round(dms / max(dn, 1), 4),
                         "algorithmic_gflop_per_launch": round(dflops / max(dn, 1) / 1e9, 2),
                         "all_conv_launches": {"achieved": round(all_tf, 2), "frac": round(all_tf / PEAK_F32_MFMA_TFLOPS, 4),
                                               "ms_per_step": round(ms / args.steps, 3), "launches": len(log)},
                         "per_kind_tflops": {k: round(v[0] / (v[1] * 1e-3) / 1e12, 2) for k, v in per_kind.items()},
                         "per_kind_ms_per_step": {k: round(v[1] / args.steps, 3) for k, v in per_kind.items()}},
        }
        if ws == 1 and not args.no_cpu_baseline:
            res["cpu_baseline"] = cpu_baseline(args.arch, hw, args.cpu_sample)
        print(json.dumps(res), flush=True)
    if ws > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
