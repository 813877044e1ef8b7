#!/usr/bin/env python
"""Where does the GPU's whole-step gradient differ from the exact one — and is it further away than the CPU reference's?

bench.py's parity block (C3D, B = 32, the headline workload) reports the HIP path's whole gradient 5.3e-3 (relative L2) from the fp64
gradient of the same state, where the oracle's own fp32 gradient is 2.4e-3 away (profiles/grad_floor.json): 2.2 "floors".  Forward
quantities agree to 1e-6, so the distance is made of ReLU-mask and max-pool arg-max DECISIONS that flip under fp32 rounding.  This
script counts them, layer by layer, on the replayed state:

  * the first step of the seeded bench run on the GPU, with a hook on the BatchNorm backward that sees every (y, mean/invstd,
    scale/shift) of the query encoder: ReLU mask, pool arg-max, batch mean / variance per ConvBN unit;
  * the same step on the oracle restatement in fp32 and in fp64 (oracle/restatement.py with a recording BatchNorm);
  * per unit: elements, mask flips and arg-max flips of GPU-vs-fp64 and oracle-fp32-vs-fp64, relative error of the batch mean and
    variance against fp64; per parameter tensor: relative L2 distance of the gradient from the fp64 gradient, both sides.

Run on the GPU box (needs cuda:0 and ~25 GB of host memory, ~4 min):  python tools/grad_census.py > profiles/r05/grad_census_c3d.json"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import torch.nn.functional as F

import bench

POOLS = {"1": ((1, 2, 2), (1, 2, 2)), "2": ((2, 2, 2), (2, 2, 2)), "3b": ((2, 2, 2), (2, 2, 2)), "4b": ((2, 2, 2), (2, 2, 2))}
ORDER = ["1", "2", "3a", "3b", "4a", "4b", "5a", "5b"]            # models/c3d.py:111-150


def decisions(z, pool):
    """z: BatchNorm output (N, C, D, H, W), any float dtype -> (ReLU mask as uint8, pool arg-max as int32 | None, pooled max > 0)."""
    mask = (z > 0)
    arg = live = None
    if pool is not None:
        mx, idx = F.max_pool3d(torch.relu(z), pool[0], pool[1], return_indices=True)
        arg, live = idx.to(torch.int32), (mx > 0)
    return mask.to(torch.uint8), arg, live


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--hw", type=int, default=112)
    ap.add_argument("--queue", type=int, default=16384)
    a = ap.parse_args()
    from rspnet_amd import ops
    from oracle import restatement as S
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    be = ops.backend()

    # ---- the GPU's first step, BatchNorm backward hooked ---------------------------------------------------------------------
    gpu = []
    real_bwd = be.bn_act_pool_bwd

    def hook(pg, y, residual, dout, gamma, mean_invstd, scale_shift, relu, *rest, **kw):
        if len(gpu) < len(ORDER):
            name = ORDER[len(ORDER) - 1 - len(gpu)]                 # backward order: 5b first
            z = torch.addcmul(scale_shift[1], y, scale_shift[0]).permute(0, 4, 1, 2, 3)      # NDHWC -> NCDHW view
            mask, arg, live = decisions(z.contiguous(), POOLS.get(name))
            var = 1.0 / (mean_invstd[1].double() ** 2) - 1e-5
            gpu.append((name, mask.cpu(), None if arg is None else arg.cpu(), None if live is None else live.cpu(),
                        mean_invstd[0].double().cpu(), var.cpu()))
            del z
        return real_bwd(pg, y, residual, dout, gamma, mean_invstd, scale_shift, relu, *rest, **kw)

    be.bn_act_pool_bwd = hook
    args = bench.parse_args(["--arch", "c3d", "--steps", "1", "--warmup", "0", "--graph", "off", "--batch", str(a.batch), "--hw", str(a.hw),
                             "--queue", str(a.queue), "--no-other-workloads"])
    m, cap = bench.measure(args, "c3d", a.batch, a.hw, 0.1, 1, 0, dev, 0, 1, want_parity=True)
    be.bn_act_pool_bwd = real_bwd
    gpu = {g[0]: g[1:] for g in gpu}
    assert set(gpu) == set(ORDER), sorted(gpu)
    torch.cuda.empty_cache()

    # ---- the same step on the oracle, fp32 and fp64, BatchNorm recording ------------------------------------------------------
    cores = bench.host_cpu()[1]
    torch.set_num_threads(cores)
    real_bn = S._bn

    def run_oracle(dtype):
        rec = {}

        def bn(sd, key, x, eps=1e-5, momentum=0.1):
            if not key.startswith("encoder_q.") or not S._BN_TRAINING[0]:
                return real_bn(sd, key, x, eps, momentum)
            sd[key + ".num_batches_tracked"] += 1
            out, mean, invstd = torch.native_batch_norm(x, sd[key + ".weight"], sd[key + ".bias"], sd[key + ".running_mean"],
                                                        sd[key + ".running_var"], True, momentum, eps)
            name = key.split(".bn")[-1]
            with torch.no_grad():
                mask, arg, live = decisions(out.detach(), POOLS.get(name))
                rec[name] = (mask, arg, live, mean.detach().double(), (1.0 / invstd.detach().double() ** 2) - eps)
            return out

        S._bn = bn
        t0 = time.perf_counter()
        prev = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        try:
            cast = (lambda v: v.clone().to(dtype) if v.dtype in (torch.float32, torch.float64) else v.clone())
            st = {k: cast(v) for k, v in cap["state"].items()}
            o = S.moco_step("c3d", [st], [cap["im_q"].to(dtype)], [cap["im_k"].to(dtype)], [cap["perm"]], (cap["sh1"], cap["sh2"]),
                            cap["speed"], K=cap["K"], lr=cap["lr"], momentum_buffers=[{}])[0]
        finally:
            torch.set_default_dtype(prev)
            S._bn = real_bn
        grads = {k: v.detach().double() for k, v in o["grads"].items() if v is not None}
        return rec, grads, time.perf_counter() - t0

    rec32, g32, t32 = run_oracle(torch.float32)
    rec64, g64, t64 = run_oracle(torch.float64)

    # ---- the census -----------------------------------------------------------------------------------------------------------
    def rel(x, ref):
        return float(((x - ref).abs() / ref.abs().clamp_min(1e-30)).max())

    units = {}
    for name in ORDER:
        mk_g, ar_g, lv_g, mean_g, var_g = gpu[name]
        mk_3, ar_3, lv_3, mean_3, var_3 = rec32[name]
        mk_6, ar_6, lv_6, mean_6, var_6 = rec64[name]
        u = {"elements": int(mk_6.numel()),
             "relu_mask_flips": {"gpu_vs_fp64": int((mk_g != mk_6).sum()), "oracle_fp32_vs_fp64": int((mk_3 != mk_6).sum()),
                                 "gpu_vs_oracle_fp32": int((mk_g != mk_3).sum())},
             "batch_mean_max_rel_err": {"gpu": rel(mean_g, mean_6), "oracle_fp32": rel(mean_3, mean_6)},
             "batch_var_max_rel_err": {"gpu": rel(var_g, var_6), "oracle_fp32": rel(var_3, var_6)}}
        if ar_6 is not None:
            u["pool_windows"] = int(ar_6.numel())
            u["pool_argmax_flips"] = {"gpu_vs_fp64": int(((ar_g != ar_6) & lv_6).sum()), "oracle_fp32_vs_fp64": int(((ar_3 != ar_6) & lv_6).sum())}
        units["conv" + name] = u

    def l2(ga, gb, keys):
        num = sum(float(((ga[k].double() - gb[k]) ** 2).sum()) for k in keys)
        den = sum(float((gb[k] ** 2).sum()) for k in keys)
        return (num / den) ** 0.5 if den > 0 else None

    ggpu = {k: v.double() for k, v in cap["grads"].items()}
    keys = [k for k in g64 if k in ggpu and k in g32]
    per_tensor = {k: {"gpu_vs_fp64": l2(ggpu, g64, [k]), "oracle_fp32_vs_fp64": l2(g32, g64, [k])}
                  for k in keys if float((g64[k] ** 2).sum()) > 0}
    out = {"workload": f"c3d pretext step, B={a.batch}, {a.hw}x{a.hw}, K={cap['K']}: the seeded first step of bench.py",
           "whole_gradient_rel_l2": {"gpu_vs_fp64": l2(ggpu, g64, keys), "oracle_fp32_vs_fp64": l2(g32, g64, keys), "gpu_vs_oracle_fp32": l2(ggpu, g32, keys)},
           "units": units, "gradient_by_tensor": per_tensor,
           "oracle_seconds": {"fp32": round(t32, 1), "fp64": round(t64, 1)}, "host_threads": cores}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
