"""GPU: the 2-rank data path with the HIP kernels, on ONE device.  Two threads act as the two ranks over PyTorch's in-process
"threaded" process group (torch.testing._internal.distributed.multi_threaded_pg: real collective semantics, tensors copied
between the ranks' buffers), both on cuda:0, and each rank's step is compared with the 2-rank fixture generated from the
reference under DDP.  This covers on the GPU everything multi-rank except RCCL itself: device-side clip all-to-all after
rsp_clip_gather, the fused key all-gather + rsp_rows_gather un-shuffle, global-queue ordering, the bucketed gradient all-reduce
launched from inside backward, and the permutation broadcast fallback through the device (no gloo side group here).
The ranks take turns (one lock, released only inside collectives) because they share the process-wide op backend and its
scratch buffer."""
import random
import threading

import numpy as np
import pytest
import torch
import torch.distributed as dist

from golden_util import build_inputs, cases_for, compare_to_golden, grad_tol, load_case, worst_grad_err

pytestmark = pytest.mark.gpu
TOL = 1e-3

def _run_rank(rank, ws, arch, seed, lock, tls, out, dev):
    from torch.testing._internal.distributed.multi_threaded_pg import _install_threaded_pg  # noqa: F401
    from model_util import make_cfg
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD
    if dev.type == "cuda":
        torch.cuda.set_device(dev)
    dist.init_process_group(backend="threaded", rank=rank, world_size=ws, store=out["store"])   # rendezvous: outside the lock
    lock.acquire()
    try:
        z, meta = load_case(arch, ws, seed)
        spec, (state, mom, clips, perms_B, sh) = build_inputs(arch, meta)
        tls.perms = [torch.from_numpy(np.asarray(p, dtype=np.int64)) for p in (perms_B[rank], sh[0], sh[1])]
        tls.speed = meta["speed"]
        wrapped = ModelFactory(make_cfg(meta.get("arch", arch), meta["K"], fc_type=meta.get("fc_type", "linear"), m=meta["m"],
                                        T=meta["T"])).build_moco_diffloss(device=dev)
        model = wrapped.module
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
        model.train()
        params = [p for p in wrapped.parameters() if p.requires_grad]
        opt = SGD(wrapped.parameters(), lr=meta["lr"], momentum=meta["sgd_momentum"], dampening=0.0, weight_decay=meta["weight_decay"],
                  nesterov=False)
        names = {id(p): n for n, p in model.named_parameters()}
        for p in params:
            if names[id(p)] in mom:
                opt.state[p]["momentum_buffer"] = torch.from_numpy(mom[names[id(p)]].copy()).to(dev)
        crit = Loss(margin=meta["margin"], A=meta["A"], M=meta["M"])
        o, tgt, rl, rt = wrapped(torch.from_numpy(clips[rank][0]).to(dev), torch.from_numpy(clips[rank][1]).to(dev))
        loss, loss_A, loss_M = crit(o, tgt, rl, rt)
        opt.zero_grad()
        loss.backward()
        grads = {names[id(p)]: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for p in params}
        opt.step()
        if dev.type == "cuda":
            torch.cuda.synchronize()
        q_A, q_M = model._last_q
        res = {"loss": loss, "loss_A": loss_A, "loss_M": loss_M, "logits1": o[0], "logits2": o[1], "l_pos_M": rl[0],
               "l_neg_M": rl[1], "q_A": q_A, "q_M": q_M}
        dim = q_A.shape[1]
        for tag, (feats, order) in zip(("kneg", "k"), model._last_k):      # this rank's encoder_k outputs, reference order
            shuf = torch.empty_like(feats)
            shuf[torch.from_numpy(order).to(feats.device)] = feats
            res[f"{tag}_A_shuf"], res[f"{tag}_M_shuf"] = shuf[:, :dim], shuf[:, dim:]
        res = {k: v.detach().cpu().numpy() for k, v in res.items()}
        post = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
        mom_post = {names[id(p)]: opt.state[p]["momentum_buffer"].detach().cpu().numpy() for p in params
                    if "momentum_buffer" in opt.state[p]}
        gate = grad_tol(arch, 2)
        from golden_util import grad_stats
        out[("stats", rank)] = grad_stats(z, rank, grads)
        errs = compare_to_golden(z, rank, res, post, mom_post, tol=TOL, tol_grad=gate)
        wkey, worst = worst_grad_err(z, rank, grads)
        assert worst <= gate, (wkey, worst)
        out[rank] = (errs, worst)
    except BaseException as e:      # noqa: BLE001 - reported by the main thread
        out[rank] = e
    finally:
        lock.release()
        try:
            dist.destroy_process_group()
        except Exception:           # noqa: BLE001
            pass


@pytest.mark.parametrize("arch,seed", [(a, s) for arch in ("c3d", "c3d:linear:4", "resnet18", "r2plus1d-vcop", "s3dg") for a, w, s in cases_for(arch, 2)])
def test_two_ranks_on_one_gpu_match_the_ddp_fixture(arch, seed):
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    run_two_ranks(arch, seed, torch.device("cuda", 0))      # the library's default tile plan: no second evaluation


def run_two_ranks(arch, seed, dev):
    from torch.testing._internal.distributed import multi_threaded_pg as tpg
    ws = 2
    lock, tls = threading.Lock(), threading.local()
    saved = {n: getattr(dist, n) for n in ("all_to_all_single", "all_gather_into_tensor", "broadcast", "all_reduce", "barrier",
                                            "new_group")}
    rp, ch = torch.randperm, random.choice

    def unlocked(fn):
        def call(*a, **k):
            lock.release()          # let the other rank reach the same collective
            try:
                return fn(*a, **k)
            finally:
                lock.acquire()
        return call

    def randperm(n, *a, **k):
        # perms = [the _diff_speed permutation (drawn with device=...), shuffle #1, shuffle #2 (host draws, in call order)]: the
        # product draws its host-side decisions first (MoCoDiffLossTwoFc._host_part), so the kinds are told apart by `device`
        if not getattr(tls, "perms", None):
            return rp(n, *a, **k)
        if k.get("device") is not None:
            p = tls.perms[0]
            assert p.numel() == n
            return p.clone().to(k["device"])
        p = tls.perms.pop(1)
        assert p.numel() == n
        return p.clone()

    def no_gloo(*a, **k):
        raise RuntimeError("no side group in the threaded world")

    tpg._install_threaded_pg()
    torch._C._distributed_c10d._set_thread_isolation_mode(True)      # per-thread process-group registry (MultiThreadedTestCase)
    try:
        for n, f in saved.items():
            setattr(dist, n, no_gloo if n == "new_group" else unlocked(f))
        torch.randperm = randperm
        random.choice = lambda seq: getattr(tls, "speed", None) or ch(seq)
        out = {"store": dist.HashStore()}
        threads = [threading.Thread(target=_run_rank, args=(r, ws, arch, seed, lock, tls, out, dev), daemon=True) for r in range(ws)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=240)
        assert all(not t.is_alive() for t in threads), "a rank hung"
    finally:
        for n, f in saved.items():
            setattr(dist, n, f)
        torch.randperm, random.choice = rp, ch
        torch._C._distributed_c10d._set_thread_isolation_mode(False)
        tpg._uninstall_threaded_pg()
    for r in range(ws):
        if ("stats", r) in out:
            st = out[("stats", r)]
            print(f"\n{arch} rank {r} gradient tensors vs golden: worst {st['worst'][1]:.2e} ({st['worst'][0]}), p90 {st['p90']:.2e}, "
                  f"median {st['median']:.2e}, whole gradient {st['whole']:.2e} ({st['n']} tensors)")
        if isinstance(out.get(r), BaseException):
            raise out[r]
        assert r in out, f"rank {r} produced nothing"
        print(f"\n{arch} rank {r}: " + ", ".join(f"{k}={v:.1e}" for k, v in out[r][0].items()) + f", grads={out[r][1]:.1e}")
