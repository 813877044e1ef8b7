"""One rank, every data-parallel collective: worker shared by tests/test_distributed_cpu.py (gloo + checker backend) and
tests/test_rccl_gpu.py (the real `nccl` backend = RCCL on cuda:0 with the HIP kernels).  The process group has ONE member and
`force_collectives` is on, so the step issues the clip all-to-alls with their split lists, the fused key all-gather, the bucketed
gradient all-reduces from inside backward and the host-side broadcast of the random draws over the gloo side group — and must
still reproduce the 1-rank fixture generated from the reference."""
import collections
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def forced_worker(rank, backend, arch, seed, port, out_path, graph=False):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch
    import torch.distributed as dist
    from golden_util import build_inputs, compare_to_golden, fwd_tol, grad_tol, load_case, worst_grad_err
    from model_util import run_model_step
    from rspnet_amd import ops
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["RSP_FORCE_COLLECTIVES"] = "1"
    if backend == "nccl":
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
        assert ops.backend().name == "hip"
        tol = 1e-3
    else:
        from cpu_ops import CpuOps
        torch.set_num_threads(4)
        dev = torch.device("cpu")
        dist.init_process_group("gloo", rank=0, world_size=1)
        ops.set_backend(CpuOps())
        tol = fwd_tol(arch, 2e-4)
    calls = collections.Counter()
    groups = []
    for name in ("all_to_all_single", "all_gather_into_tensor", "all_reduce", "broadcast", "new_group"):
        def make(name, fn):
            def spy(*a, **k):
                calls[name] += 1
                if name == "broadcast" and k.get("group") is not None:
                    groups.append(dist.get_backend(k["group"]))
                if name == "all_to_all_single":
                    assert a[2] is not None and a[3] is not None and sum(a[2]) == a[0].shape[0] and sum(a[3]) == a[1].shape[0]
                return fn(*a, **k)
            return spy
        setattr(dist, name, make(name, getattr(dist, name)))
    z, meta = load_case(arch, 1, seed)
    spec, inputs = build_inputs(arch, meta)
    from golden_util import check_step_gradients
    runs = []

    def step():
        before = dict(calls)
        out = run_model_step(arch, meta, inputs, 0, dev, "fused")
        runs.append({k: calls[k] - before.get(k, 0) for k in calls})
        return out

    errs, worst, plan, _ = check_step_gradients(arch, 1, 0, z, step, tol)
    calls = collections.Counter(runs[0])       # (the collectives of ONE step: a second run under the wide-tile plan repeats them)
    # what ran: 2 clip all-to-alls, 1 fused key all-gather, >= 1 bucket all-reduce (+1: the side-group agreement under nccl),
    # the constructor's broadcasts + 1 host-side broadcast of (speed, permutations) per step
    assert calls["all_to_all_single"] == 2 and calls["all_gather_into_tensor"] == 1, dict(calls)
    assert calls["all_reduce"] >= (2 if backend == "nccl" else 1), dict(calls)
    assert calls["broadcast"] >= 3, dict(calls)
    if backend == "nccl":
        assert calls["new_group"] == 1 and groups and groups[-1] == "gloo", (dict(calls), groups)
    with open(out_path, "w") as f:
        json.dump({"calls": dict(calls), "errs": {k: float(v) for k, v in errs.items()}, "worst_grad": float(worst),
                   "side_group": groups[-1] if groups else None}, f)
    dist.barrier()
    dist.destroy_process_group()


def segmented_worker(rank, arch, B, HW, port, out_path, mode="lanes", steps=7):
    """One RCCL rank with every collective forced on: `steps` consecutive steps of two identically initialised models — one
    driven by the eager data-parallel loop body (bucketed all-reduce from inside backward), one by GraphedPretextStep, which at
    collectives-on replays the step as HIP-graph SEGMENTS between the collective points — must leave bit-identical losses, logits,
    parameters, queue and pointer (tests/test_graph_step_gpu.py's check for the N > 1 issue mode)."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import random
    import torch
    import torch.distributed as dist
    from golden_util import load_spec
    from model_util import make_cfg
    from oracle import portable as P
    from rspnet_amd import ops
    from rspnet_amd.graph_step import GraphedPretextStep
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    os.environ["RSP_FORCE_COLLECTIVES"] = "1"
    os.environ["RSP_GRAPH_MODE"] = mode
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert ops.backend().name == "hip"
    K = 64
    calls = collections.Counter()
    # Ordering of the gradient all-reduce inside a REPLAYED step: with one rank an all-reduce moves nothing, so a bucket
    # reduced before the backward piece that writes it had finished, or an SGD graph that ran before the reduce, would pass every
    # value check.  The spy therefore (a) snapshots each flat-gradient bucket ON THE ISSUING STREAM at the moment of issue — it must
    # equal what the bucket holds when the step is over — and (b) DOUBLES the bucket there, as a sum over two identical ranks
    # would: the optimizer of either loop sees the doubled gradient only if it is ordered behind the "collective", and the two loops
    # stay bit-identical only if both are.
    cur = {"model": None, "snaps": []}
    for name in ("all_to_all_single", "all_gather_into_tensor", "all_reduce", "broadcast"):
        def make(name, fn):
            def spy(*a, **k):
                if name == "broadcast":
                    # DDP's per-forward buffer broadcast: the flat tensor of all BatchNorm running statistics, from rank 0, on the
                    # device group (the host-side message of the step's draws travels on the gloo side group: not counted here)
                    bn = getattr(cur["model"], "_bn_flat", None)
                    if bn is not None and a[0].data_ptr() == bn.data_ptr() and a[0].numel() == bn.numel():
                        calls["broadcast_buffers"] += 1
                        # with one rank the broadcast moves nothing: the spy scales the statistics on the issuing stream, as if rank
                        # 0 had sent other values — (s * 1.25) moved by this step's batch is not (s moved by the batch) * 1.25, so
                        # the final running statistics of the two loops agree only if BOTH order the broadcast in front of every
                        # BatchNorm of the step (the replayed query pass forks before the clip exchange: rspnet_amd/graph_step.py)
                        a[0].mul_(1.25)
                    return fn(*a, **k)
                calls[name] += 1
                flat = getattr(cur["model"], "_flat", None) if name == "all_reduce" else None
                if flat is not None and a[0].untyped_storage().data_ptr() == flat.g_flat.untyped_storage().data_ptr():
                    t = a[0]
                    cur["snaps"].append(((t.data_ptr() - flat.g_flat.data_ptr()) // 4, t.numel(), t.clone()))
                    t.mul_(2.0)
                return fn(*a, **k)
            return spy
        setattr(dist, name, make(name, getattr(dist, name)))
    clips = [tuple(torch.from_numpy(c).to(dev) for c in P.clips(10 + i, 0, (B, 3, 32, HW, HW))) for i in range(steps)]
    results, info = [], {}
    for how in ("eager", "segments"):
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        random.seed(7)
        wrapped = ModelFactory(make_cfg(arch, K)).build_moco_diffloss(device=dev)
        spec = dict(load_spec(arch))
        spec["queue"] = ((128, K), "float32")
        wrapped.module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.fill_state(spec, 3).items()})
        wrapped.train()
        assert wrapped.module._dp()[2], "collectives must be on"
        crit = Loss(margin=2.0, A=1.0, M=1.0)
        opt = SGD(wrapped.parameters(), lr=0.05, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
        stepper = GraphedPretextStep(wrapped, crit, opt, warmup=2, issue="graph") if how == "segments" else None
        trace = []
        before = dict(calls)
        cur["model"] = wrapped.module
        checks = []
        for im_q, im_k in clips:
            cur["snaps"] = []
            if stepper is None:
                out, tgt, rl, rt = wrapped(im_q, im_k)
                loss, la, lm = crit(out, tgt, rl, rt)
                opt.zero_grad()
                loss.backward()
                opt.step()
            else:
                loss, la, lm, out, rl = stepper(im_q, im_k)
            trace.append((loss.detach().clone(), out[0].detach().clone(), rl[0].detach().clone()))
            # (stream-ordered copy behind the step, no host synchronisation: the host keeps running ahead as in a real loop)
            checks.append((cur["snaps"], wrapped.module._flat.g_flat.clone()))
        torch.cuda.synchronize()
        for it, (snaps, final) in enumerate(checks):
            assert len(snaps) >= 1 and sum(n for _, n, _ in snaps) == final.numel(), (how, it, len(snaps))
            for off, n, snap in snaps:
                assert not torch.isnan(snap).any(), (how, it, off)
                assert torch.equal(snap * 2.0, final[off:off + n]), (how, "a gradient bucket changed after its all-reduce was issued", it, off, n)
        info.setdefault("buckets", {})[how] = len(checks[-1][0])
        info[how] = {k: calls[k] - before.get(k, 0) for k in calls}
        if stepper is not None:
            assert stepper.mode == mode and not stepper.disabled, stepper.fallback_reason
            assert len(stepper.graphs) == 1
            seq = next(iter(stepper.graphs.values()))[3]
            info["graphs"] = sum(1 for op in seq if op[0] == "g")
            info["collective_points"] = sum(1 for op in seq if op[0] == "e")
            info["lanes"] = sorted({op[1] for op in seq if op[0] == "g"})
        results.append((trace, {k: v.detach().clone() for k, v in wrapped.module.state_dict().items()}))
    (te, se), (tg, sg) = results
    for i, ((l0, o0, r0), (l1, o1, r1)) in enumerate(zip(te, tg)):
        assert torch.equal(l0, l1) and torch.equal(o0, o1) and torch.equal(r0, r1), (arch, "step", i, float(l0), float(l1))
    assert int(sg["queue_ptr"]) == int(se["queue_ptr"]) == (steps * B) % K
    for k in se:
        assert torch.equal(se[k], sg[k]), (arch, k)
    with open(out_path, "w") as f:
        json.dump(info, f)
    dist.barrier()
    dist.destroy_process_group()


def bucket_order_worker(rank, arch, seed, port, out_path):
    """Ordering of the bucketed gradient all-reduce against the weight-gradient side tasks: in a world of one rank an
    all-reduce is the identity, so a bucket let go BEFORE a side-stream weight gradient has written into it would go unnoticed by
    every value check.  Here `dist.all_reduce` is patched to snapshot the bucket ON THE ISSUING STREAM at the moment of issue; the
    gradient buffer is poisoned with NaN before the backward, every weight gradient is sent to the side stream
    (RSP_WGRAD_ASIDE_GFLOP huge), and each snapshot must hold no NaN and equal the bucket's final content bit for bit."""
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ["RSP_WGRAD_ASIDE_GFLOP"] = "1e9"         # (read when rspnet_amd.engine is imported)
    os.environ["RSP_FORCE_COLLECTIVES"] = "1"
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import torch
    import torch.distributed as dist
    from golden_util import build_inputs, load_case
    from model_util import ReplayRNG, make_cfg
    from rspnet_amd import ops
    from rspnet_amd.engine import BranchStreams
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    assert ops.backend().name == "hip" and BranchStreams.SMALL_WGRAD_FLOPS > 1e17
    z, meta = load_case(arch, 1, seed)
    spec, (state, mom, clips, perms_B, sh) = build_inputs(arch, meta)
    wrapped = ModelFactory(make_cfg(meta.get("arch", arch), meta["K"], fc_type=meta.get("fc_type", "linear"), m=meta["m"],
                                    T=meta["T"])).build_moco_diffloss(device=dev)
    model = wrapped.module
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    model.train()
    opt = SGD([p for p in wrapped.parameters() if p.requires_grad], lr=meta["lr"], momentum=0.9, weight_decay=1e-4)
    crit = Loss(margin=meta["margin"], A=meta["A"], M=meta["M"])
    im_q, im_k = torch.from_numpy(clips[0][0]).to(dev), torch.from_numpy(clips[0][1]).to(dev)
    snaps, side_issues = [], 0
    real = dist.all_reduce
    main = torch.cuda.current_stream(dev).cuda_stream

    def spy(t, *a, **k):
        nonlocal side_issues
        if model._flat is not None and t.untyped_storage().data_ptr() == model._flat.g_flat.untyped_storage().data_ptr():
            off = (t.data_ptr() - model._flat.g_flat.data_ptr()) // 4
            snaps.append((off, t.numel(), t.clone()))          # a copy on the issuing stream: what RCCL would read
            side_issues += int(torch.cuda.current_stream(dev).cuda_stream != main)
        return real(t, *a, **k)

    dist.all_reduce = spy
    for it in range(3):                                        # several steps: the side stream is busy with the previous step's tail
        snaps.clear()
        with ReplayRNG([perms_B[0], sh[0], sh[1]], meta["speed"]):
            out, tgt, rl, rt = wrapped(im_q, im_k)
        loss, _, _ = crit(out, tgt, rl, rt)
        opt.zero_grad()
        model._flat.g_flat.fill_(float("nan"))
        loss.backward()
        torch.cuda.synchronize()
        final = model._flat.g_flat.clone()
        assert not torch.isnan(final).any(), "a trained parameter received no gradient"
        assert len(snaps) >= 2, len(snaps)
        covered = 0
        for off, n, snap in snaps:
            assert not torch.isnan(snap).any(), ("bucket issued before its gradients were written", it, off, n)
            assert torch.equal(snap, final[off:off + n]), ("bucket changed after its all-reduce was issued", it, off, n)
            covered += n
        assert covered == final.numel()
        opt.step()
    with open(out_path, "w") as f:
        json.dump({"buckets": len(snaps), "issued_from_side_stream": side_issues}, f)
    dist.all_reduce = real
    dist.barrier()
    dist.destroy_process_group()
