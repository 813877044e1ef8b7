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
    res, post, mom_post, grads = run_model_step(arch, meta, inputs, 0, dev, "fused")
    errs = compare_to_golden(z, 0, res, post, mom_post, tol=tol, tol_grad=grad_tol(arch))
    wkey, worst = worst_grad_err(z, 0, grads)
    assert worst <= grad_tol(arch), (wkey, worst)
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
