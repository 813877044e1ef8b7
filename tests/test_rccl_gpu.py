"""GPU.  (1) ONE device: the real `nccl` backend (RCCL) in a world of one rank with `force_collectives` — every collective of the
data-parallel step executes on the 1-GPU box (clip all-to-all with split lists, fused key all-gather, bucketed gradient all-reduce
from inside backward with the weight-gradient side tasks joined in front of it, the gloo side group created next to the RCCL group)
and the step still matches the 1-rank fixture; eagerly and, with RSP_GRAPH_COLLECTIVES=1, as a captured HIP graph.
(2) >= 2 devices (skipped on the 1-GPU test box): the 2-rank DDP fixtures through REAL RCCL — one process per GPU,
`nccl` backend, the product's clip all-to-all (uneven splits), fused key all-gather, bucketed gradient all-reduce launched
from inside backward and the gloo side group next to the RCCL group — compared with the reference-under-DDP goldens exactly as
tests/test_distributed_cpu.py does over gloo with the checker backend and tests/test_two_rank_gpu.py does with the HIP kernels
over the in-process threaded group.  This is the only test that executes the RCCL collectives themselves."""
import os
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from golden_util import cases_for

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, ws, arch, seed, port, tmp):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    import torch.distributed as dist
    from golden_util import build_inputs, compare_to_golden, grad_tol, load_case, worst_grad_err
    from model_util import run_model_step
    from rspnet_amd import ops
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC, see DESIGN.md §6 "Environment"
    torch.cuda.set_device(rank)
    dev = torch.device("cuda", rank)
    dist.init_process_group("nccl", rank=rank, world_size=ws, device_id=dev)
    assert ops.backend().name == "hip"
    z, meta = load_case(arch, ws, seed)
    spec, inputs = build_inputs(arch, meta)
    res, post, mom_post, grads = run_model_step(arch, meta, inputs, rank, dev, "fused")
    gate = grad_tol(arch, ws)
    compare_to_golden(z, rank, res, post, mom_post, tol=1e-3, tol_grad=gate)
    wkey, worst = worst_grad_err(z, rank, grads)
    assert worst <= gate, (wkey, worst)
    np.save(os.path.join(tmp, f"ok{rank}.npy"), np.array([worst]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs two GPUs (RCCL)")
@pytest.mark.parametrize("arch,seed", [(a, s) for arch in ("c3d", "c3d:linear:4", "resnet18", "r2plus1d-vcop", "s3dg")
                                       for a, w, s in cases_for(arch, 2)])
def test_two_ranks_over_rccl_match_the_ddp_fixture(arch, seed):
    from oracle.ref_harness import _free_port
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(2, arch, seed, _free_port(), tmp), nprocs=2, join=True)
        assert os.path.exists(os.path.join(tmp, "ok0.npy")) and os.path.exists(os.path.join(tmp, "ok1.npy"))


@pytest.mark.parametrize("arch", ["c3d", "s3dg"])
def test_rccl_one_rank_forced_collectives_match_the_fixture(arch):
    """RCCL itself, on the one GPU the box has: see forced_dp_util.forced_worker."""
    import json
    from forced_dp_util import forced_worker
    from oracle.ref_harness import _free_port
    a, _, seed = cases_for(arch, 1)[0]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "ok.json")
        mp.spawn(forced_worker, args=("nccl", a, seed, _free_port(), out), nprocs=1, join=True)
        with open(out) as f:
            rep = json.load(f)
    print("\n", arch, rep)
    assert rep["side_group"] == "gloo"
    assert rep["calls"]["all_to_all_single"] == 2 and rep["calls"]["all_gather_into_tensor"] == 1 and rep["calls"]["all_reduce"] >= 2


@pytest.mark.parametrize("arch,B,HW,mode", [("c3d", 4, 32, "lanes"), ("s3dg", 4, 64, "lanes"), ("s3dg", 4, 64, "segments")])
def test_rccl_one_rank_segmented_replay_equals_the_eager_dp_step(arch, B, HW, mode):
    """The N > 1 issue mode that is not Python-bound: with the collectives on, GraphedPretextStep replays the
    step as HIP graphs between its collective points (RCCL calls issued eagerly in between) — "lanes": linear graphs, the
    three forward passes side by side on three streams; "segments": four graphs with the forks inside — bit-identical to the eager
    data-parallel loop over seven steps, two of them eager warm-ups.  The ORDER of the replayed step's collectives is checked by
    value: each gradient bucket is snapshot on the issuing stream when its all-reduce is issued and must equal the
    bucket's content at the end of the step, and the "all-reduce" doubles the bucket (a sum over two identical ranks), which the SGD
    graph of the replayed step sees exactly when it is ordered behind it.  See forced_dp_util.segmented_worker."""
    import json
    from forced_dp_util import segmented_worker
    from oracle.ref_harness import _free_port
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "ok.json")
        mp.spawn(segmented_worker, args=(arch, B, HW, _free_port(), out, mode), nprocs=1, join=True)
        with open(out) as f:
            rep = json.load(f)
    print("\n", arch, rep)
    if mode == "lanes":
        # top | query | key_k | key_kneg | keys_join | tail | backward pieces (one per 64 MiB gradient bucket) | update; the collective
        # slots: all-to-all (+ its wait on the k lane), all-gather, one all-reduce per bucket, the wait for them
        assert rep["lanes"] in (["k", "main", "q"], ["k", "main", "q", "w"]) and rep["graphs"] >= 8 and rep["collective_points"] >= 5, rep
    else:
        # (collective points: DDP's buffer broadcast in front of the step, all-to-all, all-gather, all-reduce)
        assert (rep["graphs"], rep["lanes"], rep["collective_points"]) == (4, ["main"], 4), rep
    # every step of either loop issues its buffer broadcast (BN running statistics, one flat tensor), its 2 clip all-to-alls and its
    # 1 key all-gather
    assert rep["buckets"]["eager"] >= 1 and rep["buckets"]["segments"] >= 1, rep
    for m in ("eager", "segments"):
        assert rep[m]["all_to_all_single"] == 2 * 7 and rep[m]["all_gather_into_tensor"] == 7 and rep[m]["all_reduce"] >= 7
        assert rep[m]["broadcast_buffers"] == 7, rep


def test_bucket_all_reduce_is_ordered_behind_the_side_stream_weight_gradients():
    """the data-parallel step runs weight gradients on a side stream and issues each bucket's all-reduce from a
    stream context ordered behind trunk AND side stream.  With one rank the all-reduce moves nothing, so the ordering is checked
    directly: see forced_dp_util.bucket_order_worker (NaN-poisoned gradient buffer, snapshot at issue == final content)."""
    import json
    from forced_dp_util import bucket_order_worker
    from oracle.ref_harness import _free_port
    a, _, seed = cases_for("c3d", 1)[0]
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "ok.json")
        mp.spawn(bucket_order_worker, args=(a, seed, _free_port(), out), nprocs=1, join=True)
        with open(out) as f:
            rep = json.load(f)
    print("\n", rep)
    assert rep["buckets"] >= 2 and rep["issued_from_side_stream"] >= 1
