"""GPU: bench.py honours the driver's contract -- one JSON line on stdout with the agreed keys, value = clips / time, the
roofline and cpu_baseline objects -- checked on a 2-step run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--cpu-sample", "2", "--cpu-steps", "1",
                        "--other-steps", "2", "--other-warmup", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
              "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["unit"] == "clips/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 32 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]          # whole-job clips/s of B=32
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and 0.3 < rf["frac"] < 1.0
    assert rf["traffic"] is None or rf["traffic"] > 0
    # the line carries its own evidence: the executed share of the algorithmic FLOPs, the PMC matrix-pipe busy
    # share with its source, traffic marked as a static lookup
    assert 0.5 < rf["executed_over_algorithmic"] <= 1.0 and abs(rf["executed_frac"] - rf["frac"] * rf["executed_over_algorithmic"]) < 2e-3
    assert rf["traffic_static"] is True and (rf["mfma_busy"] is None or 0.3 < rf["mfma_busy"] <= 1.0)
    # ... and tied to the build the counters were collected on: a hash of rspnet_amd/csrc + include
    assert "traffic_stale" in rf and rf["traffic_stale"] in (True, False, None)
    if rf["traffic"] is not None:
        assert isinstance(rf["traffic_stale"], bool) and "profiled_csrc_sha256" in rf["traffic_build"]
    assert rf["whole_step"]["executed_frac"] <= rf["whole_step"]["frac"]
    hk = d["hbm_kernels"]
    assert hk["peak_tb_s"] == 8.0 and {"bn_act_pool_fwd", "bn_bwd(reduce+apply)", "sgd_step", "momentum_update", "clip_gather"} <= set(hk["groups"])
    assert all(0 < g["tb_s"] < 8.0 for g in hk["groups"].values())
    # the same step issued the way N > 1 ranks issue it, and the data-parallel path itself on this GPU (RCCL group of one rank)
    if d["config"]["step_issue"].startswith("one replayed"):      # (a 1-step warm-up does not reach the capture: see test_graph_step_gpu)
        assert d["issued_eagerly"]["clips_per_s"] > 0
    dp = d["dp_path_at_one_rank"]
    assert "error" not in dp and dp["clips_per_s"] > 0 and {"all_to_all_kneg", "all_to_all_k", "all_gather_keys", "allreduce_wait"} <= set(dp["comm_ms"])
    # the data-parallel line diagnoses itself: who is in the job, how the step is issued, what issuing it costs
    # the host — and the N = 1 number in that issue mode, to divide N > 1 values by
    assert dp["rccl_ranks"] == {"world_size": 1, "distinct_devices": 1, "backend": "nccl", "host_cpus_per_rank": dp["rccl_ranks"]["host_cpus_per_rank"],
                                "pinned": False}
    assert dp["step_issue_mode"] in ("eager", "graph_segments", "graph_lanes") and dp["host_issue_idle_gpu_p50"] > 0
    assert d["n1_same_mode"]["clips_per_s"] == dp["clips_per_s"] and d["n1_same_mode"]["step_issue_mode"] == dp["step_issue_mode"]
    assert d["step_issue_mode"] in ("eager", "graph", "graph_lanes") and d["steps_ms"]["host_issue_idle_gpu_p50"] > 0
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "clips/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["cpu_model"] and cb["s_per_step"] > 0
    # the roofline block names the kernel that dominates THIS backbone and carries the per-kernel table
    # (the 128x128 tile: its channel-slice-major instance runs conv3-5 and their input gradients, the tap-major one conv2)
    assert rf["kernel"].startswith(("igemm_persist_kernel<128, 128", "igemm_ks_kernel<128, 128", "igemm_kernel<128, 128")) and rf["kernel"] in rf["per_kernel"]
    assert abs(rf["avg_launch_ms"] - rf["per_kernel"][rf["kernel"]]["avg_launch_ms"]) < 1e-3 and 0 < rf["share_of_step"] < 1
    # traffic is priced against the algorithmic bytes of the same launches; the whole-step fraction is in the line
    assert rf["algorithmic_gb_per_launch"] > 0 and "L2-miss" in rf["traffic_unit"] and 0.2 < rf["whole_step"]["frac"] < 1.0
    # per-step distribution (GPU-side intervals) next to the mean
    sm = d["steps_ms"]
    assert sm["min"] <= sm["p50"] <= sm["max"] and sm["host_enqueue_p50"] > 0
    # BASELINE configs 3-5 ride on the default line
    ow = d["other_workloads"]
    assert set(ow) == {"resnet18", "r2plus1d-vcop", "s3dg"}
    for a, o in ow.items():
        assert o["clips_per_s"] > 0 and 0.1 < o["whole_step_frac"] < 1.0 and o["dominant_kernel"], a
        # every backbone also runs the way N > 1 ranks run it: RCCL group of one rank, all collectives on
        odp = o["dp_path_at_one_rank"]
        assert "error" not in odp and odp["clips_per_s"] > 0 and odp["step_issue_mode"] in ("eager", "graph_segments", "graph_lanes"), (a, odp)
    # ... and as flat scalars, at the top level and inside the roofline object (a record that keeps only the
    # contract's keys and objects still holds the numbers of BASELINE configs 3-5)
    for key, a in (("resnet18", "resnet18"), ("r2plus1d", "r2plus1d-vcop"), ("s3dg", "s3dg")):
        assert d[f"{key}_clips_per_s"] == ow[a]["clips_per_s"] and d[f"{key}_whole_step_frac"] == ow[a]["whole_step_frac"]
        assert d[f"{key}_dominant_kernel_frac"] == ow[a]["dominant_kernel_frac"] and d[f"{key}_ms_per_step"] == ow[a]["ms_per_step"]
        assert rf["other_workloads"][f"{key}_clips_per_s"] == ow[a]["clips_per_s"]
        assert rf["other_workloads"][f"{key}_whole_step_frac"] == ow[a]["whole_step_frac"]
    assert "parity" not in d                       # --cpu-sample 2 != B: the CPU leg cannot replay the GPU's step


def test_bench_parity_object_replays_the_first_gpu_step():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--hw", "32", "--queue", "64", "--cpu-sample", "4", "--cpu-steps", "1", "--seed", "4"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    p = d["parity"]
    assert p["forward_ok"] is True and p["grad_floor_rel_l2"] is not None and p["grad_ok"] is not None, p
    assert max(p["loss_rel"], p["logits_rel"], p["features_rel"], p["queue_slab_rel"]) <= 1e-3, p
    # The whole gradient of a 4-clip state whose floor is 6e-6 (no evaluation order flips a ReLU mask or an arg-max on it) under the
    # line's own rule: 3 floors, never below 1e-4 — measured 1.1e-5.  The seed is chosen (profiles/r06/experiments_r6.txt, r6a (d):
    # of eight seeds three are free of knife edges on both sides, on two the GPU flips one mask, on three the CPU oracle does); the
    # default seed 1234 of this size has an element of conv3a's output within rounding of zero, which is why rounds 4-5 accepted 5e-3 here.
    assert p["grad_ok"] is True and p["ok"] is True and p["grad_rel_l2"] <= 1e-4, p
    assert "other_workloads" not in d


def test_bench_roofline_is_arch_aware():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--arch", "resnet18", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert "resnet18" in d["metric"] and "cpu_baseline" not in d
    assert d["roofline"]["kernel"].startswith(("igemm_persist_kernel<128, 64", "igemm_kernel<128, 64"))      # R3D-18's layer-1 / stem launches dominate, not <128,128>
