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
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "clips/s" and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]
    assert cb["cpu_model"] and cb["s_per_step"] > 0
    # the roofline block names the kernel that dominates THIS backbone and carries the per-kernel table
    # (the 128x128 tile: its channel-slice-major instance runs conv3-5 and their input gradients, the tap-major one conv2)
    assert rf["kernel"].startswith(("igemm_ks_kernel<128, 128", "igemm_kernel<128, 128")) and rf["kernel"] in rf["per_kernel"]
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
    assert "parity" not in d                       # --cpu-sample 2 != B: the CPU leg cannot replay the GPU's step


def test_bench_parity_object_replays_the_first_gpu_step():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "2", "--warmup", "1", "--batch", "4",
                        "--hw", "32", "--queue", "64", "--cpu-sample", "4", "--cpu-steps", "1"],
                       capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    p = d["parity"]
    assert p["ok"] is True, p
    assert max(p["loss_rel"], p["logits_rel"], p["features_rel"], p["queue_slab_rel"]) <= 1e-3 and p["grad_rel_l2"] <= 2e-2, p
    assert "other_workloads" not in d


def test_bench_roofline_is_arch_aware():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--arch", "resnet18", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert "resnet18" in d["metric"] and "cpu_baseline" not in d
    assert d["roofline"]["kernel"].startswith("igemm_kernel<128, 64")      # R3D-18's layer-1 / stem launches dominate, not <128,128>
