"""CPU: `python bench.py --gpus N` starts its own ranks (the driver's entry point for the 1/2/4/8-GPU curve).  The launcher
path is run end to end without a GPU: the parent spawns 2 child interpreters with a tcp://127.0.0.1 rendezvous, the children
form a gloo group and run the pretext step on the tests' checker backend at a tiny size (--selftest-cpu), rank 0 prints
the single contract line.  What this pins: argument forwarding, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* plumbing, the MAX-over-ranks
timing + single JSON line, and a non-zero exit when a rank fails (reference: pretrain.py:278-283,335-336)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")
TINY = ["--selftest-cpu", "--arch", "c3d", "--batch", "2", "--hw", "16", "--queue", "64", "--steps", "2", "--warmup", "1"]


def _env():
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    return env


def test_bench_self_launches_two_ranks():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + TINY, capture_output=True, text=True, timeout=600, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["config"]["global_batch"] == 4
    assert d["steps"] == 2 and d["warmup"] == 1 and d["scaling"] == "weak" and d["data"] == "selftest-cpu"
    assert abs(d["value"] - 4 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]
    assert "cpu_baseline" not in d                     # rank 0 at N=1 only
    # the N > 1 line explains itself: per-step stall of rank 0 behind each collective (both clip exchanges, the one key all-gather
    # and the wait for the bucketed gradient all-reduce), and the per-step distribution is absent only because this is the CPU
    cm = d["comm_ms"]
    assert {"all_to_all_kneg", "all_to_all_k", "all_gather_keys", "allreduce_wait"} <= set(cm)
    assert all(v >= 0 for v in cm.values())


def test_bench_self_launches_eight_ranks():
    """The driver's last point of the 1/2/4/8 curve, launcher and host logic end to end over gloo: eight interpreters, uneven clip
    all-to-all splits among eight peers, the key all-gather / un-shuffle map, the global queue (8 x 2 keys per step) — and the
    line says who was in the job: world size, distinct members, the step-issue mode, every rank's CPU share."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8"] + TINY, capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["config"]["parallelism"] == "dp8" and d["config"]["global_batch"] == 16
    assert abs(d["value"] - 16 / (d["ms_per_step"] * 1e-3)) <= 1e-2 * d["value"]
    rr = d["rccl_ranks"]
    assert rr["world_size"] == 8 and rr["distinct_devices"] == 8 and rr["backend"] == "gloo" and len(rr["host_cpus_per_rank"]) == 8
    assert d["step_issue_mode"] == "eager"             # (no HIP graphs on the CPU self-test)
    assert {"all_to_all_kneg", "all_to_all_k", "all_gather_keys", "allreduce_wait"} <= set(d["comm_ms"])
    import math
    assert math.isfinite(d["final_loss"])


def test_rank_cpu_sets_are_disjoint_whole_cores():
    """bench.py pins each rank's host threads to its own physical cores before the first GPU call (rank_cpu_set)."""
    sys.path.insert(0, ROOT)
    import bench
    n = len(os.sched_getaffinity(0))
    for ws in (2, 4, 8):
        sets = [bench.rank_cpu_set(r, ws) for r in range(ws)]
        if sets[0] is None:
            assert all(s is None for s in sets)        # fewer physical cores than ranks: no pinning at all, on every rank alike
            continue
        assert all(s for s in sets)
        flat = [c for s in sets for c in s]
        assert len(flat) == len(set(flat)) <= n and len({len(s) for s in sets}) == 1
    assert bench.rank_cpu_set(0, 1) is None


def test_bench_under_external_launcher_env():
    """WORLD_SIZE already set (python -m torch.distributed.run ...): the process is a rank, no second level of spawning."""
    env = _env()
    env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1"] + TINY, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "dp1"


def test_bench_launcher_reports_rank_failure():
    bad = [a if a != "16" else "1" for a in TINY]      # 1-pixel clips: every rank fails inside the model
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + bad, capture_output=True, text=True, timeout=600, cwd=ROOT, env=_env())
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_gpus_mismatch_is_an_error():
    env = _env()
    env.update({"RANK": "0", "LOCAL_RANK": "0", "WORLD_SIZE": "1"})
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2"] + TINY, capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_launcher_deadline_kills_a_hung_job():
    """A rank that never joins the group: the job is killed at the launcher's deadline and reports rc=124 instead of holding
    the node (the driver's first multi-GPU run is the first execution of the RCCL path: a hang must be diagnosable)."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-hang-rank", "1", "--deadline", "12",
                        "--collective-timeout", "600"] + TINY, capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
    assert r.returncode == 124, (r.returncode, r.stderr[-2000:])
    assert "deadline" in r.stderr and time.monotonic() - t0 < 120
    assert not [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]


def test_collective_timeout_fails_the_job():
    """Same hang, but the process-group timeout fires first: the waiting rank raises, the launcher stops the sleeper, rc != 0."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--selftest-hang-rank", "1", "--deadline", "600",
                        "--collective-timeout", "6"] + TINY, capture_output=True, text=True, timeout=300, cwd=ROOT, env=_env())
    assert r.returncode not in (0, 124), (r.returncode, r.stderr[-2000:])
    assert time.monotonic() - t0 < 200


def test_parity_leg_replays_the_first_step():
    """The line's "parity" object: the CPU leg's warm-up step replays the measured run's first step (same state, clips,
    permutations) — run here with the checker backend standing in for the GPU."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--selftest-cpu", "--selftest-parity", "--arch", "c3d", "--batch", "4",
                        "--cpu-sample", "4", "--cpu-steps", "1", "--hw", "32", "--queue", "64", "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")][0])
    p = d["parity"]
    assert p["ok"] is True and p["loss_rel"] <= 1e-4 and p["logits_rel"] <= 1e-4 and p["features_rel"] <= 1e-4
    assert p["queue_slab_rel"] <= 1e-4 and p["grad_rel_l2"] <= 1e-2
    assert d["cpu_baseline"]["value"] > 0 and d["vs_cpu_baseline"] > 0


def test_bench_under_torch_distributed_run():
    """The driver's multi-GPU command line, verbatim, with the CPU self-test standing in for the GPUs:
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 ...
    Every rank is started by torchrun (WORLD_SIZE set): bench.py must not spawn a second level, rank 0 prints the one line."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), BENCH, "--gpus", "2"] + TINY, capture_output=True, text=True, timeout=600, cwd=ROOT,
                       env=_env())
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["config"]["parallelism"] == "dp2" and d["scaling"] == "weak" and "comm_ms" in d
