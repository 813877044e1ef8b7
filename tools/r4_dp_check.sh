#!/bin/bash
# Run ON THE GPU BOX: the data-parallel path on one GPU (real RCCL group of one rank, collectives forced on).
#   gpurun --timeout 1500 -- 'bash tools/r4_dp_check.sh r4a'
set -u
TAG="$1"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/$TAG"
mkdir -p "$OUT"
cd "$R"
timeout 900 python3 -m pytest tests/test_rccl_gpu.py tests/test_two_rank_gpu.py -m gpu -x -q -s > "$OUT/pytest_dp.log" 2>&1
echo "pytest rc=$?" >> "$OUT/pytest_dp.log"
tail -5 "$OUT/pytest_dp.log"
B="--no-cpu-baseline --no-other-workloads --steps 30 --warmup 8"
for a in c3d s3dg; do
  timeout 300 python3 bench.py --arch $a $B > "$OUT/bench_graph_$a.json" 2> "$OUT/bench_graph_$a.err"
  timeout 300 python3 bench.py --arch $a $B --graph off > "$OUT/bench_eager_$a.json" 2> "$OUT/bench_eager_$a.err"
  timeout 300 python3 bench.py --arch $a $B --force-dp > "$OUT/bench_forcedp_$a.json" 2> "$OUT/bench_forcedp_$a.err"
  RSP_GRAPH_COLLECTIVES=1 timeout 300 python3 bench.py --arch $a $B --force-dp > "$OUT/bench_forcedp_graph_$a.json" 2> "$OUT/bench_forcedp_graph_$a.err"
  echo "rc(graph+collectives)=$?" >> "$OUT/bench_forcedp_graph_$a.err"
  for f in graph eager forcedp forcedp_graph; do
    python3 - "$OUT/bench_${f}_$a.json" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
    print(sys.argv[1].split("/")[-1], d["value"], d["ms_per_step"], d["config"]["step_issue"][:40], d.get("comm_ms"), d.get("issued_eagerly",{}).get("clips_per_s"), d.get("steps_ms",{}).get("host_enqueue_p50"))
except Exception as e:
    print(sys.argv[1], "ERR", e)
PY
  done
done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_forcedp_c3d" -- python3 "$R/bench.py" --arch c3d --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --force-dp \
    > "$OUT/bench_under_rocprof_forcedp_c3d.json" 2> "$OUT/prof_forcedp_c3d.err"
find "$OUT/prof_forcedp_c3d" -name '*_kernel_trace.csv' -delete
find "$OUT/prof_forcedp_c3d" -name '*kernel_stats.csv' | head -1 | xargs -I{} head -40 {}
