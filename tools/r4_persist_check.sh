#!/bin/bash
# GPU box: correctness of the persistent igemm kernels + A/B against the per-tile kernels (RSP_NO_PERSIST=1).
set -u
TAG="$1"
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
timeout 1200 python3 -m pytest tests/test_kernels_gpu.py tests/test_full_size_gpu.py -m gpu -x -q > "$OUT/pytest_kernels.log" 2>&1
echo "rc=$?" >> "$OUT/pytest_kernels.log"; tail -4 "$OUT/pytest_kernels.log"
for mode in persist classic; do
  if [ $mode = classic ]; then export RSP_NO_PERSIST=1; else unset RSP_NO_PERSIST; fi
  python3 tools/k_sweep.py > "$OUT/ksweep_$mode.txt" 2>&1
  python3 tools/conv_bench.py --r21d --what fwd,dgrad > "$OUT/convbench_r21d_$mode.txt" 2>&1
  python3 tools/conv_bench.py --what fwd,dgrad > "$OUT/convbench_c3d_$mode.txt" 2>&1
  python3 tools/conv_bench.py --s3dg --what fwd,dgrad > "$OUT/convbench_s3dg_$mode.txt" 2>&1
  python3 tools/conv_bench.py --r3d --what fwd,dgrad > "$OUT/convbench_r3d_$mode.txt" 2>&1
done
paste "$OUT/ksweep_persist.txt" "$OUT/ksweep_classic.txt" | cut -c1-200
for a in r21d c3d s3dg r3d; do echo "== $a (persist | classic)"; paste <(cut -c1-75 "$OUT/convbench_${a}_persist.txt") <(cut -c18-75 "$OUT/convbench_${a}_classic.txt"); done
