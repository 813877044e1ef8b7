#!/bin/bash
# Run ON THE GPU BOX (through gpurun): for every BASELINE backbone a bench line, a rocprofv3 kernel-trace/stats pass and the
# three PMC passes (FETCH_SIZE / WRITE_SIZE / matrix-pipe busy), each in its own run with the program directly after `--`.
#   gpurun --timeout 1500 -- 'bash tools/profile_round.sh r2a [archs...]'
# Raw outputs land under gpurun_out/; tools/summarize_profiles.py <round> <tag> turns them into profiles/<round>/.
set -u
TAG="$1"; shift
ARCHS=("$@"); [ ${#ARCHS[@]} -eq 0 ] && ARCHS=(c3d resnet18 r2plus1d-vcop s3dg)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out"
mkdir -p "$OUT"
# the build these profiles are measured on (tools/summarize_profiles.py stores it next to the counters; bench.py compares)
python3 -c "import sys; sys.path.insert(0, '$R'); from rspnet_amd import _lib; print(_lib.source_hash())" > "$OUT/csrc_hash_${TAG}.txt"
cd /tmp && export TMPDIR=/tmp
for a in "${ARCHS[@]}"; do
  # the bench line as the driver runs it for this backbone (step issue: bench.py --graph auto), then the profiled passes with the
  # step issued EAGERLY ON ONE STREAM (--graph off, RSP_NO_EAGER_OVERLAP=1): per-dispatch rows in launch order, kernel durations that
  # are the kernels' own (on side streams / as graph branches a dispatch's interval also holds its neighbours' time) — the same
  # conditions as bench.py's roofline pass, whose avg_launch_ms they must agree with; one counter pass each
  unset RSP_NO_EAGER_OVERLAP
  python3 "$R/bench.py" --arch "$a" --no-cpu-baseline --no-other-workloads > "$OUT/bench_${TAG}_$a.json" 2> "$OUT/bench_${TAG}_$a.err"
  export RSP_NO_EAGER_OVERLAP=1
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_${TAG}_$a" -- python3 "$R/bench.py" --arch "$a" --steps 10 --warmup 3 --no-cpu-baseline --no-other-workloads --graph off \
    > "$OUT/bench_under_rocprof_${TAG}_$a.json" 2> "$OUT/prof_${TAG}_$a.err"
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch_${TAG}_$a" -- python3 "$R/bench.py" --arch "$a" --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --graph off \
    > /dev/null 2> "$OUT/pmc_fetch_${TAG}_$a.err"
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write_${TAG}_$a" -- python3 "$R/bench.py" --arch "$a" --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --graph off \
    > /dev/null 2> "$OUT/pmc_write_${TAG}_$a.err"
  rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma_${TAG}_$a" -- python3 "$R/bench.py" --arch "$a" --steps 2 --warmup 1 --no-cpu-baseline --no-other-workloads --graph off \
    > /dev/null 2> "$OUT/pmc_mfma_${TAG}_$a.err"
  # keep only the small summaries of the raw traces (the per-dispatch kernel trace of 10 steps is tens of MB)
  find "$OUT/prof_${TAG}_$a" -name '*_kernel_trace.csv' -delete
  echo "$a done: $(cat "$OUT/bench_${TAG}_$a.json" | head -c 300)"
done
