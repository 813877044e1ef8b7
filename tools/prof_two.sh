R="$GRAFT_REPO_ROOT"; OUT="$R/gpurun_out"; cd /tmp && export TMPDIR=/tmp
for a in resnet18; do
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof_r2d_$a" -- python3 "$R/bench.py" --arch "$a" --steps 10 --warmup 3 --no-cpu-baseline > "$OUT/bench_under_rocprof_r2d_$a.json" 2> "$OUT/prof_r2d_$a.err"
  find "$OUT/prof_r2d_$a" -name '*_kernel_trace.csv' -delete
done
