#!/bin/bash
# which weight gradients go to the "w" lane now that it runs on a hardware queue of its own: re-sweep of engine.BranchStreams' thresholds
cd "$(dirname "$0")/.."
O=gpurun_out/r6n; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), len([k for k in g if 'gap' not in k]), d['steps_ms'].get('host_issue_idle_gpu_p50'))" >> $O/sweep.txt
}
for rep in 1 2; do
  for g in 0 20 50 120 300 100000; do
    run "aside$(printf %06d $g)" s3dg "" RSP_WGRAD_ASIDE_GFLOP=$g RSP_WGRAD_MID_GFLOP=$(( g > 400 ? g : 400 ))
  done
  run "aside000050 mid0" s3dg "" RSP_WGRAD_MID_GFLOP=0
  for a in resnet18 r2plus1d-vcop c3d; do
    for g in 50 150; do
      run "aside$(printf %06d $g)" $a "" RSP_WGRAD_ASIDE_GFLOP=$g
    done
    run "aside000050 mid1000" $a "" RSP_WGRAD_MID_GFLOP=1000 RSP_WGRAD_MID_MB=900
    run "aside000050 mid0" $a "" RSP_WGRAD_MID_GFLOP=0
  done
done
sort $O/sweep.txt
