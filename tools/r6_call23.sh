#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6w; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), len([k for k in g if 'gap' not in k]))" >> $O/cut.txt
}
export RSP_TASK_RUN_AHEAD=1 RSP_TASK_AHEAD_MAX_GFLOP=100 RSP_REDUCE_LATE=1
for rep in 1 2; do
  for c in 0 40 60 80 120; do
    run "cut$(printf %03d $c)" s3dg "" RSP_BWD_TAIL_CUT_GFLOP=$c
  done
  run "cut060 tail08" s3dg "" RSP_BWD_TAIL_CUT_GFLOP=60 RSP_BWD_TAIL_NODES=8
  run "cut060 tail11" s3dg "" RSP_BWD_TAIL_CUT_GFLOP=60 RSP_BWD_TAIL_NODES=11
  for c in 0 60 100 200; do
    run "dp cut$(printf %03d $c)" resnet18 "--force-dp" RSP_BWD_TAIL_CUT_GFLOP=$c
  done
  run "dp cut060" s3dg "--force-dp" RSP_BWD_TAIL_CUT_GFLOP=60
  run "dp cut000" s3dg "--force-dp" RSP_BWD_TAIL_CUT_GFLOP=0
done
sort $O/cut.txt
