#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6v; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), d['steps_ms'].get('host_issue_idle_gpu_p50'))" >> $O/ahead.txt
}
for rep in 1 2; do
  for a in resnet18 r2plus1d-vcop c3d; do
    run "ahead0" $a "" RSP_TASK_RUN_AHEAD=0
    for mx in 100 200 300; do
      run "ahead1 max$mx" $a "" RSP_TASK_RUN_AHEAD=1 RSP_TASK_AHEAD_MAX_GFLOP=$mx
    done
  done
done
sort $O/ahead.txt
