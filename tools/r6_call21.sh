#!/bin/bash
# eager issue: the weight-gradient lane runs ahead (no join with the trunk before the next task)
cd "$(dirname "$0")/.."
O=gpurun_out/r6u; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), d['steps_ms'].get('host_issue_idle_gpu_p50'))" >> $O/ahead.txt
}
for rep in 1 2; do
  for ra in 0 1; do
    for a in resnet18 r2plus1d-vcop c3d; do
      run "ahead$ra" $a "" RSP_TASK_RUN_AHEAD=$ra
    done
    run "dp ahead$ra" c3d "--force-dp" RSP_TASK_RUN_AHEAD=$ra
    run "dp ahead$ra" r2plus1d-vcop "--force-dp" RSP_TASK_RUN_AHEAD=$ra
  done
  run "graph cut100" resnet18 "--graph on" RSP_BWD_TAIL_CUT_GFLOP=100
  run "graph cut100" r2plus1d-vcop "--graph on" RSP_BWD_TAIL_CUT_GFLOP=100
  run "graph cut100" c3d "--graph on" RSP_BWD_TAIL_CUT_GFLOP=100
  run "graph cut000" r2plus1d-vcop "--graph on"
  run "graph cut000" c3d "--graph on"
done
sort $O/ahead.txt
