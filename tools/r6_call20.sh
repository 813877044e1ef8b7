#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6t; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
segs={k:v for k,v in g.items() if ('backward' in k or 'wgrad' in k) and ('gap' not in k or '-> main:backward' in k)}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), len([k for k in g if 'gap' not in k]), list((d.get('steps_ms',{}).get('segment_host_p50') or {}).keys())[-14:], segs)" >> $O/late.txt
}
for rep in 1 2; do
  for late in 0 1; do
    run "dp late$late cut060" s3dg "--force-dp" RSP_REDUCE_LATE=$late RSP_BWD_TAIL_CUT_GFLOP=60
    run "dp late$late cut100" resnet18 "--force-dp" RSP_REDUCE_LATE=$late RSP_BWD_TAIL_CUT_GFLOP=100
    run "dp late$late cut000" resnet18 "--force-dp" RSP_REDUCE_LATE=$late
  done
done
sort $O/late.txt
