#!/bin/bash
# the weight gradients of the last backward piece run alone behind the end of the chain: a cut right behind a big one in the final stretch
cd "$(dirname "$0")/.."
O=gpurun_out/r6s; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
segs={k:v for k,v in g.items() if 'gap' not in k and ('backward' in k or 'wgrad' in k)}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), len([k for k in g if 'gap' not in k]), segs)" >> $O/tail.txt
}
for rep in 1 2; do
  for c in 0 20 60; do
    run "cut$(printf %03d $c)" s3dg "" RSP_BWD_TAIL_CUT_GFLOP=$c
  done
  run "cut020 tail30" s3dg "" RSP_BWD_TAIL_CUT_GFLOP=20 RSP_BWD_TAIL_NODES=30
  run "cut060 tail30" s3dg "" RSP_BWD_TAIL_CUT_GFLOP=60 RSP_BWD_TAIL_NODES=30
  run "dp cut000" s3dg "--force-dp" RSP_BWD_TAIL_CUT_GFLOP=0
  run "dp cut060" s3dg "--force-dp" RSP_BWD_TAIL_CUT_GFLOP=60
  for c in 0 100 300; do
    run "dp cut$(printf %03d $c)" resnet18 "--force-dp" RSP_BWD_TAIL_CUT_GFLOP=$c
  done
  run "graph cut000" resnet18 "--graph on" RSP_BWD_TAIL_CUT_GFLOP=0
  run "graph cut100" resnet18 "--graph on" RSP_BWD_TAIL_CUT_GFLOP=100
done
sort $O/tail.txt | cut -c1-400
