#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6i; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), {k:v for k,v in g.items() if 'gap' not in k}, {k:v for k,v in g.items() if 'gap w:' in k or '-> w:' in k})" >> $O/hwq.txt
}
for q in 4 8 16; do
  run "hwq$q piece40" s3dg "" GPU_MAX_HW_QUEUES=$q RSP_BWD_PIECE=40
  run "hwq$q piece0" s3dg "" GPU_MAX_HW_QUEUES=$q RSP_BWD_PIECE=0
done
run "default piece40" s3dg "" RSP_BWD_PIECE=40
for q in 4 8; do
  run "hwq$q" resnet18 "" GPU_MAX_HW_QUEUES=$q
  run "hwq$q" c3d "" GPU_MAX_HW_QUEUES=$q
  run "hwq$q" r2plus1d-vcop "" GPU_MAX_HW_QUEUES=$q
  run "hwq$q graph piece12" resnet18 "--graph on" GPU_MAX_HW_QUEUES=$q RSP_BWD_PIECE=12
done
cat $O/hwq.txt
