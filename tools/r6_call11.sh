#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6j; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'))" >> $O/hwq2.txt
}
for rep in 1 2 3; do
  run "hwq4 piece0" s3dg "" GPU_MAX_HW_QUEUES=4 RSP_BWD_PIECE=0
  run "hwq8 piece40" s3dg "" GPU_MAX_HW_QUEUES=8 RSP_BWD_PIECE=40
  run "hwq8 piece25" s3dg "" GPU_MAX_HW_QUEUES=8 RSP_BWD_PIECE=25
  run "hwq8 piece60" s3dg "" GPU_MAX_HW_QUEUES=8 RSP_BWD_PIECE=60
  run "hwq6 piece40" s3dg "" GPU_MAX_HW_QUEUES=6 RSP_BWD_PIECE=40
  run "hwq8 piece15" s3dg "" GPU_MAX_HW_QUEUES=8 RSP_BWD_PIECE=15
  run "hwq8 dp thirds" s3dg "--force-dp" GPU_MAX_HW_QUEUES=8
  run "hwq4 dp thirds" s3dg "--force-dp" GPU_MAX_HW_QUEUES=4
  for a in resnet18 c3d r2plus1d-vcop; do
    run "hwq4 eager" $a "" GPU_MAX_HW_QUEUES=4
    run "hwq8 eager" $a "" GPU_MAX_HW_QUEUES=8
  done
  run "hwq8 dp thirds" resnet18 "--force-dp" GPU_MAX_HW_QUEUES=8
  run "hwq4 dp thirds" resnet18 "--force-dp" GPU_MAX_HW_QUEUES=4
done
sort $O/hwq2.txt
