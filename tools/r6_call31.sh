#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6fin2; mkdir -p $O
soak() {  # tag arch steps extra env...
  local tag=$1 a=$2 n=$3 extra=$4; shift 4
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps $n --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a $tag steps $n', d['value'], d['ms_per_step'], d.get('step_issue_mode'), 'final_loss', d['final_loss'])" >> $O/soak.txt
}
ONE="RSP_NO_EAGER_OVERLAP=1 RSP_NO_QOVERLAP=1 RSP_NO_KOVERLAP=1"
soak "one stream" resnet18 400 "--graph off" $ONE
soak "one stream" s3dg 300 "--graph off" $ONE
soak "one stream" r2plus1d-vcop 150 "--graph off" $ONE
soak "one stream" c3d 100 "--graph off" $ONE
soak "whole graph" s3dg 300 "--graph on" RSP_GRAPH_MODE=whole
soak "segments" resnet18 400 "--graph on" RSP_GRAPH_MODE=segments
cat $O/soak.txt
