#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6z2; mkdir -p $O
soak() {  # tag arch steps extra env...
  local tag=$1 a=$2 n=$3 extra=$4; shift 4
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps $n --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a $tag steps $n', d['value'], d['ms_per_step'], d.get('step_issue_mode'), 'final_loss', d['final_loss'], 'host_issue_idle', d['steps_ms'].get('host_issue_idle_gpu_p50'))" >> $O/soak.txt
}
soak "eager" s3dg 300 "--graph off"
soak "lanes" s3dg 300 ""
soak "eager" resnet18 400 "--graph off"
soak "lanes" resnet18 400 "--graph on"
cat $O/soak.txt
