#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6long; mkdir -p $O
soak() {  # tag arch steps extra env...
  local tag=$1 a=$2 n=$3 extra=$4; shift 4
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps $n --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('$a $tag steps $n', d['value'], d['ms_per_step'], d.get('step_issue_mode'), 'final_loss', d['final_loss'])" >> $O/soak.txt
}
soak "eager" resnet18 3000 "--graph off"
soak "lanes" resnet18 3000 "--graph on"
soak "lanes dp" resnet18 3000 "--force-dp"
soak "eager" s3dg 2000 "--graph off"
soak "lanes" s3dg 2000 ""
soak "lanes dp" s3dg 2000 "--force-dp"
cat $O/soak.txt
python3 -m pytest tests/test_graph_step_gpu.py -q -k "measured_size" 2>&1 | tail -2
