#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6l; mkdir -p $O
timeout 1500 python -m pytest tests/test_graph_step_gpu.py tests/test_rccl_gpu.py tests/test_pretrain_gpu.py -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -4 $O/tests.log
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), d.get('lanes_overlap_main'), len([k for k in g if 'gap' not in k]))" >> $O/lanes.txt
}
for rep in 1 2 3; do
  run "default" s3dg "" X=1
  run "uncut" s3dg "" RSP_BWD_PIECE=0
  run "piece10" s3dg "" RSP_BWD_PIECE=10
  run "dp default" s3dg "--force-dp" X=1
  run "dp uncut" s3dg "--force-dp" RSP_BWD_PIECE=0
  run "dp default" resnet18 "--force-dp" X=1
  run "dp uncut" resnet18 "--force-dp" RSP_BWD_PIECE=0
  run "graph default" resnet18 "--graph on" X=1
  run "default" resnet18 "" X=1
done
sort $O/lanes.txt
