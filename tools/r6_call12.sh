#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6k; mkdir -p $O
timeout 900 python -m pytest tests/test_graph_step_gpu.py tests/test_rccl_gpu.py -x -q -m gpu -p no:cacheprovider > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -4 $O/tests.log
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), d.get('lanes_overlap_main'), {k:v for k,v in g.items() if 'gap w:' in k})" >> $O/lanes.txt
}
for rep in 1 2 3; do
  run "default" s3dg "" X=1
  run "uncut" s3dg "" RSP_BWD_PIECE=0
  run "piece15" s3dg "" RSP_BWD_PIECE=15
  run "dp default" s3dg "--force-dp" X=1
  run "default" resnet18 "" X=1
  run "dp default" resnet18 "--force-dp" X=1
  run "default" c3d "" X=1
  run "default" r2plus1d-vcop "" X=1
done
run "dp default" c3d "--force-dp" X=1
run "dp default" r2plus1d-vcop "--force-dp" X=1
sort $O/lanes.txt
