#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6m; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), len([k for k in g if 'gap' not in k]), d['steps_ms'].get('host_issue_idle_gpu_p50'))" >> $O/sweep.txt
}
for rep in 1 2; do
  for pz in 4 6 8 10 12 15; do
    run "piece$(printf %02d $pz)" s3dg "" RSP_BWD_PIECE=$pz
  done
  for pz in 6 10 15; do
    run "dp piece$(printf %02d $pz)" s3dg "--force-dp" RSP_BWD_PIECE=$pz
  done
  for pz in 4 6 8 12; do
    run "dp piece$(printf %02d $pz)" resnet18 "--force-dp" RSP_BWD_PIECE=$pz
  done
done
sort $O/sweep.txt
