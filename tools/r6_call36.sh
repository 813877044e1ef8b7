#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6occ; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), 'conv ms (one stream)', r['all_conv_launches']['ms_per_step'])" >> $O/occ.txt
}
for a in r2plus1d-vcop resnet18 s3dg c3d; do
  run "base" $a ""
  run "wpc2" $a "" RSP_PERSIST_WPC_MAX=2
  run "wpc1" $a "" RSP_PERSIST_WPC_MAX=1
  run "spare16" $a "" RSP_PERSIST_SPARE_CUS=16
  run "spare32" $a "" RSP_PERSIST_SPARE_CUS=32
  run "base" $a ""
done
sort $O/occ.txt
