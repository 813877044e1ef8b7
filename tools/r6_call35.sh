#!/bin/bash
# the end of the backward chain: last weight gradient inline, the big one before it aside (bytes cap of the mid rule raised)
cd "$(dirname "$0")/.."
O=gpurun_out/r6tail2; mkdir -p $O
run() {  # tag arch extra env...
  local tag=$1 a=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); g=d['steps_ms'].get('segment_gpu_p50') or {}
segs={k:v for k,v in g.items() if 'gap' not in k and ('backward' in k or 'wgrad' in k)}
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), len([k for k in g if 'gap' not in k]), list(segs.items())[-8:])" >> $O/tail2.txt
}
for rep in 1 2; do
  run "base" s3dg ""
  run "last-inline" s3dg "" RSP_LAST_WGRAD_INLINE=1
  run "last-inline mid900" s3dg "" RSP_LAST_WGRAD_INLINE=1 RSP_WGRAD_MID_MB=900
  run "last-inline mid700" s3dg "" RSP_LAST_WGRAD_INLINE=1 RSP_WGRAD_MID_MB=700
  run "mid900" s3dg "" RSP_WGRAD_MID_MB=900
  run "last-inline mid900 tail30" s3dg "" RSP_LAST_WGRAD_INLINE=1 RSP_WGRAD_MID_MB=900 RSP_BWD_TAIL_NODES=30
  for a in resnet18 r2plus1d-vcop c3d; do
    run "base" $a ""
    run "last-inline" $a "" RSP_LAST_WGRAD_INLINE=1
  done
done
sort $O/tail2.txt | cut -c1-420
