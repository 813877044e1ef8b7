#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5u; mkdir -p $O
for arch in s3dg resnet18 r2plus1d-vcop; do
  for v in A B A B; do
    if [ "$v" = A ]; then unset RSP_NARROW_128; else export RSP_NARROW_128=1; fi
    python3 bench.py --arch $arch --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$arch $v', d['value'], d['ms_per_step'], 'conv ms', r['all_conv_launches']['ms_per_step'])"
  done
done > $O/ab_narrow128.txt 2>&1; cat $O/ab_narrow128.txt
