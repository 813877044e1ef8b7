#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r6h; mkdir -p $O
for pz in 0 40 90; do
  RSP_BWD_PIECE=$pz python3 bench.py --arch s3dg --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1])
print('s3dg piece$pz', d['value'], d['ms_per_step'], json.dumps(d['steps_ms'].get('segment_gpu_p50')))" >> $O/gaps.txt
done
cat $O/gaps.txt
