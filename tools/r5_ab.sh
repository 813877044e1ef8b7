#!/bin/bash
cd "$(dirname "$0")/.."
O=gpurun_out/r5m; mkdir -p $O
timeout 600 python -m pytest tests/test_kernels_gpu.py -x -q -m gpu -k "rowgeom or clip_gather" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log; tail -3 $O/tests.log
bash tools/ab_lib.sh tools/librspnet_hip_ks48.so c3d resnet18 r2plus1d-vcop > $O/ab_ks48.txt 2>&1; cat $O/ab_ks48.txt
for v in A B A B; do
  if [ "$v" = A ]; then unset RSPNET_HIP_LIB; else export RSPNET_HIP_LIB=tools/librspnet_hip_gather_plain.so; fi
  python3 bench.py --arch c3d --no-cpu-baseline --no-other-workloads --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('gather $v', d['value'], d['hbm_kernels']['groups']['clip_gather'])"
done > $O/ab_gather.txt 2>&1; cat $O/ab_gather.txt
