#!/bin/bash
# round 6, GPU call 3: the data-parallel step (one RCCL rank, every collective on) with the backward in pieces; instance tests again
cd "$(dirname "$0")/.."
O=gpurun_out/r6c; mkdir -p $O
timeout 900 python -m pytest tests/test_persistent_gpu.py tests/test_kernels_gpu.py -x -q -m gpu -k "persistent or two_level or tall" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -4 $O/tests.log
run() {  # arch tag extra-args env...
  local a=$1 tag=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), 'host', d.get('steps_ms',{}).get('host_issue_idle_gpu_p50'), d.get('comm_ms'), d.get('steps_ms',{}).get('segment_gpu_p50'))" >> $O/ab_dp.txt
}
for rep in 1 2 3; do
  for pz in 0 40 25; do
    run s3dg dp_piece$pz "--force-dp" RSP_BWD_PIECE=$pz
  done
done
run s3dg n1 "" X=1
run s3dg n1 "" X=1
for rep in 1 2; do
  for pz in 0 12 6; do
    run resnet18 dp_piece$pz "--force-dp" RSP_BWD_PIECE=$pz
  done
done
run resnet18 n1_eager "" X=1
for pz in 0 10; do
  run r2plus1d-vcop dp_piece$pz "--force-dp" RSP_BWD_PIECE=$pz
  run c3d dp_piece$pz "--force-dp" RSP_BWD_PIECE=$pz
done
cat $O/ab_dp.txt
