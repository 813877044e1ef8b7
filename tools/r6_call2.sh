#!/bin/bash
# round 6, GPU call 2: backward pieces + weight-gradient lane at one rank (S3D-G), two-level K summation (C3D), re-run of the instance tests
cd "$(dirname "$0")/.."
O=gpurun_out/r6b; mkdir -p $O
timeout 1500 python -m pytest tests/test_abi.py tests/test_kernels_gpu.py tests/test_persistent_gpu.py tests/test_graph_step_gpu.py -x -q -m gpu -k "conv or persistent or abi or k_split or graphed_step" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
run() {  # arch tag extra-args env...
  local a=$1 tag=$2 extra=$3; shift 3
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 $extra 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$a $tag', d['value'], d['ms_per_step'], d.get('step_issue_mode'), 'conv ms', r['all_conv_launches']['ms_per_step'], d.get('steps_ms',{}).get('segment_gpu_p50'))" >> $O/ab_pieces.txt
}
for pz in 0 25 40 60 90 0 40; do
  run s3dg piece$pz "" RSP_BWD_PIECE=$pz
done
for pz in 0 12 0 12; do
  run resnet18 graph_piece$pz "--graph on" RSP_BWD_PIECE=$pz
done
run resnet18 eager "" X=1
for pz in 0 40; do
  run s3dg dp_piece$pz "--force-dp" RSP_BWD_PIECE=$pz
done
cat $O/ab_pieces.txt
# two-level summation over K on C3D's long-K layers
python3 - > $O/geom_two_level.txt 2>&1 <<'PY'
import sys, os
sys.argv = ["geom_bench", "c3dlong", "fwd", "--opt", "two_level_min_chunks=0,100"]
sys.path.insert(0, "tools")
import importlib.util
src = open("tools/geom_bench.py").read().replace('"stem": [', '"c3dlong": [(32, 8, 28, 28, 128, 256, (3, 3, 3)), (32, 8, 28, 28, 256, 256, (3, 3, 3)), (32, 4, 14, 14, 256, 512, (3, 3, 3)), (32, 4, 14, 14, 512, 512, (3, 3, 3)), (32, 16, 56, 56, 64, 128, (3, 3, 3))],\n    "stem": [')
exec(compile(src, "tools/geom_bench.py", "exec"))
PY
cat $O/geom_two_level.txt | tail -8
for v in 0 100; do
  RSP_TWO_LEVEL_MIN_CHUNKS=$v python3 bench.py --steps 20 --warmup 5 --no-other-workloads --grad-floor live 2>$O/two_level_$v.err | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['parity']; r=d['roofline']
print('c3d two_level=$v', d['value'], d['ms_per_step'], 'grad_rel_l2', p['grad_rel_l2'], 'floor', p['grad_floor_rel_l2'], 'vs_fp64', p['grad_vs_fp64_rel_l2'], 'logits', p['logits_rel'], {k:(x['tflops'],x['ms_per_step']) for k,x in list(r['per_kernel'].items())[:4]})" >> $O/two_level.txt
done
cat $O/two_level.txt
