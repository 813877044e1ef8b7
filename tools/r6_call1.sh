#!/bin/bash
# round 6, GPU call 1: new tile instances (256 x 64; 32-wide instead of a K split) — kernel tests, per-layer A/B, per-step A/B; parity-seed scan
cd "$(dirname "$0")/.."
O=gpurun_out/r6a; mkdir -p $O
timeout 1200 python -m pytest tests/test_abi.py tests/test_kernels_gpu.py tests/test_persistent_gpu.py -x -q -m gpu -k "conv or persistent or abi or k_split" > $O/tests.log 2>&1; echo "tests rc $?" >> $O/tests.log
tail -5 $O/tests.log
for m in fwd dgrad; do
  timeout 300 python tools/geom_bench.py tall $m --opt tall_min_tiles=0,768 > $O/geom_tall_$m.txt 2>&1
  timeout 300 python tools/geom_bench.py tiny $m --opt narrow32_max_units=0,256,512 > $O/geom_tiny_$m.txt 2>&1
done
timeout 300 python tools/geom_bench.py s3dg14 fwd --opt narrow32_max_units=0,256,512 > $O/geom_s3dg14_fwd.txt 2>&1
cat $O/geom_tall_fwd.txt $O/geom_tall_dgrad.txt $O/geom_tiny_fwd.txt | tail -60
run() {  # arch tag env...
  local a=$1 tag=$2; shift 2
  env "$@" python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$a $tag', d['value'], d['ms_per_step'], 'conv ms', r['all_conv_launches']['ms_per_step'], 'launches', r['all_conv_launches']['launches'], d.get('steps_ms',{}).get('segment_gpu_p50'), {k.replace('igemm_persist_kernel','P').replace('wgrad_dma_kernel','W'):(v['tflops'],v['ms_per_step']) for k,v in list(r['per_kernel'].items())[:6]})" >> $O/ab_step.txt
}
for a in resnet18 s3dg; do
  run $a base RSP_TALL_MIN_TILES=0 RSP_NARROW32_MAX_UNITS=0
  run $a both X=1
  run $a tall RSP_NARROW32_MAX_UNITS=0
  run $a n32 RSP_TALL_MIN_TILES=0
  run $a n32_512 RSP_TALL_MIN_TILES=0 RSP_NARROW32_MAX_UNITS=512
  run $a base RSP_TALL_MIN_TILES=0 RSP_NARROW32_MAX_UNITS=0
  run $a both X=1
done
for a in r2plus1d-vcop c3d; do
  run $a base RSP_TALL_MIN_TILES=0 RSP_NARROW32_MAX_UNITS=0
  run $a both X=1
  run $a base RSP_TALL_MIN_TILES=0 RSP_NARROW32_MAX_UNITS=0
  run $a both X=1
done
cat $O/ab_step.txt
for s in 1 2 3 4 5 6 7 8; do
  python3 bench.py --gpus 1 --steps 2 --warmup 1 --batch 4 --hw 32 --queue 64 --cpu-sample 4 --cpu-steps 1 --seed $s 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); p=d['parity']
print('seed $s', p['grad_rel_l2'], p['grad_floor_rel_l2'], p['grad_ok'], p['forward_ok'])" >> $O/parity_seeds.txt
done
cat $O/parity_seeds.txt
