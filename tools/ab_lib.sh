#!/bin/bash
# GPU box: whole-step A/B of two builds of the library on all four backbones:  bash tools/ab_lib.sh path/to/other.so [archs...]
# (A = rspnet_amd/librspnet_hip.so, B = the other build through RSPNET_HIP_LIB; alternating runs on the same box)
set -u
OTHER="$1"; shift
ARCHS=("$@"); [ ${#ARCHS[@]} -eq 0 ] && ARCHS=(c3d resnet18 r2plus1d-vcop s3dg)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
for a in "${ARCHS[@]}"; do
  for v in A B A B; do
    if [ "$v" = A ]; then unset RSPNET_HIP_LIB; else export RSPNET_HIP_LIB="$OTHER"; fi
    python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$a $v', d['value'], d['ms_per_step'], 'conv ms', r['all_conv_launches']['ms_per_step'], {k.replace('igemm_persist_kernel','P').replace('wgrad_dma_kernel','W'):(v['tflops'],v['ms_per_step']) for k,v in list(r['per_kernel'].items())[:5]})"
  done
done
