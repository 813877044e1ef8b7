#!/bin/bash
# GPU box: whole-step A/B of an environment switch on all four backbones:  bash tools/ab_step.sh VAR [archs...]
set -u
VAR="$1"; shift
ARCHS=("$@"); [ ${#ARCHS[@]} -eq 0 ] && ARCHS=(c3d resnet18 r2plus1d-vcop s3dg)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
for a in "${ARCHS[@]}"; do
  for v in "" 1 "" 1; do
    if [ -z "$v" ]; then unset $VAR; else export $VAR=1; fi
    python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$a $VAR=${v:-0}', d['value'], d['ms_per_step'], 'conv ms', r['all_conv_launches']['ms_per_step'], {k:(v['tflops'],v['ms_per_step']) for k,v in list(r['per_kernel'].items())[:4]})"
  done
done
