#!/bin/bash
# GPU box: whole-step A/B over the VALUES of one environment variable:  bash tools/ab_step_val.sh VAR "v1 v2 ..." [archs...]   ("-" = unset)
set -u
VAR="$1"; VALS="$2"; shift 2
ARCHS=("$@"); [ ${#ARCHS[@]} -eq 0 ] && ARCHS=(c3d resnet18 r2plus1d-vcop s3dg)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
for a in "${ARCHS[@]}"; do
  for rep in 1 2; do for v in $VALS; do
    if [ "$v" = "-" ]; then unset $VAR; else export $VAR=$v; fi
    python3 bench.py --arch $a --no-cpu-baseline --no-other-workloads --steps 30 --warmup 8 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); r=d['roofline']
print('$a $VAR=$v', d['value'], d['ms_per_step'], 'conv ms', r['all_conv_launches']['ms_per_step'], 'launches', r['all_conv_launches']['launches'], d['steps_ms'].get('host_submit_p50'))"
  done; done
done
