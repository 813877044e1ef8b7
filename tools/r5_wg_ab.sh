#!/bin/bash
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
for a in s3dg resnet18; do
for f in 1 0 1 0; do
  if [ $f = 1 ]; then export RSP_NO_HALF_BLOCK=1; else unset RSP_NO_HALF_BLOCK; fi
  python bench.py --arch $a --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$a', 'half off' if $f else 'half on', d['value'], d.get('step_issue_mode'))"
done
done
unset RSP_NO_HALF_BLOCK
bash tools/r5_verify.sh
