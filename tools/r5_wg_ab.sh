#!/bin/bash
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
cd "$R"
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -3
for v in 1 0; do
  if [ $v = 1 ]; then export RSP_NO_HALF_BLOCK=1; else unset RSP_NO_HALF_BLOCK; fi
  echo "== RSP_NO_HALF_BLOCK=$v"
  python tools/geom_bench.py wg wgrad 2>&1 | grep -v amdgpu.ids | head -4
done
for f in 1 0 1 0; do
  if [ $f = 1 ]; then export RSP_NO_HALF_BLOCK=1; else unset RSP_NO_HALF_BLOCK; fi
  python bench.py --arch r2plus1d-vcop --no-cpu-baseline --no-other-workloads 2>/dev/null | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('half off' if $f else 'half on', d['value'])"
done
