#!/bin/bash
set -u
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; cd "$R"
for px in 0 1 2 3; do
  export RSP_PX=$px
  echo "== PX=$px"; python3 tools/k_sweep.py 2>/dev/null | head -4 | awk '{print $3, $7, $8, $9, $10}' | tr '\n' ';'; echo
  python3 tools/conv_bench.py --r21d --what fwd --layers c2.sp,c2.tm,c3b.tm 2>/dev/null | grep "^c" | cut -c1-60
done
