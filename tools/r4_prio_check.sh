#!/bin/bash
set -u
TAG="$1"; R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"; OUT="$R/gpurun_out/$TAG"; mkdir -p "$OUT"; cd "$R"
for persist in 1 0; do for prio in 0 1 2; do
  if [ $persist = 0 ]; then export RSP_NO_PERSIST=1; else unset RSP_NO_PERSIST; fi
  export RSP_PRIO=$prio
  python3 tools/k_sweep.py 2>/dev/null | head -6 > "$OUT/ks_p${persist}_q${prio}.txt"
  python3 tools/conv_bench.py --what fwd --layers conv2,conv3b,conv4b 2>/dev/null | grep conv > "$OUT/c3d_p${persist}_q${prio}.txt"
  python3 tools/conv_bench.py --r21d --what fwd,dgrad --layers c2.sp,c2.tm,c3b.sp,c3b.tm 2>/dev/null | grep "^c" > "$OUT/r21d_p${persist}_q${prio}.txt"
  echo "== persist=$persist prio=$prio"; awk '{print $3, $7, $8, $9, $10}' "$OUT/ks_p${persist}_q${prio}.txt" | tr '\n' ';'; echo; cut -c1-75 "$OUT/c3d_p${persist}_q${prio}.txt"; cut -c1-110 "$OUT/r21d_p${persist}_q${prio}.txt"
done; done
