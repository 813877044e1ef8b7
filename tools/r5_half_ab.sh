#!/bin/bash
# GPU box: the 144-wide tile (16-column half block), tests then A/B (RSP_NO_HALF_BLOCK=1: the 160-wide instance).
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
OUT="$R/gpurun_out"; mkdir -p "$OUT"
cd "$R"
python -m pytest tests/test_kernels_gpu.py -x -q -k "conv" 2>&1 | tail -5
for a in "$@"; do
  for f in 1 0 1 0; do
    if [ $f = 1 ]; then export RSP_NO_HALF_BLOCK=1; else unset RSP_NO_HALF_BLOCK; fi
    python bench.py --arch $a --no-cpu-baseline --no-other-workloads > "$OUT/half_${a}_off$f.json" 2> "$OUT/half_${a}_off$f.err"
    python - "$OUT/half_${a}_off$f.json" <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
pk = d["roofline"].get("per_kernel", {})
k = {n: (v["ms_per_step"], v["tflops"]) for n, v in pk.items() if "160" in n or "144" in n}
print(sys.argv[1].split("/")[-1], d["value"], d.get("step_issue_mode"), d["parity"]["ok"] if "parity" in d else None, k)
PY
  done
done
