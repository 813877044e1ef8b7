#!/bin/bash
# Run ON THE GPU BOX: one default-length bench line per BASELINE backbone -> gpurun_out/bench_<tag>_<arch>.json
TAG="$1"; shift
ARCHS=("$@"); [ ${#ARCHS[@]} -eq 0 ] && ARCHS=(c3d resnet18 r2plus1d-vcop s3dg)
R="${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}"
mkdir -p "$R/gpurun_out"
for a in "${ARCHS[@]}"; do
  python3 "$R/bench.py" --arch "$a" --no-cpu-baseline > "$R/gpurun_out/bench_${TAG}_$a.json" 2> "$R/gpurun_out/bench_${TAG}_$a.err"
  python3 - "$R/gpurun_out/bench_${TAG}_$a.json" <<'PY'
import json, sys
try:
    d = json.load(open(sys.argv[1])); r = d["roofline"]
    print(d["metric"], d["value"], "clips/s", d["ms_per_step"], "ms | dom", r["kernel"], r["achieved"], "TF frac", r["frac"], "| conv ms", r["all_conv_launches"]["ms_per_step"])
    for k, v in list(r["per_kernel"].items())[:6]: print("    ", k, v)
except Exception as e:
    print("bench failed:", e); print(open(sys.argv[1].replace(".json", ".err")).read()[-1500:])
PY
done
