#!/bin/bash
# tools/bench4.sh OUTDIR [reps]: bench.py per backbone (no CPU leg, no other workloads), prints clips/s and ms/step.
OUT="gpurun_out/$1"; REPS="${2:-2}"; mkdir -p "$OUT"
for a in c3d resnet18 r2plus1d-vcop s3dg; do
  for rep in $(seq 1 $REPS); do
    python bench.py --arch $a --no-cpu-baseline --no-other-workloads > "$OUT/${a}_$rep.json" 2> "$OUT/${a}_$rep.err"
  done
done
python - "$OUT" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["roofline"]["whole_step"]["frac"] if "whole_step" in d.get("roofline", {}) else "")
    except Exception as e:
        print(f, "ERR", e)
PY
