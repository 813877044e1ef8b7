#!/bin/bash
# A/B of one environment switch on the GPU box: tools/ab_env.sh OUTDIR VAR "arch[:extra bench flags] ..." — runs bench.py per arch with
# VAR unset and VAR=1 (same box, alternating) and prints clips/s and ms/step.
OUT="gpurun_out/$1"; VAR="$2"; shift 2
mkdir -p "$OUT"
for spec in "$@"; do
  a="${spec%%:*}"; extra=""; [ "$spec" != "$a" ] && extra="${spec#*:}"
  tag="$a$(echo "$extra" | tr -d ' -')"
  for rep in 1 2; do
    python bench.py --arch $a --no-cpu-baseline --no-other-workloads $extra > "$OUT/off_${tag}_$rep.json" 2> "$OUT/off_${tag}_$rep.err"
    env $VAR=1 python bench.py --arch $a --no-cpu-baseline --no-other-workloads $extra > "$OUT/on_${tag}_$rep.json" 2> "$OUT/on_${tag}_$rep.err"
  done
done
python - "$OUT" <<'PY'
import glob, json, sys
for f in sorted(glob.glob(sys.argv[1] + "/*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1])
        print(f.split("/")[-1], d["value"], d["ms_per_step"], d["config"].get("step_issue", "")[:20])
    except Exception as e:
        print(f, "ERR", e)
PY
