#!/bin/bash
# usage: tools/prof_arch.sh <arch>   (run on the GPU box; prints per-kernel time per step)
R=$GRAFT_REPO_ROOT; A=$1
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$A
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$A -- python3 $R/bench.py --arch $A --steps 3 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_$A.json 2>/dev/null
cd $R
python3 - <<PY
import csv,glob,json
d=json.load(open("gpurun_out/prof_$A.json")); print("$A", d["value"], "clips/s", d["ms_per_step"], "ms/step")
rows=list(csv.DictReader(open(glob.glob("gpurun_out/prof_$A/*/*_kernel_stats.csv")[0])))
tot=sum(float(r["TotalDurationNs"]) for r in rows); calls=sum(int(r["Calls"]) for r in rows)
print("  kernel ms per step", round(tot/1e6/5,1), "launches per step", calls/5)
for r in rows[:10]: print("  ", r["Name"].replace("(anonymous namespace)::","")[:56].ljust(56), r["Calls"], round(float(r["TotalDurationNs"])/1e6/5,2), r["Percentage"])
PY
