#!/bin/bash
# Run ON THE GPU BOX: PMC FETCH_SIZE per igemm launch of a few C3D layers (x2 per MI355X_MICROARCH.md), keyed by grid size.
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
rm -rf $R/gpurun_out/fetch_probe
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/fetch_probe -- python3 $R/tools/conv_bench.py --what fwd,dgrad --layers ${1:-conv2,conv3b,conv4b} --iters 2 > /dev/null 2>&1
python3 - "$R/gpurun_out/fetch_probe" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for fn in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        if r.get("Counter_Name") == "FETCH_SIZE" and "igemm_" in r["Kernel_Name"]:
            name = "igemm_" + r["Kernel_Name"].split("igemm_")[1].split(">")[0] + ">"
            k = (name, r.get("Grid_Size"))
            acc[k][0] += 1; acc[k][1] += float(r["Counter_Value"])
for k, (n, v) in sorted(acc.items()):
    print("   %s grid %s: %d launches, fetch x2 %.0f MB/launch" % (k[0], k[1], n, v * 2 * 1024 / n / 1e6))
PY
