#!/bin/bash
# Run ON THE GPU BOX: PMC FETCH_SIZE (x2 per MI355X_MICROARCH.md) and WRITE_SIZE per conv launch of a few layers, keyed by kernel
# and grid size.   bash tools/fetch_probe.sh [layers] [what] [extra conv_bench flags...]     (what: fwd,dgrad | wgrad | ...)
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
LAYERS=${1:-conv2,conv3b,conv4b}; WHAT=${2:-fwd,dgrad}; shift; shift
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/fetch_probe_$C
  rocprofv3 --pmc $C --output-format csv -d $R/gpurun_out/fetch_probe_$C -- python3 $R/tools/conv_bench.py --what $WHAT --layers $LAYERS --iters 2 "$@" > /dev/null 2>&1
done
python3 - "$R/gpurun_out" <<'PY'
import sys, glob, csv, collections
acc = collections.defaultdict(lambda: [0, 0.0, 0.0])
for ci, c in enumerate(("FETCH_SIZE", "WRITE_SIZE")):
    for fn in glob.glob(sys.argv[1] + f"/fetch_probe_{c}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            kn = r["Kernel_Name"]
            if r.get("Counter_Name") == c and any(s in kn for s in ("igemm_", "wgrad_dma", "wgrad_kernel", "stem_")):
                name = kn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                k = (name, r.get("Grid_Size"))
                if ci == 0:
                    acc[k][0] += 1
                acc[k][1 + ci] += float(r["Counter_Value"])
for k, (n, f, w) in sorted(acc.items()):
    print("   %-40s grid %9s: %d launches, fetch x2 %8.0f MB/launch, write %8.0f MB/launch" % (k[0], k[1], n, f * 2 * 1024 / n / 1e6, w * 1024 / n / 1e6))
PY
