#!/usr/bin/env python
"""Turn the raw rocprofv3 outputs under gpurun_out/ into the small committed summaries under profiles/<round>/ and
profiles/traffic.json (HBM bytes per launch of the dominant kernel, read by bench.py for roofline.traffic).
FETCH_SIZE is doubled as MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on gfx950; values are KiB."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]          # e.g. r01 r1b
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")
shutil.copy(glob.glob(f"{g}/prof_{tag}/*/*_kernel_stats.csv")[0], f"{out}/kernel_stats_c3d_b32_{tag}.csv")


def pmc(kind):
    f = glob.glob(f"{g}/pmc_{kind}_{tag}/*/*_counter_collection.csv")[0]
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    with open(f"{out}/pmc_{kind}_size_by_kernel_{tag}.txt", "w") as o:
        o.write(f"# rocprofv3 --pmc {kind.upper()}_SIZE -- python3 bench.py --steps 2 --warmup 1 (3 steps); raw counter sum (KiB)\n")
        for k, (n, v) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
            o.write(f"{k:60s} calls={n:5d} sum_KiB={v:.0f} avg_MiB_per_launch={v / n / 1024:.1f}\n")
    return agg


fetch, write = pmc("fetch"), pmc("write")
dom = [k for k in fetch if k.startswith("void igemm_kernel<128, 128")][0]
n = fetch[dom][0]
traffic = {"kernel": dom, "launches_profiled": n,
           "fetch_bytes_per_launch": fetch[dom][1] * 1024 * 2 / n, "write_bytes_per_launch": write[dom][1] * 1024 / write[dom][0],
           "note": "FETCH_SIZE x2 (gfx950 half-count for wide coalesced reads) + WRITE_SIZE, averaged over the fwd+dgrad "
                   "launches of 3 C3D B=32 steps", "source": f"profiles/{rnd}/pmc_*_{tag}.txt"}
traffic["hbm_bytes_per_launch"] = traffic["fetch_bytes_per_launch"] + traffic["write_bytes_per_launch"]
json.dump(traffic, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
print(json.dumps(traffic, indent=1))
rows = list(csv.DictReader(open(f"{out}/kernel_stats_c3d_b32_{tag}.csv")))
for r in rows[:8]:
    print(r["Name"].replace("(anonymous namespace)::", "")[:70].ljust(70), r["Calls"], f'{float(r["AverageNs"]) / 1e6:.3f} ms', r["Percentage"])
