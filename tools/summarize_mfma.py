#!/usr/bin/env python
"""profiles/<round>/pmc_mfma_util_<tag>.txt from gpurun_out/pmc_mfma_<tag>: per kernel, matrix-pipe busy share =
SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs).  SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024
SIMDs and counts cycles (64 per v_mfma_f32_32x32x2_f32, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is summed over the 8 XCDs."""
import collections
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{ROOT}/gpurun_out/pmc_mfma_{tag}/*/*_counter_collection.csv")[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        calls[k] += 1
rows = []
for k, c in agg.items():
    act = c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    if act > 0:
        rows.append((c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (act * 1024.0), act, k))
rows.sort(key=lambda r: -r[1])
out = os.path.join(ROOT, "profiles", rnd, f"pmc_mfma_util_{tag}.txt")
with open(out, "w") as o:
    o.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE -- python3 bench.py --steps 2 --warmup 1 (3 steps, C3D B=32)\n")
    o.write("# mfma_busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 * 1024 SIMDs); shader_Mcycles = GRBM_GUI_ACTIVE/8 summed over launches\n")
    for util, act, k in rows[:12]:
        o.write(f"{k:58s} launches={calls[k]:4d} shader_Mcycles={act / 1e6:9.1f} mfma_busy={util:.3f}\n")
print(open(out).read())
