#!/usr/bin/env python3
"""tools/reg_report.py file.hip [filter]: VGPRs / SGPRs / scratch / occupancy per kernel of one translation unit (gfx950)."""
import os
import re
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "..", "rspnet_amd", "csrc")
out = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result",
                      "-I" + os.path.join(HERE, "..", "include"), "-I" + SRC, "-c", os.path.join(SRC, sys.argv[1]), "-o", "/dev/null",
                      "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:], capture_output=True, text=True).stderr
rows, cur = [], {}
for line in out.splitlines():
    m = re.search(r"remark: +(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|VGPRs Spill): (\S+)", line)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == "Function Name":
        cur = {"name": v}
        rows.append(cur)
    else:
        cur["Spill" if k == "VGPRs Spill" else k.split(" ")[0]] = v
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for r in rows:
    name = subprocess.run(["c++filt", r["name"]], capture_output=True, text=True).stdout.strip()
    name = name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    if flt in name:
        print("%-55s VGPR %4s AGPR %3s SGPR %4s scratch %3s spill %3s occ %s" % (name, r.get("VGPRs", "?"), r.get("AGPRs", "?"), r.get("TotalSGPRs", "?"),
                                                                                r.get("ScratchSize", "?"), r.get("Spill", "?"), r.get("Occupancy", "?")))
