"""CPU: compile every kernel source for gfx950 (assembly only) and count, per kernel, the `s_waitcnt vmcnt(0)` that sit within five
instructions in front of a global / buffer store.  A load whose first use is inside the branch around a store (a bias, say) makes
hipcc emit such a wait in front of EVERY store of the sequence — and on gfx9 loads and stores share the counter, so each wait also
waits for the previous store's acknowledgement: the stores leave one at a time (found in the stem kernel's epilogue, round 4).
    python tools/isa_store_waits.py [min_count]"""
import glob
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "rspnet_amd", "csrc")
low = int(sys.argv[1]) if len(sys.argv) > 1 else 3
with tempfile.TemporaryDirectory() as tmp:
    procs = []
    for src in sorted(glob.glob(os.path.join(CSRC, "*.hip"))):
        out = os.path.join(tmp, os.path.basename(src)[:-4] + ".s")
        procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                                             "-I" + CSRC, "--cuda-device-only", "-S", src, "-o", out], stderr=subprocess.DEVNULL)))
    for out, pr in procs:
        pr.wait()
        if not os.path.exists(out):
            continue
        lines = open(out).read().split("\n")
        kern, res = None, {}
        for i, l in enumerate(lines):
            m = re.match(r"^(_Z\w+):", l)
            if m:
                kern = m.group(1)
            if kern and "s_waitcnt" in l and "vmcnt(0)" in l:
                k, j = 0, i + 1
                while j < len(lines) and k < 5:
                    t = lines[j].strip()
                    if t and not t.startswith((";", ".")):
                        k += 1
                        if re.match(r"(global_store|buffer_store|flat_store)", t):
                            res[kern] = res.get(kern, 0) + 1
                            break
                    j += 1
        for k, v in sorted(res.items(), key=lambda kv: -kv[1]):
            if v >= low:
                name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "")
                print(f"{os.path.basename(out):16s} {v:4d}  {name[:120]}")
