#!/usr/bin/env python
"""Turn the raw rocprofv3 outputs of tools/profile_round.sh (gpurun_out/{prof,pmc_fetch,pmc_write,pmc_mfma}_<tag>_<arch>) into
the small committed summaries under profiles/<round>/ and into profiles/traffic.json (HBM bytes per launch of every conv kernel
of every backbone, read by bench.py for roofline.traffic).

    python tools/summarize_profiles.py r02 r2a

FETCH_SIZE is doubled as MI355X_MICROARCH.md §HBM prescribes for wide coalesced reads on gfx950; both counters are in KiB.
Matrix-pipe busy share = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd, tag = sys.argv[1], sys.argv[2]
out = os.path.join(ROOT, "profiles", rnd)
os.makedirs(out, exist_ok=True)
g = os.path.join(ROOT, "gpurun_out")
BATCH = {"c3d": 32, "resnet18": 32, "r2plus1d-vcop": 32, "s3dg": 16}
PROFILED_STEPS = 11         # --steps 2 --warmup 1 in the PMC passes, + bench.py's 5 idle-queue issue samples + its roofline pass (1 + 2 steps)
TRACED_STEPS = 29           # --steps 10 --warmup 3 in the kernel-trace pass, + the 5 issue samples + the roofline pass (1 + 10 steps)


def short(name):
    return name.replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "").strip()


def one(path_glob):
    f = glob.glob(path_glob)
    return f[0] if f else None


def pmc_by_kernel(path):
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    calls = collections.defaultdict(lambda: collections.Counter())
    for r in csv.DictReader(open(path)):
        k = short(r["Kernel_Name"])
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        calls[k][r["Counter_Name"]] += 1
    return agg, calls


tpath = os.path.join(ROOT, "profiles", "traffic.json")
traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}
if "kernel" in traffic:      # round-1 single-kernel format
    traffic = {}

sys.path.insert(0, ROOT)
from rspnet_amd import _lib  # noqa: E402

hpath = f"{g}/csrc_hash_{tag}.txt"
# the hash written on the GPU box by tools/profile_round.sh = the sources the profiled build was compiled from
build = {"csrc_sha256": open(hpath).read().strip() if os.path.exists(hpath) else _lib.source_hash(),
         "hash_from": "the profiled run (tools/profile_round.sh)" if os.path.exists(hpath) else "the tree at summarise time",
         "commit": os.popen(f"git -C {ROOT} rev-parse HEAD").read().strip() or None, "tag": tag}

for arch, B in BATCH.items():
    ks = one(f"{g}/prof_{tag}_{arch}/*/*_kernel_stats.csv")
    if not ks:
        continue
    shutil.copy(ks, f"{out}/kernel_stats_{arch}_b{B}_{tag}.csv")
    for kind in ("bench", "bench_under_rocprof"):
        src = f"{g}/{kind}_{tag}_{arch}.json"
        if os.path.exists(src) and os.path.getsize(src):
            shutil.copy(src, f"{out}/{kind}_{arch}_b{B}_{tag}.json")
    fetch_f, write_f, mfma_f = (one(f"{g}/pmc_{k}_{tag}_{arch}/*/*_counter_collection.csv") for k in ("fetch", "write", "mfma"))
    lines = [f"# {arch} B={B}: rocprofv3 --pmc <counter> -- python3 bench.py --arch {arch} --steps 2 --warmup 1 ({PROFILED_STEPS} steps), "
             f"one pass per counter group; FETCH x2 per MI355X_MICROARCH.md (gfx950 half-count), KiB -> bytes",
             f"# {'kernel':58s} launches  fetch_MB/launch(x2)  write_MB/launch  hbm_MB/launch  mfma_busy"]
    fetch = pmc_by_kernel(fetch_f) if fetch_f else ({}, {})
    write = pmc_by_kernel(write_f) if write_f else ({}, {})
    mfma = pmc_by_kernel(mfma_f) if mfma_f else ({}, {})
    ent = {}
    names = sorted(fetch[0], key=lambda k: -(fetch[0][k]["FETCH_SIZE"] + write[0].get(k, {}).get("WRITE_SIZE", 0.0)))
    for k in names[:24]:
        n = fetch[1][k]["FETCH_SIZE"]
        fb = fetch[0][k]["FETCH_SIZE"] * 1024 * 2 / max(n, 1)
        wn = write[1].get(k, {}).get("WRITE_SIZE", 0)
        wb = write[0].get(k, {}).get("WRITE_SIZE", 0.0) * 1024 / max(wn, 1)
        busy = ""
        if k in mfma[0]:
            act = mfma[0][k].get("GRBM_GUI_ACTIVE", 0.0) / 8.0
            if act > 0:
                busy = f"{mfma[0][k].get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0) / (act * 1024.0):.3f}"
        lines.append(f"{k:60s} {n:6d}  {fb / 1e6:12.2f}  {wb / 1e6:12.2f}  {(fb + wb) / 1e6:12.2f}  {busy}")
        if any(s in k for s in ("igemm_kernel", "igemm_ks_kernel", "igemm_persist_kernel", "igemm_multi_kernel", "wgrad_dma_kernel", "wgrad_kernel", "stem_")):
            ent[k] = {"launches_profiled": n, "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                      "hbm_bytes_per_launch": fb + wb, "mfma_busy": float(busy) if busy else None,
                      "source": f"profiles/{rnd}/pmc_by_kernel_{arch}_b{B}_{tag}.txt"}
    with open(f"{out}/pmc_by_kernel_{arch}_b{B}_{tag}.txt", "w") as o:
        o.write("\n".join(lines) + "\n")
    ent["_build"] = build
    traffic[f"{arch}_b{B}"] = ent
    rows = list(csv.DictReader(open(ks)))
    tot = sum(float(r["TotalDurationNs"]) for r in rows)
    print(f"== {arch} B={B}: GPU time {tot / 1e6 / TRACED_STEPS:.2f} ms/step over {TRACED_STEPS} profiled steps")
    for r in rows[:10]:
        k = short(r["Name"])
        t = traffic[f"{arch}_b{B}"].get(k, {})
        print(f"  {k[:56]:56s} calls {int(r['Calls']):6d}  avg {float(r['AverageNs']) / 1e6:8.4f} ms  {float(r['Percentage']):5.1f}%  "
              f"hbm {t.get('hbm_bytes_per_launch', 0) / 1e6:8.1f} MB/launch  busy {t.get('mfma_busy')}")

json.dump(traffic, open(tpath, "w"), indent=1)
