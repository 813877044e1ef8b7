#!/usr/bin/env python
"""Per-layer L2-miss traffic of the C3D B=32 step from the per-dispatch PMC rows of tools/profile_round.sh
(gpurun_out/pmc_{fetch,write}_<tag>_c3d): FETCH_SIZE x2 (gfx950 half-count, MI355X_MICROARCH.md) and WRITE_SIZE of every conv
launch of the last profiled step, beside the launch's algorithmic bytes (input + weights read once, output written once).
FETCH_SIZE counts the requests that leave the XCD's L2 — Infinity-Cache (256 MiB) hits included — so it bounds HBM reads from
above; weights (<= 28 MB per layer) and the smaller activations are served from the Infinity Cache.

    python tools/fetch_per_layer.py r2t > profiles/r02/fetch_per_layer_c3d_r2t.txt"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
B = 32
# (name, T, HW, Cin, Cout): C3D conv stack at the conv's own resolution
L = [("conv1", 16, 112, 4, 64), ("conv2", 16, 56, 64, 128), ("conv3a", 8, 28, 128, 256), ("conv3b", 8, 28, 256, 256),
     ("conv4a", 4, 14, 256, 512), ("conv4b", 4, 14, 512, 512), ("conv5a", 2, 7, 512, 512), ("conv5b", 2, 7, 512, 512)]


def alg(l, what):
    name, T, HW, ci, co = l
    rows = B * T * HW * HW
    w = ci * co * 27 * 4
    if what == "fwd":
        return rows * ci * 4 + w, rows * co * 4
    if what == "dgrad":
        return rows * co * 4 + w, rows * ci * 4
    return rows * (ci + co) * 4, w


def conv_rows(kind):
    f = glob.glob(f"{ROOT}/gpurun_out/pmc_{kind}_{tag}_c3d/*/*_counter_collection.csv")[0]
    return [r for r in csv.DictReader(open(f)) if any(s in r["Kernel_Name"] for s in ("igemm_kernel", "igemm_ks_kernel", "igemm_multi_kernel", "stem_resident", "stem_kernel", "wgrad_dma"))]


fr, wr = conv_rows("fetch"), conv_rows("write")
assert len(fr) == len(wr)
# a step = 3 x (stem + 7 igemm) forward, then wgrad/dgrad pairs from conv5b down to conv2, then wgrad conv1: find the last
# dispatch that is wgrad_dma<64,128> (conv1's wgrad) and take the 39 conv dispatches ending there
end = max(i for i, r in enumerate(fr) if "wgrad_dma_kernel<64, 128" in r["Kernel_Name"]) + 1
fr, wr = fr[end - 39:end], wr[end - 39:end]
plan = []
for p in ("key pass 1", "key pass 2", "query fwd"):
    plan += [(L[0], "fwd", p)] + [(l, "fwd", p) for l in L[1:]]
for l in reversed(L[1:]):
    plan += [(l, "wgrad", "backward"), (l, "dgrad", "backward")]
plan += [(L[0], "wgrad", "backward")]
print(f"# C3D B=32, last profiled step of profiles tag {tag}: per conv launch, L2-miss traffic (PMC) vs algorithmic bytes")
print(f"# {'pass':11s} {'layer':7s} {'op':6s} {'kernel':34s} {'fetch MB':>9s} {'alg read':>9s} {'ratio':>6s} {'write MB':>9s} {'alg write':>9s} {'ratio':>6s}")
for (l, what, p), f, w in zip(plan, fr, wr):
    k = f["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")
    ar, aw = alg(l, what)
    fb, wb = float(f["Counter_Value"]) * 2 * 1024, float(w["Counter_Value"]) * 1024
    print(f"  {p:11s} {l[0]:7s} {what:6s} {k:34s} {fb / 1e6:9.1f} {ar / 1e6:9.1f} {fb / ar:6.2f} {wb / 1e6:9.1f} {aw / 1e6:9.1f} {wb / aw:6.2f}")
