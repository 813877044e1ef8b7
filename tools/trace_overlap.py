#!/usr/bin/env python
"""How busy is the GPU inside a replayed step?  Reads a rocprofv3 --kernel-trace CSV of `bench.py --arch A --steps N` and reports, for the
last steps: wall time per step, the union of kernel intervals, time with exactly 1 / 2 / >= 3 kernels in flight, and which kernels
run ALONE longest.  CAVEAT (measured, r4t): under rocprofv3 --kernel-trace the dispatches of a replayed graph are SERIALISED — never two
kernels in flight, S3D-G 60.8 ms per step instead of 40 — so this shows the serial kernel sum and the per-dispatch gap (~8 us), not
the overlap of an unprofiled run.  Usage: python tools/trace_overlap.py <kernel_trace.csv> [steps]"""
import collections
import csv
import sys

path = sys.argv[1]
nsteps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
rows = []
for r in csv.DictReader(open(path)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0].replace("void ", "")))
rows.sort()
# steps are delimited by the sgd kernel (one per step)
ends = [e for s, e, n in rows if n.startswith("sgd_kernel")]
ends = ends[-(nsteps + 1):]
t0, t1 = ends[0], ends[-1]
sel = [(s, e, n) for s, e, n in rows if s >= t0 and e <= t1]
ev = sorted([(s, 1, n) for s, e, n in sel] + [(e, -1, n) for s, e, n in sel])
cur, last = 0, t0
hist = collections.Counter()
alone = collections.Counter()
active = collections.Counter()
for t, d, n in ev:
    hist[min(cur, 3)] += t - last
    if cur == 1:
        alone[next(iter(k for k, v in active.items() if v > 0))] += t - last
    cur += d
    active[n] += d
    last = t
hist[0] += t1 - last
wall = (t1 - t0) / nsteps / 1e6
print(f"steps {nsteps}: wall {wall:.2f} ms/step; kernels {len(sel) / nsteps:.0f}/step, kernel time sum {sum(e - s for s, e, n in sel) / nsteps / 1e6:.2f} ms/step")
for k in (0, 1, 2, 3):
    print(f"  {'idle' if k == 0 else str(k) + (' or more' if k == 3 else '') + ' kernel(s) in flight'}: {hist[k] / nsteps / 1e6:6.2f} ms/step ({hist[k] / (t1 - t0) * 100:4.1f} %)")
print("  running alone, by kernel (ms/step):")
for n, v in alone.most_common(14):
    print(f"    {n[:60]:60s} {v / nsteps / 1e6:6.2f}")
