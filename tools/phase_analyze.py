#!/usr/bin/env python
"""Offline analysis of tools/phase_probe.py dumps (per-wave s_memtime stamps; every CU has its own counter).
Usage: python tools/phase_analyze.py gpurun_out/<tag>/raw [name ...]"""
import glob
import os
import sys

import numpy as np


def analyze(path, verbose=False):
    d = np.load(path)
    meta = open(path[:-4] + ".txt").read().split()
    ms, flops, rows, K, cout = float(meta[0]), float(meta[1]), int(meta[2]), int(meta[3]), int(meta[4])
    mfma_per_chunk = float(meta[5]) if len(meta) > 5 else None
    t = d[:, :5].astype(np.int64)
    hw = d[:, 5]
    xcc = ((hw >> np.uint64(32)) & np.uint64(15)).astype(np.int64)
    se = ((hw >> np.uint64(13)) & np.uint64(7)).astype(np.int64)
    cu = ((hw >> np.uint64(8)) & np.uint64(15)).astype(np.int64)
    simd = ((hw >> np.uint64(4)) & np.uint64(3)).astype(np.int64)
    chunks = (d[:, 7] >> np.uint64(32)).astype(np.int64)
    cukey = (xcc * 8 + se) * 16 + cu
    spans = []
    for k in np.unique(cukey):
        m = cukey == k
        t[m] -= t[m, 0].min()
        spans.append(t[m, 4].max())
    span = float(np.median(spans))
    ghz = span / (ms * 1e6)
    us = lambda c: c / (ghz * 1e3)
    ph = np.diff(t, axis=1)
    full = chunks == chunks.max()
    key = cukey * 4 + simd
    cov, occ, res, inpro, lock = [], [], [], [], []
    for kk in np.unique(key):
        m = key == kk
        sp = t[cukey == (kk // 4), 4].max()

        def area(a, b):
            ev = sorted([(x, 1) for x in a] + [(x, -1) for x in b])
            cur, last, busy, ar, hist = 0, 0, 0, 0, {}
            for tt, dl in ev:
                if cur > 0:
                    busy += tt - last
                ar += cur * (tt - last)
                hist[cur] = hist.get(cur, 0) + tt - last
                cur += dl
                last = tt
            return busy / sp, ar / sp, hist
        b, a, _ = area(t[m, 1], t[m, 2])
        cov.append(b)
        occ.append(a)
        _, a2, h = area(t[m, 0], t[m, 4])
        res.append(a2)
    cyc_chunk = ph[full, 1].mean() / chunks.max()
    print(f"{os.path.basename(path)[:-4]:16s} K={K:5d} rows={rows:8d} N={cout:4d} {ms * 1e3:8.1f} us {flops / ms / 1e9:6.1f} TF | counter {ghz:.2f} GHz | per wave (us): "
          f"pro {us(ph[full, 0].mean()):5.2f} loop {us(ph[full, 1].mean()):6.2f} ({chunks.max()} ch, {cyc_chunk:6.0f} cyc/ch) st {us(ph[full, 2].mean()):5.2f} "
          f"stat {us(ph[full, 3].mean()):5.2f} life {us((t[full, 4] - t[full, 0]).mean()):6.2f} | SIMD: >=1 in loop {np.mean(cov):.3f}, waves in loop {np.mean(occ):.2f}, "
          f"resident {np.mean(res):.2f} | loop share of life {ph[full, 1].mean() / (t[full, 4] - t[full, 0]).mean():.2f}")
    if verbose:
        k = np.unique(cukey)[3]
        m = cukey == k
        o = np.argsort(t[m, 0])
        print(np.c_[t[m][o], simd[m][o], (hw[m][o] & np.uint64(15)).astype(int), chunks[m][o]][:48])


if __name__ == "__main__":
    root = sys.argv[1]
    names = sys.argv[2:]
    for f in sorted(glob.glob(os.path.join(root, "*.npy"))):
        if names and not any(n in f for n in names):
            continue
        analyze(f, verbose=bool(os.environ.get("VERBOSE")))
