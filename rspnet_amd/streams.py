"""HIP streams that really run side by side.

The HIP runtime multiplexes a process's streams onto a handful of HARDWARE queues (GPU_MAX_HW_QUEUES, 4 by default): two streams that
land on the same queue execute strictly one after the other, whatever events say.  Round 6 found the "w" lane of the replayed step
(rspnet_amd/graph_step.py) on the main lane's queue — the weight-gradient graph it was meant to run BESIDE the next backward piece ran
between the pieces instead (gap +7 us where -1.4 ms was expected; profiles/r06/experiments_r6.txt r6h-j), and a run with six queues put
two of the three forward lanes on one queue (S3D-G 420 -> 280 clips/s).  Which queue a stream gets depends on how many streams the
process has created before it, so it cannot be planned; it is MEASURED here: `distinct(...)` hands out streams that were observed to
overlap with the caller's stream and with one another (two ~0.4 ms spin kernels, one per stream, finish in the time of one)."""
from __future__ import annotations

import time
from typing import Dict, List, Tuple

import torch

_calibrated: Dict[int, int] = {}          # device index -> spin cycles for ~0.4 ms
_pairs: Dict[Tuple[int, int], bool] = {}  # (raw stream handle, raw stream handle) -> overlap observed
_pool: Dict[int, List["torch.cuda.Stream"]] = {}


def _spin_cycles(dev) -> int:
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    if idx not in _calibrated:
        cycles = 200_000
        for _ in range(8):                 # double until one spin takes >= 0.3 ms (the counter's rate differs between parts)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            torch.cuda._sleep(cycles)
            torch.cuda.synchronize(dev)
            if time.perf_counter() - t0 >= 3e-4:
                break
            cycles *= 2
        _calibrated[idx] = cycles
    return _calibrated[idx]


def _timed(streams, dev, cycles) -> float:
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for s in streams:
        with torch.cuda.stream(s):
            torch.cuda._sleep(cycles)
    for s in streams:
        s.synchronize()
    return time.perf_counter() - t0


def overlap(a: "torch.cuda.Stream", b: "torch.cuda.Stream", dev) -> bool:
    """Do kernels on `a` and `b` run concurrently?  (Measured once per pair: the queue of a stream does not change.)"""
    key = (min(a.cuda_stream, b.cuda_stream), max(a.cuda_stream, b.cuda_stream))
    if key not in _pairs:
        cycles = _spin_cycles(dev)
        one = min(_timed([a], dev, cycles), _timed([b], dev, cycles))
        both = min(_timed([a, b], dev, cycles), _timed([b, a], dev, cycles))
        _pairs[key] = both < 1.5 * one
    return _pairs[key]


def distinct(dev, n: int, beside: "torch.cuda.Stream" = None, max_candidates: int = 24) -> List["torch.cuda.Stream"]:
    """`n` streams on `dev` that overlap with `beside` (default: the current stream) and with one another.  Falls back to fresh streams
    for whatever cannot be found (fewer hardware queues than lanes): the step is then correct and partly serialised, as before.
    Must not be called inside a stream capture."""
    beside = beside if beside is not None else torch.cuda.current_stream(dev)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    pool = _pool.setdefault(idx, [])
    chosen: List["torch.cuda.Stream"] = []
    tried = 0
    while len(chosen) < n and tried < max_candidates:
        if tried >= len(pool):
            pool.append(torch.cuda.Stream(device=dev))
        c = pool[tried]
        tried += 1
        if c.cuda_stream == beside.cuda_stream or any(c.cuda_stream == s.cuda_stream for s in chosen):
            continue
        if overlap(c, beside, dev) and all(overlap(c, s, dev) for s in chosen):
            chosen.append(c)
    while len(chosen) < n:
        chosen.append(torch.cuda.Stream(device=dev))
    return chosen


_lanes: Dict[int, Dict[str, "torch.cuda.Stream"]] = {}


def lane(dev, name: str) -> "torch.cuda.Stream":
    """The process's three side lanes "q", "k", "w" of `dev` (query pass / second key pass / weight gradients and gradient buckets): the
    same three streams for the eagerly issued step's side streams and for the replayed graphs' lanes, chosen once — on first use,
    beside the stream that is current then — so that main + q + k + w sit on four different hardware queues."""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    got = _lanes.get(idx)
    if got is None:
        if torch.cuda.is_current_stream_capturing() or not hasattr(torch.cuda, "_sleep"):
            return torch.cuda.Stream(device=dev)          # (nothing is measured inside a capture; the next eager call decides)
        q, k, w = distinct(dev, 3)
        got = _lanes[idx] = {"q": q, "k": k, "w": w}
    return got[name]


def lanes_overlap(dev) -> Dict[str, bool]:
    """For reports: does each lane overlap with the current stream? ({} before the lanes exist)"""
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    got = _lanes.get(idx)
    if not got or torch.cuda.is_current_stream_capturing():
        return {}
    cur = torch.cuda.current_stream(dev)
    return {n: (s.cuda_stream != cur.cuda_stream and overlap(s, cur, dev)) for n, s in got.items()}
