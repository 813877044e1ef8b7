#!/usr/bin/env python
"""What does ONE dependent launch cost inside a replayed HIP graph — and what does a FORK inside the graph do to it?
A chain of N kernels (each depends on the previous one: same buffer), replayed
  (a) as one linear graph;
  (b) as ONE graph holding 2 / 3 such chains forked onto side streams inside the capture;
  (c) as 2 / 3 LINEAR graphs, one per chain, replayed side by side on streams of their own (ordered by events, outside any capture).
Time per chain link = launch-to-launch latency of a dependent chain, the quantity that prices S3D-G's BatchNorm chains (reduce ->
finalize -> apply ...).  Also the host time of the replay call(s) while the GPU is busy with the previous replay.
usage: chain_gap_probe.py [elements per kernel: 64 = tiny, 1048576 ~ 6 us]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch

from rspnet_amd import ops

dev = torch.device("cuda", 0)
be = ops.backend()
N = 1000


def chain(buf, n):
    for _ in range(n):
        be.eltwise("relu_fwd", buf, out=buf)


def timed(run, reps=5):
    run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    hs = []
    e0.record()
    for _ in range(reps):
        t0 = time.perf_counter()
        run()
        hs.append((time.perf_counter() - t0) * 1e3)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, sorted(hs)[len(hs) // 2]


for elems in [int(a) for a in sys.argv[1:]] or [64, 1 << 20]:
    print(f"--- {elems} floats per kernel")
    x = torch.zeros(elems, device=dev)
    chain(x, 10)
    torch.cuda.synchronize()
    for lanes in (1, 2, 3):
        bufs = [torch.zeros(elems, device=dev) for _ in range(lanes)]
        side = [torch.cuda.Stream() for _ in range(lanes - 1)]
        # (b) one graph, chains forked inside the capture
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            main = torch.cuda.current_stream()
            for s in side:
                s.wait_stream(main)
            for i, s in enumerate(side):
                with torch.cuda.stream(s):
                    chain(bufs[i + 1], N)
            chain(bufs[0], N)
            for s in side:
                main.wait_stream(s)
        ms, h = timed(g.replay)
        print(f"one graph, {lanes} chain(s) forked inside: {ms / N * 1e3:6.2f} us per link   replay call {h:6.2f} ms of host (GPU busy)")
        if lanes == 1:
            continue
        # (c) one LINEAR graph per chain, replayed on its own stream
        gs = []
        for i in range(lanes):
            gi = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gi):
                chain(bufs[i], N)
            gs.append(gi)

        def lanes_run():
            main = torch.cuda.current_stream()
            for s in side:
                s.wait_stream(main)
            for i, s in enumerate(side):
                with torch.cuda.stream(s):
                    gs[i + 1].replay()
            gs[0].replay()
            for s in side:
                main.wait_stream(s)

        ms, h = timed(lanes_run)
        print(f"{lanes} linear graphs on {lanes} streams:          {ms / N * 1e3:6.2f} us per link   replay calls {h:6.2f} ms of host (GPU busy)")
    # the same chain issued eagerly on one stream
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    chain(x, N)
    e1.record()
    th = time.perf_counter() - t0
    torch.cuda.synchronize()
    print(f"eager, one stream: {e0.elapsed_time(e1) / N * 1e3:.2f} us per link on the GPU, {th / N * 1e6:.2f} us of host per launch")
