#!/usr/bin/env python
"""How much do the four independent branches of an S3D-G inception block gain from running on separate HIP streams?
Times the forward convolutions (+ BN statistics finalize + BN-apply) of one block's branches back to back on one stream and
spread over four streams, for a 28x28, a 14x14 and a 7x7 block at B=16 (the late blocks launch far fewer workgroups than the
256 CUs x 2 slots)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom, PoolGeom

be = ops.backend()
dev = torch.device("cuda", 0)
B = 16
# (name, T, HW, cin, [o0..o5])
BLOCKS = [("3c", 8, 28, 256, [128, 128, 192, 32, 96, 64]), ("4c", 4, 14, 512, [160, 112, 224, 24, 64, 64]),
          ("4f", 4, 14, 528, [256, 160, 320, 32, 128, 128]), ("5c", 2, 7, 832, [384, 192, 384, 48, 128, 128])]


def unit(x, cin, cout, k, p):
    N, T, H, W, _ = x.shape
    g = ConvGeom(N, T, H, W, cin, cout, k, (1, 1, 1), p)
    w = torch.randn(cout, cin, *k, device=dev) * 0.05
    wp = be.conv_pack_fwd(g, w)
    gamma, beta = torch.ones(cout, device=dev), torch.zeros(cout, device=dev)
    rm, rv = torch.zeros(cout, device=dev), torch.ones(cout, device=dev)
    pg = PoolGeom(N, T, H, W, cout)

    def run(xin):
        y, st = be.conv_fwd(g, xin, wp, None, True)
        mi, ss = be.bn_finalize(st, g.rows, None, gamma, beta, 1e-3, 1e-3, rm, rv)
        return be.bn_act_pool_fwd(pg, y, ss, None, True)
    return run


for name, T, HW, cin, o in BLOCKS:
    x = torch.randn(B, T, HW, HW, cin, device=dev)
    k1, p0 = (1, 1, 1), (0, 0, 0)
    br = [[unit(x, cin, o[0], k1, p0)],
          [unit(x, cin, o[1], k1, p0), unit(x, o[1], o[2], (1, 3, 3), (0, 1, 1)), unit(x, o[2], o[2], (3, 1, 1), (1, 0, 0))],
          [unit(x, cin, o[3], k1, p0), unit(x, o[3], o[4], (1, 3, 3), (0, 1, 1)), unit(x, o[4], o[4], (3, 1, 1), (1, 0, 0))],
          [unit(x, cin, o[5], k1, p0)]]
    streams = [torch.cuda.Stream(dev) for _ in range(4)]

    def seq():
        for b in br:
            h = x
            for u in b:
                h = u(h)

    def par():
        main = torch.cuda.current_stream(dev)
        ev = torch.cuda.Event()
        ev.record(main)
        for b, s in zip(br, streams):
            s.wait_event(ev)
            with torch.cuda.stream(s):
                h = x
                for u in b:
                    h = u(h)
        for s in streams:
            main.wait_stream(s)

    def timeit(fn, it=20):
        fn(); fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(it):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / it * 1e3

    a, b_ = timeit(seq), timeit(par)
    # the same two schedules as captured HIP graphs: no host cost per launch, so what is measured is the GPU's own time — eager
    # launches of this block cost the host ~15-20 us each, about as much as the small kernels run
    side = torch.cuda.Stream(dev)
    graphs = []
    for fn in (seq, par):
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            fn(); fn()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            fn()
        graphs.append(g)
    ga, gb = timeit(graphs[0].replay), timeit(graphs[1].replay)
    print(f"sepInc_{name} ({T}x{HW}x{HW}, cin {cin}): eager one stream {a:7.1f} us, four streams {b_:7.1f} us (x{a / b_:.2f}) | "
          f"HIP graph one stream {ga:7.1f} us, four streams {gb:7.1f} us (x{ga / gb:.2f})", flush=True)
