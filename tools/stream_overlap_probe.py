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
BLOCKS = [("3b", 8, 28, 192, [64, 96, 128, 16, 32, 32]), ("3c", 8, 28, 256, [128, 128, 192, 32, 96, 64]),
          ("4b", 4, 14, 480, [192, 96, 208, 16, 48, 64]), ("4c", 4, 14, 512, [160, 112, 224, 24, 64, 64]),
          ("4d", 4, 14, 512, [128, 128, 256, 24, 64, 64]), ("4e", 4, 14, 512, [112, 144, 288, 32, 64, 64]),
          ("4f", 4, 14, 528, [256, 160, 320, 32, 128, 128]), ("5b", 2, 7, 832, [256, 160, 320, 32, 128, 128]),
          ("5c", 2, 7, 832, [384, 192, 384, 48, 128, 128])]
TOTAL = {"seq": 0.0, "lanes": 0.0}


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
    # round 6: ... and the way the lanes schedule would run sibling branches — every branch a LINEAR graph of its own, replayed side by
    # side on streams MEASURED to sit on hardware queues of their own (rspnet_amd/streams.py): the best case of "the siblings' launches
    # grouped into one" (all their workgroups on the machine together, no fork bookkeeping inside a graph)
    from rspnet_amd import streams as _st
    lanes = _st.distinct(dev, 3)
    bgraphs = []
    for b in br:
        def one(b=b):
            h = x
            for u in b:
                h = u(h)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            one(); one()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            one()
        bgraphs.append(g)

    def lanes_replay():
        main = torch.cuda.current_stream(dev)
        for s_ in lanes:
            s_.wait_stream(main)
        # the two separable branches and the pooled one on the side lanes, the pointwise branch on the main lane
        for g, s_ in zip((bgraphs[1], bgraphs[2], bgraphs[3]), lanes):
            with torch.cuda.stream(s_):
                g.replay()
        bgraphs[0].replay()
        for s_ in lanes:
            main.wait_stream(s_)
    gl = timeit(lanes_replay)
    per = [timeit(g.replay) for g in bgraphs]
    TOTAL["seq"] += ga
    TOTAL["lanes"] += gl
    print(f"sepInc_{name}: branches alone {' / '.join(f'{t:.0f}' for t in per)} us; one linear graph {ga:7.1f} us, four linear graphs on four "
          f"hardware queues {gl:7.1f} us (x{ga / gl:.2f}; longest branch {max(per):.0f} us)")
    print(f"sepInc_{name} ({T}x{HW}x{HW}, cin {cin}): eager one stream {a:7.1f} us, four streams {b_:7.1f} us (x{a / b_:.2f}) | "
          f"HIP graph one stream {ga:7.1f} us, four streams {gb:7.1f} us (x{ga / gb:.2f})", flush=True)
print(f"all nine blocks, forward units of one pass: one linear graph {TOTAL['seq'] / 1e3:.2f} ms, branches side by side {TOTAL['lanes'] / 1e3:.2f} ms "
      f"(-{(TOTAL['seq'] - TOTAL['lanes']) / 1e3:.2f} ms per pass if NOTHING else ran beside it; in the step the other two passes do)")
