#!/usr/bin/env python
"""GPU box: one small convolution through the persistent kernels vs the per-tile ones (RSP_NO_PERSIST) — forward and input gradient."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 1 and sys.argv[1] == "child":
    import torch
    from rspnet_amd import ops
    from rspnet_amd.ops import ConvGeom
    be = ops.backend(); dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    out = {}
    for name, (N, D, H, W, cin, cout, k, s, p) in {"a": (2, 4, 12, 12, 64, 3, (3, 3, 3), (1, 1, 1), (1, 1, 1)), "b": (2, 4, 12, 12, 32, 64, (1, 3, 3), (1, 1, 1), (0, 1, 1)),
                                                  "c": (4, 8, 28, 28, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)), "d": (2, 4, 12, 12, 64, 32, (3, 3, 3), (2, 2, 2), (1, 1, 1)),
                                                  "f1": (3, 3, 18, 21, 8, 130, (3, 1, 7), (1, 1, 2), (1, 0, 2)), "f2": (2, 3, 18, 10, 20, 96, (3, 3, 1), (2, 2, 1), (1, 1, 0)),
                                                  "f3": (2, 3, 9, 17, 16, 16, (3, 7, 1), (1, 1, 1), (0, 2, 0)), "f4": (16, 4, 19, 11, 128, 32, (3, 1, 3), (1, 1, 1), (1, 0, 0)),
                                                  "f5": (8, 3, 7, 21, 4, 128, (3, 1, 1), (1, 1, 1), (1, 0, 0))}.items():
        g = ConvGeom(N, D, H, W, cin, cout, k, s, p)
        x = torch.randn(N, D, H, W, cin, device=dev); w = torch.randn(cout, cin, *k, device=dev) * 0.05
        torch.cuda.synchronize()
        junk = torch.full((1 << 24,), 7.0, device=dev); del junk      # stale memory is recognisable
        y, st = be.conv_fwd(g, x, be.conv_pack_fwd(g, w), None, True)
        dy = torch.randn(N, *g.out_dims, cout, device=dev)
        junk = torch.full((1 << 24,), 9.0, device=dev); del junk
        dx = be.conv_dgrad(g, dy, w)
        out[name] = (y.cpu(), st.cpu(), dx.cpu(), be.lib.rsp_last_conv_kernel().decode())
    torch.save(out, sys.argv[2])
else:
    import torch
    res = {}
    for mode in ("persist", "classic"):
        env = dict(os.environ)
        if mode == "classic":
            env["RSP_NO_PERSIST"] = "1"
        f = f"/tmp/pd_{mode}.pt"
        subprocess.run([sys.executable, os.path.abspath(__file__), "child", f], env=env, check=True)
        res[mode] = torch.load(f)
    for name in res["persist"]:
        yp, sp, dp, kp = res["persist"][name]; yc, sc, dc, kc = res["classic"][name]
        bad = (yp != yc).reshape(-1, yc.shape[-1])
        rows = bad.any(1).nonzero().view(-1)
        cols = bad.any(0).nonzero().view(-1)
        badd = (dp != dc).reshape(-1, dc.shape[-1])
        print(name, "shape", tuple(yc.shape), "bad fwd rows", rows.numel(), rows[:8].tolist(), rows[-4:].tolist(), "cols", cols[:6].tolist(), cols[-3:].tolist(),
              "| bad dgrad rows", int(badd.any(1).sum()), badd.any(1).nonzero().view(-1)[:8].tolist(), "cols", badd.any(0).nonzero().view(-1)[:6].tolist())
        if rows.numel():
            r0 = int(rows[0])
            print("   fwd row", r0, "persist", yp.reshape(-1, yc.shape[-1])[r0, :6].tolist(), "classic", yc.reshape(-1, yc.shape[-1])[r0, :6].tolist())
        br = badd.any(1).nonzero().view(-1)
        if br.numel():
            r0 = int(br[0])
            print("   dgrad row", r0, "persist", dp.reshape(-1, dc.shape[-1])[r0, :6].tolist(), "classic", dc.reshape(-1, dc.shape[-1])[r0, :6].tolist())
        print(name, kp, "|", kc, "fwd maxdiff", float((yp - yc).abs().max()), "of", float(yc.abs().max()), "stat", float((sp - sc).abs().max()),
              "dgrad", float((dp - dc).abs().max()), "of", float(dc.abs().max()), "nz", float((yp != 0).float().mean()), float((dp != 0).float().mean()))
