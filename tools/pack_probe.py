#!/usr/bin/env python
"""Times the weight re-pack paths on one C3D conv5-sized weight: single-conv rsp_conv3d_pack_fwd vs the batched PackSet
(forward layout / dgrad layout / both in one launch)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom
be = ops.backend()
dev = torch.device("cuda", 0)

def timeit(fn, it=20):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it * 1e3

for cin, cout in ((512, 512), (64, 128), (256, 256)):
    g = ConvGeom(32, 2, 7, 7, cin, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    w = torch.randn(cout, cin, 3, 3, 3, device=dev)
    n = w.numel()
    t_single = timeit(lambda: be.conv_pack_fwd(g, w))
    pf, pd, pb = be.pack_set([(g, 0, w)]), be.pack_set([(g, 1, w)]), be.pack_set([(g, 0, w), (g, 1, w)])
    print(f"{cout}x{cin}x27 ({n * 4 / 1e6:.1f} MB): single fwd {t_single:7.1f} us | batch fwd {timeit(pf.run):7.1f} us | batch dgrad "
          f"{timeit(pd.run):7.1f} us | batch both {timeit(pb.run):7.1f} us")
