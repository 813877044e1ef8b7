#!/usr/bin/env python
"""Run one C3D conv layer forward a few times (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom
name = sys.argv[1] if len(sys.argv) > 1 else "conv2"
L = {"conv2": (16, 56, 64, 128), "conv3b": (8, 28, 256, 256), "conv4b": (4, 14, 512, 512)}[name]
T, HW, cin, cout = L
be = ops.backend(); dev = torch.device("cuda", 0)
g = ConvGeom(32, T, HW, HW, cin, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
x = torch.randn(32, T, HW, HW, cin, device=dev); w = torch.randn(cout, cin, 3, 3, 3, device=dev) * 0.05
wp = be.conv_pack_fwd(g, w)
for _ in range(6):
    be.conv_fwd(g, x, wp, None, True)
torch.cuda.synchronize()
