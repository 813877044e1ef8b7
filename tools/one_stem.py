#!/usr/bin/env python
"""Run the C3D stem conv forward a few times (for rocprofv3 --pmc runs)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd import ops
from rspnet_amd.ops import ConvGeom
be = ops.backend(); dev = torch.device("cuda", 0)
g = ConvGeom(32, 16, 112, 112, 4, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1))
x = torch.randn(32, 16, 112, 112, 4, device=dev); w = torch.randn(64, 4, 3, 3, 3, device=dev) * 0.1
wp = be.conv_pack_fwd(g, w)
for _ in range(6):
    be.conv_fwd(g, x, wp, None, True)
torch.cuda.synchronize()
