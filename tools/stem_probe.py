"""GPU box: time C3D conv1 (the resident stem kernel) alone; with tools/librspnet_hip_stemdbg.so (tools/build_stem_dbg.sh) and
STEM_DBG=<bits> the same launch without its output stores (1), statistics (2), or next-halo copy (4) — what each
part of a patch costs.   python3 tools/stem_probe.py   |   RSPNET_HIP_LIB=tools/librspnet_hip_stemdbg.so STEM_DBG=1 python3 tools/stem_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from rspnet_amd import ops  # noqa: E402
from rspnet_amd.ops import ConvGeom  # noqa: E402

be = ops.backend()
dev = torch.device("cuda:0")
g = ConvGeom(32, 16, 112, 112, 4, 64, (3, 3, 3), (1, 1, 1), (1, 1, 1), Cin_alg=3)
x = torch.randn(32, 16, 112, 112, 4, device=dev)
x[..., 3] = 0
w = torch.randn(64, 3, 3, 3, 3, device=dev) * 0.1      # RGB filters: the three-k-step instance
bias = torch.randn(64, device=dev)
ps = be.pack_set([(g, 0, w)])
ps.run()
wp = ps.packed[0]
for _ in range(5):
    y, st = be.conv_fwd(g, x, wp, bias, True)
torch.cuda.synchronize()
ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
n = 30
ev[0].record()
for _ in range(n):
    y, st = be.conv_fwd(g, x, wp, bias, True)
ev[1].record()
torch.cuda.synchronize()
ms = ev[0].elapsed_time(ev[1]) / n
print(f"STEM_DBG={os.environ.get('STEM_DBG', '0')}  {ms * 1e3:.1f} us  {g.flops / ms / 1e9:.1f} TF algorithmic")
