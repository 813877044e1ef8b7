#!/usr/bin/env python
"""Which host call sites issue device-to-device memcpys during one pretext step?  (torch.profiler with stacks; eager step.)"""
import collections
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from torch.profiler import ProfilerActivity, profile

arch = sys.argv[1] if len(sys.argv) > 1 else "s3dg"
from model_util import make_cfg
from rspnet_amd.moco import ModelFactory
from rspnet_amd.moco.builder_diffspeed_diffloss import Loss

dev = torch.device("cuda", 0)
hw, B = (224, 16) if arch == "s3dg" else (112, 32)
cfg = make_cfg(arch, 16384)
model = ModelFactory(cfg).build_moco_diffloss(device=dev)
crit = Loss(margin=2.0, A=1.0, M=1.0)
opt = torch.optim.SGD([p for p in model.parameters() if p.requires_grad], lr=0.05, momentum=0.9, weight_decay=1e-4)
im_q = torch.randn(B, 3, 32, hw, hw, device=dev)
im_k = torch.randn(B, 3, 32, hw, hw, device=dev)


def step():
    out, tgt, rl, rt = model(im_q, im_k)
    loss, _, _ = crit(out, tgt, rl, rt)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    step()
    torch.cuda.synchronize()
cnt = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::copy_", "aten::clone", "aten::contiguous", "aten::cat", "aten::stack", "aten::index_select", "aten::to"):
        st = [s for s in (ev.stack or []) if "rspnet_amd" in s or "torch/optim" in s]
        cnt[(ev.name, st[0] if st else "?")] += 1
for (name, where), n in cnt.most_common(25):
    print(f"{n:5d}  {name:18s} {where}")
names = collections.Counter(ev.name for ev in prof.events() if "emcpy" in ev.name or "copyBuffer" in ev.name)
print(names)
