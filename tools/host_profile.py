"""GPU box: where the HOST spends an eagerly issued step (cProfile over N steps, top functions by own time).
    python3 tools/host_profile.py --arch s3dg --steps 10"""
import argparse
import cProfile
import os
import pstats
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="s3dg")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--top", type=int, default=45)
a = ap.parse_args()
from model_util import make_cfg  # noqa: E402
from rspnet_amd.moco import Loss, ModelFactory  # noqa: E402
from rspnet_amd.optim import SGD  # noqa: E402

dev = torch.device("cuda:0")
B, hw = (16, 224) if a.arch == "s3dg" else (32, 112)
wrapped = ModelFactory(make_cfg(a.arch, 16384)).build_moco_diffloss(device=dev)
wrapped.train()
opt = SGD(wrapped.parameters(), lr=0.01, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
crit = Loss(margin=1.0, A=1.0, M=1.0)
im_q, im_k = (torch.randn(B, 3, 32, hw, hw, device=dev) for _ in range(2))


def step():
    out, tgt, rl, rt = wrapped(im_q, im_k)
    loss, _, _ = crit(out, tgt, rl, rt)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(3):
    step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(a.steps):
    step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats("tottime")
tot = sum(v[2] for v in st.stats.values())
print(f"{a.arch}: {tot / a.steps * 1e3:.1f} ms of host time per step under the profiler")
rows = sorted(st.stats.items(), key=lambda kv: -kv[1][2])[:a.top]
for (fn, line, name), (cc, nc, tt, ct, _) in rows:
    print(f"{tt / a.steps * 1e3:8.2f} ms  {nc / a.steps:8.1f} calls  {os.path.basename(fn)}:{line} {name}")
