"""Per-layer table of the convolution launches of one pretext step: for every distinct (pass, geometry) the launches per step,
time per step, achieved TFLOP/s and the kernel instance that ran it — sorted by time.  `python tools/layer_table.py --arch s3dg`."""
import argparse
import os
os.environ.setdefault("RSP_NO_EAGER_OVERLAP", "1")      # per-launch event intervals: the step on ONE stream (no side streams)
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import ARCHS  # noqa: E402
from rspnet_amd import ops  # noqa: E402
from rspnet_amd.moco import Loss, ModelFactory  # noqa: E402
from rspnet_amd.optim import SGD  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="c3d")
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--top", type=int, default=60)
args = ap.parse_args()
dev = torch.device("cuda", 0)
B, hw, base_lr = ARCHS[args.arch]
cfg = {"model": {"arch": args.arch}, "moco": {"dim": 128, "k": 16384, "m": 0.999, "t": 0.07, "fc_type": "linear", "diff_speed": [2]}}
torch.manual_seed(1234)
model = ModelFactory(cfg).build_moco_diffloss(device=dev)
model.train()
crit = Loss(margin=2.0, A=1.0, M=1.0)
opt = SGD(model.parameters(), lr=base_lr * B / 64, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
g = torch.Generator(device=dev).manual_seed(1234)
im_q = torch.randn(B, 3, 32, hw, hw, device=dev, generator=g)
im_k = torch.randn(B, 3, 32, hw, hw, device=dev, generator=g)


def step():
    out, tgt, rl, rt = model(im_q, im_k)
    loss, _, _ = crit(out, tgt, rl, rt)
    opt.zero_grad()
    loss.backward()
    opt.step()


for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
be = ops.backend()
be.event_log = []
e0 = torch.cuda.Event(enable_timing=True)
e1 = torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(args.steps):
    step()
e1.record()
torch.cuda.synchronize()
log, be.event_log = be.event_log, None
step_ms = e0.elapsed_time(e1) / args.steps
tab = {}
for kind, f, a, b, kernel, nbytes, geom, _fx in log:
    key = (kind, geom, kernel)
    t = tab.setdefault(key, [0, 0.0, 0.0])
    t[0] += 1
    t[1] += a.elapsed_time(b)
    t[2] += f
tot = sum(t[1] for t in tab.values()) / args.steps
print(f"{args.arch}: step {step_ms:.2f} ms, conv launches {tot:.2f} ms/step, {sum(t[0] for t in tab.values()) // args.steps} conv calls/step")
print(f"{'pass':6s} {'N x D x H x W x Cin -> Cout  k / s':52s} {'n/step':>6s} {'ms/step':>8s} {'TF':>7s} {'GF/call':>8s}  kernel")
cum = 0.0
for (kind, gm, kernel), t in sorted(tab.items(), key=lambda kv: -kv[1][1])[:args.top]:
    ms = t[1] / args.steps
    cum += ms
    shape = f"{gm.N}x{gm.Di}x{gm.Hi}x{gm.Wi}x{gm.Cin}->{gm.Cout} k{gm.k} s{gm.s}"
    print(f"{kind[5:]:6s} {shape:52s} {t[0] / args.steps:6.1f} {ms:8.3f} {t[2] / t[1] / 1e9:7.1f} {t[2] / t[0] / 1e9:8.2f}  {kernel}   (cum {cum:.1f})")
