"""Where does the host block inside a pretext step, and what causes the occasional long step?  Prints, for N steps of the
bench workload: per-step GPU interval / host enqueue time / allocator state / garbage-collector pauses, then a cProfile of
the host side (tottime: the call the host sits in while the GPU drains)."""
import argparse
import cProfile
import gc
import os
import pstats
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402

from bench import ARCHS  # noqa: E402
from rspnet_amd.moco import Loss, ModelFactory  # noqa: E402
from rspnet_amd.optim import SGD  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--arch", default="c3d")
ap.add_argument("--steps", type=int, default=40)
ap.add_argument("--warmup", type=int, default=5)
ap.add_argument("--no-gc", action="store_true")
ap.add_argument("--profile", action="store_true")
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
    return loss


gc_log = []
_t = [0.0]


def gc_cb(phase, info):
    if phase == "start":
        _t[0] = time.perf_counter()
    else:
        gc_log.append((info["generation"], (time.perf_counter() - _t[0]) * 1e3, info.get("collected", 0)))


gc.callbacks.append(gc_cb)
for _ in range(args.warmup):
    step()
torch.cuda.synchronize()
if args.no_gc:
    gc.collect()
    gc.freeze()
    gc.disable()
marks = [torch.cuda.Event(enable_timing=True)]
marks[0].record()
rows = []
prof = cProfile.Profile() if args.profile else None
if prof:
    prof.enable()
for i in range(args.steps):
    n_gc = len(gc_log)
    st = torch.cuda.memory_stats()
    h0 = time.perf_counter()
    step()
    h = (time.perf_counter() - h0) * 1e3
    marks.append(torch.cuda.Event(enable_timing=True))
    marks[-1].record()
    st2 = torch.cuda.memory_stats()
    rows.append((h, st2["reserved_bytes.all.current"] / 2**20, st2["num_device_alloc"] - st["num_device_alloc"],
                 st2["num_device_free"] - st["num_device_free"], st2["num_alloc_retries"], gc_log[n_gc:]))
if prof:
    prof.disable()
torch.cuda.synchronize()
for i, r in enumerate(rows):
    print(f"step {i:3d} gpu {marks[i].elapsed_time(marks[i + 1]):8.2f} ms host {r[0]:8.2f} ms reserved {r[1]:8.0f} MiB "
          f"hipMalloc +{r[2]} hipFree +{r[3]} retries {r[4]} gc {[(g_, round(ms, 1), c) for g_, ms, c in r[5]]}")
if prof:
    pstats.Stats(prof).sort_stats("tottime").print_stats(18)
