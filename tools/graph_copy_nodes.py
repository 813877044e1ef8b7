"""Does a REPLAYED pretext step (lanes) contain memset / memcpy nodes?  (torch.profiler over one replayed step)
Round 6: a hipMemsetAsync captured into a linear graph was not reliably ordered against its neighbour kernels (R3D-18's shortcut input
gradients); this lists what is left.     python3 tools/graph_copy_nodes.py resnet18 32 112"""
import collections
import os
import random
import sys

import torch
from torch.profiler import ProfilerActivity, profile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_spec  # noqa: E402
from model_util import make_cfg  # noqa: E402
from oracle import portable as P  # noqa: E402


def main():
    arch, B, HW = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    K = 16384
    dev = torch.device("cuda", 0)
    from rspnet_amd.graph_step import GraphedPretextStep
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD
    torch.manual_seed(7)
    random.seed(7)
    wrapped = ModelFactory(make_cfg(arch, K)).build_moco_diffloss(device=dev)
    spec = dict(load_spec(arch))
    spec["queue"] = ((128, K), "float32")
    wrapped.module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.fill_state(spec, 3).items()})
    wrapped.train()
    crit = Loss(margin=2.0, A=1.0, M=1.0)
    opt = SGD(wrapped.parameters(), lr=0.05, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
    stepper = GraphedPretextStep(wrapped, crit, opt, warmup=2, issue="graph")
    im_q, im_k = (torch.from_numpy(c).to(dev) for c in P.clips(10, 0, (B, 3, 32, HW, HW)))
    bq, bk = stepper.clip_buffers(im_q, im_k)
    bq.copy_(im_q)
    bk.copy_(im_k)
    for _ in range(5):
        stepper(bq, bk)
    torch.cuda.synchronize()
    assert not stepper.disabled and stepper.graphs, stepper.fallback_reason
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA]) as prof:
        stepper(bq, bk)
        torch.cuda.synchronize()
    names = collections.Counter()
    for e in prof.events():
        n = e.name
        if "emcpy" in n or "emset" in n:
            names[n] += 1
    nk = sum(1 for e in prof.events() if e.device_type is not None and str(e.device_type).endswith("CUDA"))
    print(arch, "replayed step:", dict(names) if names else "no memcpy / memset activity", f"({nk} device events)")


if __name__ == "__main__":
    main()
