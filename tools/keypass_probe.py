#!/usr/bin/env python
"""How much would one key-encoder pass over 2B clips save against two passes over B clips (the step runs two: k and k_negative,
builder_diffspeed_diffloss.py:507-515)?  Times encoder_k.forward_ndhwc (no grad, train-mode BN) at B and 2B per backbone."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from rspnet_amd.moco import ModelFactory
dev = torch.device("cuda", 0)
ARCHS = {"c3d": (32, 112), "resnet18": (32, 112), "r2plus1d-vcop": (32, 112), "s3dg": (16, 224)}
def timeit(fn, it=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / it
for arch in (sys.argv[1:] or list(ARCHS)):
    B, hw = ARCHS[arch]
    cfg = {"model": {"arch": arch}, "moco": {"dim": 128, "k": 16384, "m": 0.999, "t": 0.07, "fc_type": "linear", "diff_speed": [2]}}
    model = ModelFactory(cfg).build_moco_diffloss(device=dev).module
    model.train(); model._prepare()
    enc = model.encoder_k
    res = {}
    for b in (B, 2 * B):
        x = torch.randn(b, 16, hw, hw, 4, device=dev)
        with torch.no_grad():
            res[b] = timeit(lambda: enc.forward_ndhwc(x, keep=False))
    print(f"{arch:14s} B={B}: {res[B]:7.2f} ms   2B={2*B}: {res[2*B]:7.2f} ms   2 x B = {2*res[B]:7.2f} ms   saving {2*res[B]-res[2*B]:6.2f} ms ({100*(1-res[2*B]/(2*res[B])):.1f} %)", flush=True)
    del model, enc
    torch.cuda.empty_cache()
