"""Generate tests/golden/finetune_<arch>.npz from the REFERENCE's MultiTaskWrapper(finetune=True) (build container only).
TEST INFRASTRUCTURE ONLY.

    python -m oracle.gen_golden_finetune

Per backbone: portable state (oracle.portable.fill_state over the reference model's own state-dict spec, with the guard band of
oracle/guard.py around the encoder's ReLU decisions: the moved BatchNorm bias values travel in the file), portable clips, labels
-> eval-mode logits, train-mode logits, CrossEntropyLoss, summaries of every parameter gradient, post-forward BN buffers.
The restatement (oracle.restatement.finetune_step / finetune_forward) is checked against the same run before writing."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import portable as P
from oracle import ref_harness as R
from oracle import restatement as S

# (arch, B, T, HW, classes, seed)
CASES = [("c3d", 4, 16, 32, 11, 4), ("resnet18", 4, 16, 64, 11, 9), ("r2plus1d-vcop", 4, 16, 32, 11, 1), ("s3dg", 4, 16, 64, 11, 8)]
# S3D-G's deep stack decides some ReLU masks / pool arg-maxes by rounding on every seed (same situation as its pretext
# fixtures: tests/golden_util.py, oracle/gen_conditioning.py); its screen is looser
SEED_GATE = {"s3dg": 2e-2}


def product_grad_error(arch, ncls, state, x, target, grads):
    """Worst relative L2 gradient distance (estimated from 16 random projections, oracle/portable.py) of the product's host
    logic on the torch checker backend: used only to reject seeds whose tiny late layers hold a ReLU / max-pool decision
    inside the fp32 rounding band ("Gradient tolerance")."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from cpu_ops import CpuOps
    from golden_util import summary_err
    from rspnet_amd import ops
    from rspnet_amd.models import get_model_class
    from rspnet_amd.moco.split_wrapper import MultiTaskWrapper
    prev = ops.set_backend(CpuOps())
    try:
        model = MultiTaskWrapper(get_model_class(arch=arch), num_classes=ncls, finetune=True)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
        model.train()
        loss = torch.nn.CrossEntropyLoss()(model(torch.from_numpy(x)), torch.from_numpy(target))
        loss.backward()
        return max(P.proj_rel_err(n, p.grad.numpy(), P.projections(n, grads[n])) for n, p in model.named_parameters()
                   if grads[n] is not None and float(np.sqrt((grads[n].astype(np.float64) ** 2).sum())) >= 1e-4)   # (skip the
        # mathematically-zero gradients of conv biases in front of train-mode BN: rounding noise in the reference)
    finally:
        ops.set_backend(prev)


def main():
    torch.manual_seed(0)
    only = sys.argv[1:]
    for arch, B, T, HW, ncls, seed0 in CASES:
        if only and arch not in only:
            continue
        model = R.build_reference_finetune(arch, ncls)
        spec = R.state_spec(model)
        from oracle import guard
        for seed in range(seed0, seed0 + 24):
            state = P.fill_state(spec, seed)
            x = P.clips(seed, 0, (B, 3, T, HW, HW))[0]
            target = ((np.arange(B) * 3 + seed) % ncls).astype(np.int64)
            nudges, rep = guard.guard_band(arch, "linear", state, [x], forward=lambda sd, xx: S.finetune_forward(arch, sd, xx, training=True))
            guard.apply_nudges(state, nudges)
            print(f"{arch} seed {seed}: guard {rep}", flush=True)
            le, lt, loss, grads, post = R.run_reference_finetune(model, state, x, target)
            margin = float(post.pop("__margin__"))
            perr = product_grad_error(arch, ncls, state, x, target, grads)
            print(f"{arch} seed {seed}: checker-backend gradient error {perr:.1e}, ReLU / pool margin {margin:.1e}")
            if perr <= SEED_GATE.get(arch, 3e-4) and (margin >= 3e-6 or arch == "s3dg"):
                break
        else:
            raise SystemExit(f"{arch}: no well-conditioned seed found")
        # restatement vs reference
        sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
        mle = S.finetune_forward(arch, sd, torch.from_numpy(x), training=False)
        mlt, mloss, mg = S.finetune_step(arch, sd, torch.from_numpy(x), torch.from_numpy(target))
        e1 = float((mle - torch.from_numpy(le)).abs().max())
        e2 = float((mlt - torch.from_numpy(lt)).abs().max())
        e3 = max(float((mg[k] - torch.from_numpy(g)).abs().max()) for k, g in grads.items() if g is not None)
        assert all((mg[k] is None) == (g is None) for k, g in grads.items())
        print(f"{arch}: loss {loss:.5f}; restatement vs reference: eval logits {e1:.1e}, train logits {e2:.1e}, grads {e3:.1e}")
        assert max(e1, e2) <= 1e-5 and e3 <= 1e-5 and abs(float(mloss) - loss) <= 1e-6
        out = {"meta": np.frombuffer(json.dumps({"arch": arch, "B": B, "T": T, "HW": HW, "classes": ncls, "seed": seed}).encode(),
                                     dtype=np.uint8),
               "target": target, "logits_eval": le, "logits": lt, "loss": np.float64(loss)}
        for k, g in grads.items():
            out["gradsum." + k] = np.zeros(0) if g is None else P.summarise(k, g)
            if g is not None:
                out["gradproj." + k] = P.projections(k, g)
        for k, v in post.items():
            if k.endswith(("running_mean", "running_var", "num_batches_tracked")):
                out["post." + k] = P.summarise(k, v) if v.ndim else np.asarray(v)
        for k, (idx, val) in nudges.items():
            out["nudge.idx." + k] = np.asarray(idx, dtype=np.int32)
            out["nudge.val." + k] = np.asarray(val, dtype=np.float32)
        tag = arch.replace("-", "_")
        with open(os.path.join(ROOT, "tests", "golden", f"finetune_spec_{tag}.json"), "w") as f:
            json.dump({k: [list(s), d] for k, (s, d) in spec.items()}, f)
        np.savez_compressed(os.path.join(ROOT, "tests", "golden", f"finetune_{tag}.npz"), **out)


if __name__ == "__main__":
    main()
