"""Record the structure of the REAL reference's checkpoint['optimizer'] entry.  TEST INFRASTRUCTURE ONLY.

``python -m oracle.gen_golden_optimizer`` (build container only; needs /root/reference) builds the reference's
MoCoDiffLossTwoFc(C3D), the optimizer exactly as pretrain.py:65-72 does (torch.optim.SGD over model.parameters(), i.e. the
frozen encoder_k parameters included), runs one tiny step so the momentum buffers exist, and writes
tests/golden/optimizer_state_c3d.json: parameter names in model.parameters() order, param_groups (hyper-parameters + index
list) and the shapes of the state entries.  Values are not stored (momentum buffers are re-derived from oracle/portable.py)."""
from __future__ import annotations

import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)


def reference_optimizer_state(K=64, B=4, HW=32, seed=2):
    import torch
    from oracle import portable as P
    from oracle import ref_harness as R
    from oracle.gen_golden import case_inputs
    R.ensure_process_group()
    model = R.build_reference_model("c3d", K=K)
    spec = R.state_spec(model)
    state, mom, clips, perms_B, sh = case_inputs(spec, "c3d", B, HW, K, 1, seed)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    model.train()
    from moco.builder_diffspeed_diffloss import Loss
    opt = torch.optim.SGD(model.parameters(), lr=0.05, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
    with R._ReplayRNG([perms_B[0], sh[0], sh[1]], 2):
        out, tgt, rl, rt = model(torch.from_numpy(clips[0][0]), torch.from_numpy(clips[0][1]))
    loss, _, _ = Loss(margin=2.0, A=1.0, M=1.0)(out, tgt, rl, rt.view(-1, 1))
    opt.zero_grad()
    loss.backward()
    opt.step()
    return opt.state_dict(), [n for n, _ in model.named_parameters()]


def main():
    sd, names = reference_optimizer_state()
    out = {"param_names": names, "param_groups": sd["param_groups"],
           "state_shapes": {str(i): list(st["momentum_buffer"].shape) for i, st in sd["state"].items()}}
    path = os.path.join(ROOT, "tests", "golden", "optimizer_state_c3d.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", path, len(names), "params,", len(out["state_shapes"]), "with state")


if __name__ == "__main__":
    main()
