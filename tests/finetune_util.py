"""Shared by the fine-tune tests: load a tests/golden/finetune_<arch>.npz case, rebuild its portable inputs, run the product's
MultiTaskWrapper(finetune=True) on a device / op backend, compare with the fixture."""
import json
import os

import numpy as np
import torch

from golden_util import grad_tol, rel_err, summary_err
from oracle import portable as P

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ARCHS = ["c3d", "resnet18", "r2plus1d-vcop", "s3dg"]


def load(arch):
    tag = arch.replace("-", "_")
    z = np.load(os.path.join(GOLDEN, f"finetune_{tag}.npz"))
    meta = json.loads(bytes(z["meta"]).decode())
    with open(os.path.join(GOLDEN, f"finetune_spec_{tag}.json")) as f:
        spec = {k: (tuple(s), d) for k, (s, d) in json.load(f).items()}
    state = P.fill_state(spec, meta["seed"])
    from oracle.gen_golden import nudges_from_npz
    for key, (idx, val) in nudges_from_npz(z).items():      # the fixture's guard band (oracle/guard.py): part of its state
        state[key][np.asarray(idx, dtype=np.int64)] = np.asarray(val, dtype=np.float32)
    x = P.clips(meta["seed"], 0, (meta["B"], 3, meta["T"], meta["HW"], meta["HW"]))[0]
    return z, meta, spec, state, x


def build_model(arch, ncls, state, device):
    from rspnet_amd.models import get_model_class
    from rspnet_amd.moco.split_wrapper import MultiTaskWrapper
    model = MultiTaskWrapper(get_model_class(arch=arch), num_classes=ncls, finetune=True).to(device)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    return model


def check_case(arch, device, fwd_tol):
    z, meta, spec, state, x = load(arch)
    model = build_model(arch, meta["classes"], state, device)
    # state-dict contract of the reference's MultiTaskWrapper(finetune=True): same keys, shapes, dtypes
    sd = model.state_dict()
    assert list(sd.keys()) == list(spec.keys())
    for k, (shape, dtype) in spec.items():
        assert tuple(sd[k].shape) == shape and str(sd[k].dtype).replace("torch.", "") == dtype, k
    xt = torch.from_numpy(x).to(device)
    tt = torch.from_numpy(z["target"]).to(device)
    model.eval()
    with torch.no_grad():
        le = model(xt)
    assert rel_err(le.cpu().numpy(), z["logits_eval"]) <= fwd_tol, ("eval logits", rel_err(le.cpu().numpy(), z["logits_eval"]))
    assert not any(k.endswith("num_batches_tracked") and int(v) != int(state[k]) for k, v in model.state_dict().items())
    model.train()
    logits = model(xt)
    loss = torch.nn.CrossEntropyLoss()(logits, tt)
    loss.backward()
    assert rel_err(logits.detach().cpu().numpy(), z["logits"]) <= fwd_tol
    assert abs(float(loss.detach()) - float(z["loss"])) <= fwd_tol * max(1.0, abs(float(z["loss"])))
    worst = 0.0
    for n, p in model.named_parameters():
        g = z["gradsum." + n]
        if g.size == 0:
            assert p.grad is None, n
        else:
            assert p.grad is not None, n
            mine = p.grad.detach().cpu().numpy()
            if "gradproj." + n in z.files and g[0] >= 1e-4:
                l2 = float(np.sqrt((mine.astype(np.float64) ** 2).sum()))
                worst = max(worst, P.proj_rel_err(n, mine, z["gradproj." + n]), abs(l2 - g[0]) / g[0])
            else:
                worst = max(worst, summary_err(n, mine, g))
    # (the fine-tune fixtures carry the same guard band as the pretext ones and are screened against the checker backend at 3e-4
    #  — S3D-G 2e-2 —: the pretext family's gate)
    assert worst <= grad_tol(arch), worst
    post = model.state_dict()
    for name in z.files:
        if name.startswith("post."):
            k = name[5:]
            v = post[k].detach().cpu().numpy()
            if v.ndim == 0:
                assert int(v) == int(z[name]), k
            else:
                assert rel_err(P.summarise(k, v), z[name]) <= max(fwd_tol, 2e-5), k
    return worst
