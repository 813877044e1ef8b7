"""Value comparison of ONE pretext step at an arbitrary size: rspnet_amd (any device / op backend) against the oracle
restatement (oracle/restatement.py:moco_step, pinned to the reference by tests/test_oracle_golden.py and
tests/test_oracle_vs_reference.py) on the SAME seeded state, clips and injected permutations.

The committed fixtures hold the reference's outputs at B=4..8, 32..64 px, K=64 (a full-size state is 100-250 MB and a full-size
reference step needs the reference itself, which cannot travel to the GPU box); this helper closes the gap at the size that is
measured — BASELINE.json configs 2-5: B=32 (16 for S3D-G), 112x112 (224x224), K=16384 — by running the oracle live on the GPU
host's cores next to the HIP step.  Reference path: /root/reference/moco/builder_diffspeed_diffloss.py:492-547,
/root/reference/pretrain.py:157-165.
"""
import numpy as np
import torch

from golden_util import load_spec, rel_err, run_restatement
from model_util import run_model_step
from oracle.gen_golden import case_inputs

FWD_KEYS = ("loss", "loss_A", "loss_M", "logits1", "logits2", "l_pos_M", "l_neg_M", "q_A", "q_M",
            "k_A_shuf", "k_M_shuf", "kneg_A_shuf", "kneg_M_shuf")


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64).reshape(-1)
    b = np.asarray(b, dtype=np.float64).reshape(-1)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))


def full_size_meta(arch, B, HW, K, seed, lr=0.05, fc_type="linear", speed=2):
    return dict(arch=arch, fc_type=fc_type, B=B, HW=HW, K=K, ws=1, seed=seed, lr=lr, speed=speed, T_in=32, m=0.999, T=0.07,
                sgd_momentum=0.9, weight_decay=1e-4, margin=2.0, A=1.0, M=1.0)


def step_vs_oracle(arch, B, HW, K, seed, device, threads=None):
    """Returns (errs, detail): errs maps quantity -> relative error (max-norm for forward quantities / queue / BN statistics,
    relative L2 for gradient-derived tensors); detail carries the worst tensors."""
    meta = full_size_meta(arch, B, HW, K, seed)
    spec = dict(load_spec(arch))
    spec["queue"] = ((128, K), "float32")
    inputs = case_inputs(spec, arch, B, HW, K, 1, seed)
    res, post, mom_post, grads = run_model_step(arch, meta, inputs, 0, device, "fused")
    if threads:
        torch.set_num_threads(threads)
    outs, states, moms = run_restatement(arch, meta, inputs)
    o, st, mo = outs[0], states[0], moms[0]
    errs, detail = {}, {}
    for k in FWD_KEYS:
        errs[k] = rel_err(res[k], o[k].numpy())
    errs["queue"] = rel_err(post["queue"], st["queue"].numpy())
    assert int(np.asarray(post["queue_ptr"]).reshape(-1)[0]) == int(st["queue_ptr"][0])
    # BatchNorm side effects of the three encoder passes (running statistics moved by the batch moments)
    worst = ("", 0.0)
    for k, v in st.items():
        if k.endswith(("running_mean", "running_var")):
            e = rel_err(post[k], v.numpy(), floor=1e-3)
            if e > worst[1]:
                worst = (k, e)
        elif k.endswith("num_batches_tracked"):
            assert int(post[k]) == int(v), k
    errs["bn_running_stats"], detail["bn_running_stats"] = worst[1], worst[0]
    # momentum-updated key encoder (before the key passes) — exact formula, no discontinuity
    worst = ("", 0.0)
    for k, v in st.items():
        if k.startswith("encoder_k.") and not k.endswith(("running_mean", "running_var", "num_batches_tracked")):
            e = rel_err(post[k], v.numpy())
            if e > worst[1]:
                worst = (k, e)
    errs["encoder_k_params"], detail["encoder_k_params"] = worst[1], worst[0]
    # gradients: per tensor relative L2, and the whole gradient as one vector
    worst, num, den = ("", 0.0), 0.0, 0.0
    for k, g in o["grads"].items():
        if g is None:
            assert grads[k] is None, k
            continue
        g = g.numpy().astype(np.float64)
        mine = grads[k].astype(np.float64)
        n2, d2 = float(((mine - g) ** 2).sum()), float((g ** 2).sum())
        num, den = num + n2, den + d2
        if d2 < 1e-8:                       # conv bias in front of train-mode BN: identically zero in exact arithmetic
            assert n2 < 1e-6, k
            continue
        e = (n2 / d2) ** 0.5
        if e > worst[1]:
            worst = (k, e)
    errs["grad_worst_tensor"], detail["grad_worst_tensor"] = worst[1], worst[0]
    errs["grad_whole"] = (num / den) ** 0.5
    worst = ("", 0.0)
    for k, v in mo.items():
        if o["grads"].get(k) is None:
            continue
        e = rel_l2(mom_post[k], v.numpy())
        if e > worst[1]:
            worst = (k, e)
    errs["momentum_post"], detail["momentum_post"] = worst[1], worst[0]
    worst = ("", 0.0)
    for k in o["grads"]:
        e = rel_l2(post[k], st[k].numpy())
        if e > worst[1]:
            worst = (k, e)
    errs["encoder_q_params_post"], detail["encoder_q_params_post"] = worst[1], worst[0]
    return errs, detail
