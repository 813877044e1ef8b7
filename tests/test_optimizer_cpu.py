"""CPU (checker backend): optimizer interplay with the packed-weight cache and the reference's optimizer checkpoint entry.

* Several consecutive steps with torch.optim.SGD / nesterov (paths that update the parameters WITHOUT going through the
  fused kernel) must track the fused-kernel run: the query encoder's packed forward weights have to be rebuilt whenever any
  optimizer moved the parameters (they were only invalidated by the fused path, so step >= 2 silently diverged).
* SGD is built over model.parameters() exactly as pretrain.py:65-72 does (frozen encoder_k included), so the reference's
  checkpoint['optimizer'] loads here and ours loads there."""
import copy
import json
import os

import numpy as np
import pytest
import torch

from cpu_ops import CpuOps
from golden_util import GOLDEN, build_inputs, cases_for, load_case
from model_util import ReplayRNG, make_cfg
from oracle import portable as P
from rspnet_amd import ops
from rspnet_amd.moco import Loss, ModelFactory
from rspnet_amd.optim import SGD


@pytest.fixture()
def cpu_backend():
    prev = ops.set_backend(CpuOps())
    yield
    ops.set_backend(prev)


def _build(meta, state):
    wrapped = ModelFactory(make_cfg(meta["arch"], meta["K"], m=meta["m"], T=meta["T"])).build_moco_diffloss(device=torch.device("cpu"))
    wrapped.module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    wrapped.train()
    return wrapped


def _run_steps(meta, inputs, make_opt, nsteps=3):
    state, mom, clips, perms_B, sh = inputs
    wrapped = _build(meta, state)
    opt = make_opt(wrapped)
    crit = Loss(margin=meta["margin"], A=meta["A"], M=meta["M"])
    im_q, im_k = torch.from_numpy(clips[0][0]), torch.from_numpy(clips[0][1])
    losses, feats = [], []
    for it in range(nsteps):
        B = im_q.shape[0]
        perms = [P.permutation(f"ms{it}:{j}", meta["seed"], B) for j in range(3)]
        with ReplayRNG(perms, meta["speed"]):
            out, tgt, rl, rt = wrapped(im_q, im_k)
        loss, _, _ = crit(out, tgt, rl, rt)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(float(loss.detach()))
        feats.append(wrapped.module._last_q[0].clone())
    return losses, feats, wrapped, opt


def _case():
    arch, ws, seed = cases_for("c3d", 1)[0]
    z, meta = load_case(arch, ws, seed)
    return meta, build_inputs(arch, meta)[1]


def _kw(meta, **over):
    kw = dict(lr=meta["lr"], momentum=meta["sgd_momentum"], dampening=0.0, weight_decay=meta["weight_decay"], nesterov=False)
    kw.update(over)
    return kw


def test_torch_sgd_tracks_fused_sgd_over_several_steps(cpu_backend):
    meta, inputs = _case()
    lf, ff, _, _ = _run_steps(meta, inputs, lambda w: SGD(w.parameters(), **_kw(meta)))
    lt, ft, _, _ = _run_steps(meta, inputs, lambda w: torch.optim.SGD(w.parameters(), **_kw(meta)))
    assert abs(lf[0] - lt[0]) < 1e-6
    assert abs(lf[1] - lf[0]) > 1e-3                       # the steps do move the loss: the comparison below is not vacuous
    for a, b in zip(lf, lt):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (lf, lt)
    for a, b in zip(ff, ft):
        assert float((a - b).abs().max()) <= 2e-5


def test_unfused_options_take_torch_path_and_stay_consistent(cpu_backend):
    """nesterov=True is not covered by the fused kernel: rspnet_amd.optim.SGD falls back to torch's step for the flat
    parameters and must invalidate the packed weights afterwards."""
    meta, inputs = _case()
    ls, fs, _, _ = _run_steps(meta, inputs, lambda w: SGD(w.parameters(), **_kw(meta, nesterov=True)))
    lt, ft, _, _ = _run_steps(meta, inputs, lambda w: torch.optim.SGD(w.parameters(), **_kw(meta, nesterov=True)))
    for a, b in zip(ls, lt):
        assert abs(a - b) <= 2e-5 * max(1.0, abs(a)), (ls, lt)
    for a, b in zip(fs, ft):
        assert float((a - b).abs().max()) <= 2e-5


def test_manual_parameter_edit_is_seen_by_next_forward(cpu_backend):
    meta, inputs = _case()
    state = inputs[0]
    wrapped = _build(meta, state)
    im_q, im_k = torch.from_numpy(inputs[2][0][0]), torch.from_numpy(inputs[2][0][1])
    perms = [inputs[3][0], inputs[4][0], inputs[4][1]]
    with ReplayRNG(perms, meta["speed"]):
        wrapped(im_q, im_k)
    q0 = wrapped.module._last_q[0].clone()
    with torch.no_grad():
        for p in wrapped.module.encoder_q.parameters():
            if p.dim() == 5:                       # (a plain rescale would be undone by the BatchNorm behind each conv)
                p.add_(torch.from_numpy(P.uniform("edit", 1, tuple(p.shape), -1.0, 1.0)) * p.abs().mean())
    wrapped.module.queue_ptr.zero_()
    with ReplayRNG(perms, meta["speed"]):
        wrapped(im_q, im_k)
    assert float((wrapped.module._last_q[0] - q0).abs().max()) > 1e-4


# ---- checkpoint['optimizer'] interchange ------------------------------------------------------------------------------
OPT_SPEC = os.path.join(GOLDEN, "optimizer_state_c3d.json")


def _reference_like_state_dict(spec, seed=3):
    """An optimizer state_dict with the reference's structure (oracle/gen_golden_optimizer.py recorded it from
    torch.optim.SGD(model.parameters()) of the real reference after one step) and portable momentum values."""
    sd = {"param_groups": copy.deepcopy(spec["param_groups"]), "state": {}}
    for idx, shape in spec["state_shapes"].items():
        sd["state"][int(idx)] = {"momentum_buffer": torch.from_numpy(P.uniform(f"optmom:{idx}", seed, tuple(shape), -0.01, 0.01))}
    return sd


def test_reference_optimizer_state_dict_loads(cpu_backend):
    with open(OPT_SPEC) as f:
        spec = json.load(f)
    meta, inputs = _case()
    wrapped = _build(meta, inputs[0])
    names = [n for n, _ in wrapped.module.named_parameters()]
    assert names == spec["param_names"]                               # same parameter order as the reference model
    opt = SGD(wrapped.parameters(), **_kw(meta))
    assert len(opt.param_groups[0]["params"]) == len(spec["param_groups"][0]["params"])
    sd = _reference_like_state_dict(spec)
    opt.load_state_dict(sd)
    # the loaded momentum is what the next (fused) step uses
    params = list(wrapped.parameters())
    i0 = sorted(int(i) for i in spec["state_shapes"])[0]
    before = opt.state[params[i0]]["momentum_buffer"].clone()
    im_q, im_k = torch.from_numpy(inputs[2][0][0]), torch.from_numpy(inputs[2][0][1])
    with ReplayRNG([inputs[3][0], inputs[4][0], inputs[4][1]], meta["speed"]):
        out, tgt, rl, rt = wrapped(im_q, im_k)
    loss, _, _ = Loss(margin=2.0)(out, tgt, rl, rt)
    opt.zero_grad()
    loss.backward()
    g = params[i0].grad.clone()
    w = params[i0].detach().clone()
    opt.step()
    expect = meta["sgd_momentum"] * before + (g + meta["weight_decay"] * w)
    got = opt.state[params[i0]]["momentum_buffer"]
    assert float((got - expect).abs().max()) <= 1e-6 * max(1.0, float(expect.abs().max()))
    # and ours goes back into a plain torch.optim.SGD built the reference's way
    theirs = torch.optim.SGD(wrapped.parameters(), **_kw(meta))
    theirs.load_state_dict(opt.state_dict())
    assert set(theirs.state_dict()["state"].keys()) == {int(i) for i in spec["state_shapes"]}


def test_live_reference_optimizer_state_dict(cpu_backend):
    from oracle import ref_harness as R
    if not R.reference_available():
        pytest.skip("/root/reference not present (GPU box)")
    from oracle.gen_golden_optimizer import reference_optimizer_state
    sd, names = reference_optimizer_state()
    meta, inputs = _case()
    wrapped = _build(meta, inputs[0])
    assert [n for n, _ in wrapped.module.named_parameters()] == names
    opt = SGD(wrapped.parameters(), **_kw(meta))
    opt.load_state_dict(sd)
    with open(OPT_SPEC) as f:
        spec = json.load(f)
    assert sorted(sd["state"].keys()) == sorted(int(i) for i in spec["state_shapes"])
    assert len(sd["param_groups"][0]["params"]) == len(spec["param_groups"][0]["params"])
