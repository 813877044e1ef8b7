"""GPU: one full pretext step of rspnet_amd (HIP kernels through the C ABI) against the golden fixtures generated
from the reference, and against the oracle restatement on the same seeded inputs.  North-star tolerance: loss / logits /
features within 1e-3 relative; gradients & post-SGD state are gated at the same bar."""
import pytest
import torch

from golden_util import build_inputs, cases_for, compare_to_golden, load_case, worst_grad_err, fwd_tol, grad_tol
from model_util import run_model_step

pytestmark = pytest.mark.gpu
TOL = 1e-3


CASES = [(a, s, "fused") for arch in ("c3d", "resnet18", "resnet34", "r2plus1d-vcop", "s3dg", "c3d:mlp", "c3d:conv", "c3d:convbn", "c3d:speednet", "c3d:linear:4", "c3d:linear:1", "resnet50") for a, w, s in cases_for(arch, 1)] + [
    ("c3d", cases_for("c3d", 1)[0][2], "torch")]


@pytest.mark.parametrize("arch,seed,optimizer", CASES)
def test_step_matches_golden(arch, seed, optimizer):
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    z, meta = load_case(arch, 1, seed)
    spec, inputs = build_inputs(arch, meta)
    from golden_util import check_step_gradients, grad_stats
    kept = {}

    def step():
        out = run_model_step(arch, meta, inputs, 0, torch.device("cuda", 0), optimizer)
        kept["stats"] = grad_stats(z, 0, out[3])
        st = kept["stats"]
        print(f"\n{arch} seed {seed} gradient tensors vs golden: worst {st['worst'][1]:.2e} ({st['worst'][0]}), p90 {st['p90']:.2e}, "
              f"median {st['median']:.2e}, whole gradient {st['whole']:.2e} ({st['n']} tensors)")
        return out

    errs, worst, plan, post = check_step_gradients(arch, 1, 0, z, step, TOL)
    assert list(post.keys()) == list(spec.keys())
    print(f"\n{arch} seed {seed} [{optimizer}] ({plan} tile plan) rel errs: " + ", ".join(f"{k}={v:.2e}" for k, v in errs.items())
          + f", grads={worst:.2e}")


def test_wrapper_forward_ncdhw_matches_oracle():
    """MultiTaskWrapper.forward(x) with the reference's NCDHW input (split_wrapper.py:128-152) vs the restatement."""
    import numpy as np
    from golden_util import load_spec
    from model_util import make_cfg
    from oracle import portable as P
    from oracle import restatement as S
    from rspnet_amd.moco import ModelFactory
    dev = torch.device("cuda", 0)
    model = ModelFactory(make_cfg("c3d", 64)).build_moco_diffloss(device=dev).module
    state = P.fill_state(load_spec("c3d"), 5)
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    x = torch.from_numpy(P.clips(5, 0, (4, 3, 16, 32, 32))[0])
    a, m = model.encoder_q(x.to(dev))
    sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    ra, rm, _ = S.encoder_forward("c3d", sd, "encoder_q", x)
    assert float((a.cpu() - ra).abs().max()) < 1e-4 and float((m.cpu() - rm).abs().max()) < 1e-4
    # train-mode BN side effect: running stats moved like the restatement's
    got = model.state_dict()["encoder_q.encoder.bn5b.running_mean"].cpu()
    assert float((got - sd["encoder_q.encoder.bn5b.running_mean"]).abs().max()) < 1e-5


def test_resnet_stem_pool_fusion_changes_no_bit(monkeypatch):
    """engine._pool_fusion: bn1 + relu + the 3x3x3/2 max-pool of the ResNet stem run as one kernel — without the arg-max in a forward
    that keeps nothing (key passes), with it in the one a backward follows.  Against the three ops run apart (RSP_NO_POOL_FUSION=1):
    same embeddings and, after a backward, the same stem gradients, bit for bit."""
    from model_util import make_cfg
    from oracle import portable as P
    from golden_util import load_spec
    from rspnet_amd import engine
    from rspnet_amd.moco import ModelFactory
    dev = torch.device("cuda", 0)
    out = {}
    for fused in (True, False):
        if fused:
            monkeypatch.delenv("RSP_NO_POOL_FUSION", raising=False)
        else:
            monkeypatch.setenv("RSP_NO_POOL_FUSION", "1")
        model = ModelFactory(make_cfg("resnet18", 64)).build_moco_diffloss(device=dev).module
        state = P.fill_state(load_spec("resnet18"), 5)
        model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
        enc = model.encoder_q
        enc.train()
        assert engine._pool_fusion(enc.plan()) == ({0: 1} if fused else {})
        xn = enc._to_ndhwc(torch.from_numpy(P.clips(7, 0, (4, 3, 16, 64, 64))[0]).to(dev))
        sd0 = {k: v.clone() for k, v in enc.state_dict().items()}
        with torch.no_grad():
            a0, m0, ctx0 = enc.forward_ndhwc(xn, keep=False)
            enc.load_state_dict(sd0)                               # (running statistics back)
            a1, m1, ctx1 = enc.forward_ndhwc(xn, keep=True)
            assert ctx0 is None and ctx1 is not None and torch.equal(a0, a1) and torch.equal(m0, m1)
            grads = {}

            def grad_of(prm):
                return grads.setdefault(id(prm), torch.zeros_like(prm))
            enc.backward_ndhwc(ctx1, torch.ones_like(a1) * 0.01, torch.ones_like(m1) * -0.02, grad_of)
        torch.cuda.synchronize()
        out[fused] = (a0.clone(), m0.clone(), grads[id(enc.encoder.conv1.weight)].clone(), grads[id(enc.encoder.bn1.weight)].clone())
    for x, y in zip(out[True], out[False]):
        assert torch.equal(x, y)
