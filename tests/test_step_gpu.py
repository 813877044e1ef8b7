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
    from golden_util import check_step_gradients
    errs, worst, plan, post = check_step_gradients(arch, 1, 0, z, lambda: run_model_step(arch, meta, inputs, 0, torch.device("cuda", 0),
                                                                                      optimizer), TOL)
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
