"""GPU: one full pretext step of rspnet_amd (HIP kernels through the C ABI) against the golden fixtures generated
from the reference, and against the oracle restatement on the same seeded inputs.  North-star tolerance: loss / logits /
features within 1e-3 relative; gradients & post-SGD state are gated at the same bar."""
import pytest
import torch

from golden_util import build_inputs, cases_for, compare_to_golden, load_case, summary_err, fwd_tol, grad_tol
from model_util import run_model_step

pytestmark = pytest.mark.gpu
TOL = 1e-3


CASES = [(a, s, "fused") for arch in ("c3d", "resnet18", "r2plus1d-vcop", "s3dg", "c3d:mlp") for a, w, s in cases_for(arch, 1)] + [
    ("c3d", cases_for("c3d", 1)[0][2], "torch")]


@pytest.mark.parametrize("arch,seed,optimizer", CASES)
def test_step_matches_golden(arch, seed, optimizer):
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    z, meta = load_case(arch, 1, seed)
    spec, inputs = build_inputs(arch, meta)
    res, post, mom_post, grads = run_model_step(arch, meta, inputs, 0, torch.device("cuda", 0), optimizer)
    assert list(post.keys()) == list(spec.keys())
    errs = compare_to_golden(z, 0, res, post, mom_post, tol=TOL, tol_grad=grad_tol(arch))
    worst = 0.0
    for name in z.files:
        if name.startswith("r0.gradsum."):
            key = name[len("r0.gradsum."):]
            if z[name].size == 0:
                assert grads[key] is None, key
            else:
                worst = max(worst, summary_err(key, grads[key], z[name]))
    assert worst <= grad_tol(arch), worst
    print(f"\n{arch} seed {seed} [{optimizer}] rel errs: " + ", ".join(f"{k}={v:.2e}" for k, v in errs.items())
          + f", grads={worst:.2e}")
