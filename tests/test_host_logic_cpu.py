"""CPU: rspnet_amd's host logic (plan executor, flat params, MoCo forward/backward orchestration, fused-SGD
bookkeeping) reproduces the golden fixtures when its HIP ops are replaced by the torch checker backend.
The kernels themselves are tested on the GPU (tests/test_kernels_gpu.py, tests/test_step_gpu.py).  Every fixture family runs
here too — ResNet-34 / -50 (Bottleneck) and the speed-1 case included — so that the teacher-forced replay's chain
"HIP kernel == checker op at 2e-5" + "checker == reference fixture" has both links for every architecture."""
import numpy as np
import pytest
import torch

from cpu_ops import CpuOps
from golden_util import build_inputs, cases_for, compare_to_golden, load_case, worst_grad_err, fwd_tol, grad_tol
from model_util import run_model_step
from rspnet_amd import ops


@pytest.fixture()
def cpu_backend():
    prev = ops.set_backend(CpuOps())
    yield
    ops.set_backend(prev)


_C3D = cases_for("c3d", 1)
CASES = [(_C3D[0][0], _C3D[0][2], "fused"), (_C3D[1][0], _C3D[1][2], "torch")] + [
    (a, s, "fused") for arch in ("resnet18", "r2plus1d-vcop", "s3dg", "c3d:mlp", "c3d:conv", "c3d:convbn", "c3d:speednet", "c3d:linear:4", "c3d:linear:1",
                                  "resnet34", "resnet50") for a, w, s in cases_for(arch, 1)]


@pytest.mark.parametrize("arch,seed,optimizer", CASES)
def test_step_matches_golden_ws1(cpu_backend, arch, seed, optimizer):
    z, meta = load_case(arch, 1, seed)
    spec, inputs = build_inputs(arch, meta)
    res, post, mom_post, grads = run_model_step(arch, meta, inputs, 0, torch.device("cpu"), optimizer)
    # state-dict contract: same keys, shapes, dtypes as the reference
    assert list(post.keys()) == list(spec.keys())
    for k, (shape, dtype) in spec.items():
        assert tuple(post[k].shape) == shape and str(post[k].dtype) == dtype, k
    compare_to_golden(z, 0, res, post, mom_post, tol=fwd_tol(arch, 2e-4), tol_grad=grad_tol(arch))
    wkey, worst = worst_grad_err(z, 0, grads)
    assert worst <= grad_tol(arch), (wkey, worst)
