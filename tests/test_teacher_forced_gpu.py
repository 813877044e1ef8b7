"""GPU: every HIP kernel call of one full pretext step, teacher-forced.  The step is first run on the CPU checker backend
(pinned to the reference goldens) under tests/teacher_forced.Recorder; each recorded op call is then replayed on the HIP
backend with the recorded inputs and compared at 2e-5 — forward AND backward of every ConvBN / Gate / Pool / head / loss unit
in the backbone's real composition (S3D-G's gate bwd -> concat slices -> fan-out adds -> overlapping max-pool bwd, the
Bottleneck's 1x1x1 / strided / residual units ...), where the whole-step gradient check has to allow for ReLU / arg-max
knife edges.  Elements whose mask is undecidable in fp32 get a zero incoming gradient in the recorded chain (see the
harness), so nothing left is ambiguous and the gate is tight."""
import pytest
import torch

from cpu_ops import CpuOps
from golden_util import build_inputs, cases_for, load_case
from model_util import run_model_step
from rspnet_amd import ops
from teacher_forced import Recorder, replay

pytestmark = pytest.mark.gpu
TOL = 2e-5
ARCHS = ["c3d", "c3d:mlp", "c3d:conv", "c3d:convbn", "c3d:speednet", "c3d:linear:4", "resnet18", "resnet34", "resnet50", "r2plus1d-vcop", "s3dg"]


@pytest.mark.parametrize("arch", [a for a in ARCHS if cases_for(a, 1)])
def test_every_op_of_a_step_teacher_forced(arch):
    a, ws, seed = cases_for(arch, 1)[0]
    z, meta = load_case(a, ws, seed)
    spec, inputs = build_inputs(a, meta)
    rec = Recorder(CpuOps())
    prev = ops.set_backend(rec)
    try:
        run_model_step(a, meta, inputs, 0, torch.device("cpu"), "fused")
    finally:
        ops.set_backend(prev)
    be = ops.backend()
    assert be.name == "hip"
    worst, where = replay(rec.calls, be, torch.device("cuda", 0), tol=TOL)
    torch.cuda.synchronize()
    for must in ("conv_fwd", "conv_dgrad_packed", "conv_wgrad", "bn_finalize", "bn_act_pool_fwd", "bn_act_pool_bwd", "logits_fwd",
                 "logits_bwd", "loss_fwd_bwd", "sgd_step", "momentum_update", "clip_gather", "queue_enqueue"):
        assert must in worst, must
    if arch == "s3dg":
        assert "gate_bwd" in worst and "maxpool_bwd" in worst
    print(arch, len(rec.calls), "calls,", rec.neutralised, "knife-edge elements neutralised; worst rel err per op:",
          {k: float(f"{v:.1e}") for k, v in sorted(worst.items())})
