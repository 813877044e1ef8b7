"""CPU: the record/replay harness of tests/teacher_forced.py is self-consistent — a step recorded on the checker backend
replays on a fresh checker backend with zero error, the knife-edge neutralisation touches only a handful of elements, and the
recorded chain still matches the reference golden (so what the GPU test replays IS the reference computation)."""
import pytest
import torch

from cpu_ops import CpuOps
from golden_util import build_inputs, cases_for, compare_to_golden, fwd_tol, load_case
from model_util import run_model_step
from rspnet_amd import ops
from teacher_forced import Recorder, replay


@pytest.mark.parametrize("arch", ["c3d", "s3dg"])
def test_recorded_step_replays_exactly(arch):
    a, ws, seed = cases_for(arch, 1)[0]
    z, meta = load_case(a, ws, seed)
    spec, inputs = build_inputs(a, meta)
    rec = Recorder(CpuOps())
    prev = ops.set_backend(rec)
    try:
        res, post, mom_post, grads = run_model_step(a, meta, inputs, 0, torch.device("cpu"), "fused")
    finally:
        ops.set_backend(prev)
    compare_to_golden(z, 0, res, post, None, tol=fwd_tol(a, 2e-4), check=("fwd",))
    names = {c.name for c in rec.calls}
    assert {"conv_fwd", "conv_dgrad_packed", "conv_wgrad", "bn_act_pool_bwd", "sgd_step", "logits_bwd"} <= names
    total = sum(c.args[3].data.numel() for c in rec.calls if c.name == "bn_act_pool_bwd")
    assert rec.neutralised <= 1e-3 * total, (rec.neutralised, total)
    worst, _ = replay(rec.calls, CpuOps(), torch.device("cpu"), tol=1e-6)
    assert max(worst.values()) <= 1e-6
