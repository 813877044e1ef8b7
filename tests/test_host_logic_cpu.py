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


def test_graphed_step_wrapper_falls_back_to_the_eager_statements(cpu_backend):
    """rspnet_amd.graph_step.GraphedPretextStep off the GPU (or with more than one rank) issues the reference's five statements
    eagerly, split as host part / device part — same results as calling the model directly, same golden."""
    import random
    from model_util import ReplayRNG, make_cfg
    from rspnet_amd.graph_step import GraphedPretextStep
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD
    arch, _, seed = _C3D[0]
    z, meta = load_case(arch, 1, seed)
    spec, (state, mom, clips, perms_B, sh) = build_inputs(arch, meta)
    outs = []
    for use_wrapper in (False, True):
        random.seed(3)
        wrapped = ModelFactory(make_cfg(arch, meta["K"], m=meta["m"], T=meta["T"])).build_moco_diffloss(device=torch.device("cpu"))
        wrapped.module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
        wrapped.train()
        opt = SGD(wrapped.parameters(), lr=meta["lr"], momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
        crit = Loss(margin=meta["margin"], A=meta["A"], M=meta["M"])
        im_q, im_k = torch.from_numpy(clips[0][0]), torch.from_numpy(clips[0][1])
        stepper = GraphedPretextStep(wrapped, crit, opt)
        for _ in range(2):
            with ReplayRNG([perms_B[0], sh[0], sh[1]], meta["speed"]):
                if use_wrapper:
                    loss, loss_A, loss_M, out, rl = stepper(im_q, im_k)
                else:
                    out, tgt, rl, rt = wrapped(im_q, im_k)
                    loss, loss_A, loss_M = crit(out, tgt, rl, rt)
                    opt.zero_grad()
                    loss.backward()
                    opt.step()
        assert not stepper.graphs                                   # nothing to capture on the CPU
        outs.append((loss.detach().clone(), out[0].detach().clone(), {k: v.clone() for k, v in wrapped.module.state_dict().items()}))
    (l0, o0, s0), (l1, o1, s1) = outs
    assert torch.equal(l0, l1) and torch.equal(o0, o1)
    for k in s0:
        assert torch.equal(s0[k], s1[k]), k


def test_gate_fusion_table_and_the_unfused_path(cpu_backend, monkeypatch):
    """engine._gate_fusion pairs every sep_conv unit of S3D-G (models/s3dg.py:36-72) with its gate — and the two front-end ones
    with the max-pool behind them — and nothing else; with both fusions switched off (the A/B switches of DESIGN 5c) the step
    still reproduces the golden fixture, and the parameter gradients of the two ways agree to rounding."""
    from rspnet_amd import engine
    from rspnet_amd.models.s3dg import S3D_G
    plan = S3D_G().plan()
    table = engine._gate_fusion(plan)
    assert len(table) == 20 and sum(1 for v in table.values() if v[1] is not None) == 2
    for ci, (gi, pi) in table.items():
        conv, gate = plan.nodes[ci], plan.nodes[gi]
        assert isinstance(conv, engine.ConvBN) and isinstance(gate, engine.Gate) and gate.src == conv.dst and gi == ci + 1
        if pi is not None:
            assert isinstance(plan.nodes[pi], engine.Pool) and plan.nodes[pi].src == gate.dst and gate.into is None
    arch, _, seed = cases_for("s3dg", 1)[0]
    z, meta = load_case(arch, 1, seed)
    spec, inputs = build_inputs(arch, meta)
    _, _, _, grads_fused = run_model_step(arch, meta, inputs, 0, torch.device("cpu"), "fused")
    monkeypatch.setenv("RSP_NO_GATE_FUSION", "1")        # read when a plan's table is built: every model builds its own plan
    monkeypatch.setattr(engine, "GATE_BWD_FUSED", False)
    res, post, mom_post, grads = run_model_step(arch, meta, inputs, 0, torch.device("cpu"), "fused")
    compare_to_golden(z, 0, res, post, mom_post, tol=fwd_tol(arch, 2e-4), tol_grad=grad_tol(arch))
    for k, g in grads.items():
        if g is not None:
            ref = grads_fused[k]
            assert float(np.abs(g - ref).max()) <= 1e-5 * max(float(np.abs(ref).max()), 1e-6), k


def test_queue_pointer_must_be_a_multiple_of_the_global_batch():
    """A queue pointer that is not a multiple of the enqueue size (a checkpoint saved with another batch / world size) makes the
    reference's slice assignment fail (builder_diffspeed_diffloss.py:353-356).  The device-side enqueue of graph-captured steps would
    silently drop the slab instead, so the pointer is validated on the host once per loaded state."""
    import pytest
    from model_util import make_cfg
    from rspnet_amd.moco import ModelFactory
    m = ModelFactory(make_cfg("c3d", 64)).build_moco_diffloss(device=torch.device("cpu")).module
    m.queue_ptr.fill_(8)
    m._check_queue_ptr(4)
    assert m._ptr_checked
    m.queue_ptr.fill_(6)
    m._ptr_checked = False
    with pytest.raises(ValueError, match="not a multiple of the global batch"):
        m._check_queue_ptr(4)
    sd = m.state_dict()
    m._ptr_checked = True
    m.load_state_dict(sd)                       # a loaded state is unchecked again
    assert m._ptr_checked is False


def test_launcher_refuses_to_start_without_a_visible_gpu(monkeypatch):
    """pretrain's world size comes from a child interpreter's torch.cuda.device_count() (what a rank will see); zero GPUs is an
    error, not a silent single-rank run."""
    import pytest
    from rspnet_amd import pretrain
    with pytest.raises(EnvironmentError, match="no GPU is visible"):
        pretrain.visible_gpu_count()
