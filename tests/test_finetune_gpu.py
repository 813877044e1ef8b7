"""GPU: fine-tune forward (train and eval mode) / backward of MultiTaskWrapper(finetune=True) on the HIP kernels against the
fixtures generated from the reference; north-star tolerance 1e-3 on logits / loss, gradients at the per-backbone gate."""
import pytest
import torch

from finetune_util import ARCHS, check_case, load

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("arch", ARCHS)
def test_finetune_forward_backward_matches_fixture(arch):
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    worst = check_case(arch, torch.device("cuda", 0), 1e-3)
    print(f"\n{arch}: worst gradient summary error {worst:.2e}")


def test_factory_train_and_validate_steps():
    """models.ModelFactory.build_multitask_wrapper + finetune.train_step / validate_step (10-crop style averaging)."""
    from rspnet_amd.finetune import train_step, validate_step
    from rspnet_amd.models import ModelFactory
    z, meta, spec, state, x = load("c3d")
    wrapped = ModelFactory({"model": {"arch": "c3d"}, "dataset": {"num_classes": meta["classes"]}}).build_multitask_wrapper(0)
    wrapped.module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    dev = torch.device("cuda", 0)
    xt, tt = torch.from_numpy(x).to(dev), torch.from_numpy(z["target"]).to(dev)
    crit = torch.nn.CrossEntropyLoss()
    opt = torch.optim.SGD(wrapped.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
    wrapped.train()
    losses = [float(train_step(wrapped, crit, opt, xt, tt)["loss"]) for _ in range(8)]
    assert losses[-1] < losses[0]
    wrapped.eval()
    clip3 = torch.cat([xt, xt.flip(4), xt], dim=2)                     # 3 "crops" stacked along time (finetune.py:44-52)
    out = validate_step(wrapped, crit, clip3, tt, n_crop=3)
    single = [wrapped(c) for c in (xt, xt.flip(4), xt)]
    want = (single[0] + single[1] + single[2]) / 3
    assert out["output"].shape == (meta["B"], meta["classes"])
    assert float((out["output"] - want).abs().max()) <= 1e-4 * float(want.abs().max())


@pytest.mark.parametrize("arch", ["resnet18", "r2plus1d-vcop"])
def test_finetune_backward_with_the_lane_running_ahead_changes_no_bit(arch, monkeypatch):
    """The fine-tune backward runs through the same engine as the pretext step's: its weight gradients go to the task lane and, since
    round 6, run ahead of the trunk (engine.BranchStreams.side_task).  Every weight gradient sent aside and none too big to run ahead,
    against the join-before-every-task form: the same logits and parameter gradients, bit for bit."""
    import finetune_util as F
    from rspnet_amd.engine import BranchStreams
    monkeypatch.setattr(BranchStreams, "SMALL_WGRAD_FLOPS", 1e18)
    monkeypatch.setattr(BranchStreams, "AHEAD_MAX_FLOPS", 1e18)
    z, meta, spec, state, x = F.load(arch)
    dev = torch.device("cuda", 0)
    out = []
    for ahead in (True, False):
        monkeypatch.setattr(BranchStreams, "RUN_AHEAD", ahead)
        model = F.build_model(arch, meta["classes"], state, dev)
        model.train()
        logits = model(torch.from_numpy(x).to(dev))
        loss = torch.nn.CrossEntropyLoss()(logits, torch.from_numpy(z["target"]).to(dev))
        loss.backward()
        torch.cuda.synchronize()
        out.append((logits.detach().clone(), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}))
    (la, ga), (lb, gb) = out
    assert torch.equal(la, lb) and ga.keys() == gb.keys() and len(ga) > 10
    for n in ga:
        assert torch.equal(ga[n], gb[n]), n
