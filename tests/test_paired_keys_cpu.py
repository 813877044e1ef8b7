"""CPU: the two key passes of a step (reference: moco/builder_diffspeed_diffloss.py:445 k_negative, :512 k) run as ONE forward over
their 2B clips (builder `_key_pass_pair`, engine.run_forward(pair=True)) give what two consecutive forwards give: the same key
features, logits, loss, gradients, queue, and the same BatchNorm running statistics of encoder_k."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cpu_ops import CpuOps
from golden_util import build_inputs, cases_for, compare_to_golden, fwd_tol, grad_tol, load_case, worst_grad_err
from model_util import make_cfg, run_model_step
from rspnet_amd import ops
from rspnet_amd.moco import ModelFactory


class SplitStatsOps(CpuOps):
    """The checker with the device's statistics layout in miniature: one tile of BatchNorm partials per sample, so that a
    paired forward takes the one-convolution-over-both-batches route and splits the tiles between the batches."""

    def stats_split_ok(self, g):
        return g.N % 2 == 0

    def conv_fwd(self, g, x, w_packed, bias, want_stats, out=None, out_ld=None, in_ld=None):
        y, _ = super().conv_fwd(g, x, w_packed, bias, False, out=out, out_ld=out_ld, in_ld=in_ld)
        stats = None
        if want_stats:
            y0 = (y if bias is None else y - bias.view(1, 1, 1, 1, -1)).double()
            stats = torch.stack([y0.sum(dim=(1, 2, 3)), (y0 * y0).sum(dim=(1, 2, 3))], dim=2).contiguous()
        return y, stats


@pytest.mark.parametrize("arch", ["c3d", "resnet18", "r2plus1d-vcop", "s3dg"])
def test_paired_step_is_the_step(arch, monkeypatch):
    a, ws, seed = cases_for(arch, 1)[0]
    z, meta = load_case(a, ws, seed)
    spec, inputs = build_inputs(a, meta)
    prev = ops.set_backend(CpuOps())
    try:
        monkeypatch.delenv("RSP_PAIR_KEYS", raising=False)
        ref = run_model_step(a, meta, inputs, 0, torch.device("cpu"), "fused")
        monkeypatch.setenv("RSP_PAIR_KEYS", "1")
        got = run_model_step(a, meta, inputs, 0, torch.device("cpu"), "fused")
    finally:
        ops.set_backend(prev)
    # on the checker a paired forward convolves batch by batch (its statistics come as one tile) and runs pools / heads over both:
    # the same step up to the rounding of a different batch blocking
    for k in ref[0]:
        assert float(np.abs(ref[0][k] - got[0][k]).max()) <= 2e-5 * max(1.0, float(np.abs(ref[0][k]).max())), k
    for k in ref[1]:
        if "encoder_k" in k or "queue" in k:
            r, g = ref[1][k].astype(np.float64), got[1][k].astype(np.float64)
            assert float(np.abs(r - g).max()) <= 2e-5 * max(1.0, float(np.abs(r).max())), k
    compare_to_golden(z, 0, got[0], got[1], got[2], tol=fwd_tol(a, 2e-4), tol_grad=grad_tol(a))


@pytest.mark.parametrize("arch", ["c3d", "resnet18", "r2plus1d-vcop", "s3dg"])
def test_one_convolution_over_both_batches(arch):
    """The route the device takes: convolutions over 2B rows, BatchNorm partials split between the batches."""
    torch.manual_seed(3)
    prev = ops.set_backend(SplitStatsOps())
    try:
        wrapped = ModelFactory(make_cfg(arch, 64)).build_moco_diffloss(device=torch.device("cpu"))
        model = wrapped.module
        model.train()
        with torch.no_grad():
            for m in model.encoder_k.modules():             # statistics that two different batches move differently
                if isinstance(m, torch.nn.modules.batchnorm._BatchNorm):
                    m.running_mean.normal_()
                    m.running_var.uniform_(0.5, 2.0)
        B, T, hw = 2, 16, (64 if arch == "s3dg" else 32)
        x = torch.randn(2 * B, T, hw, hw, 4)
        x[..., 3] = 0
        two, one = copy.deepcopy(model), copy.deepcopy(model)
        with torch.no_grad():
            d2 = two._deferred_k()
            a0, m0, _ = two.encoder_k.forward_ndhwc(x[:B].contiguous(), keep=False)
            a1, m1, _ = two.encoder_k.forward_ndhwc(x[B:].contiguous(), keep=False, deferred=d2)
            two._ema_k.run()
            d1 = one._deferred_k()
            a, m, _ = one.encoder_k.forward_ndhwc(x, keep=False, deferred=d1, pair=True)
            one._ema_k.run()
    finally:
        ops.set_backend(prev)
    for got, exp in ((a[:B], a0), (a[B:], a1), (m[:B], m0), (m[B:], m1)):
        assert float((got - exp).abs().max()) <= 1e-5 * max(1.0, float(exp.abs().max()))
    sd1, sd2 = one.encoder_k.state_dict(), two.encoder_k.state_dict()
    for k in sd2:
        if "running_" in k:
            assert float((sd1[k] - sd2[k]).abs().max()) <= 1e-5 * max(1.0, float(sd2[k].abs().max())), k
