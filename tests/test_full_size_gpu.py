"""GPU: BASELINE-size checks (C3D, B=32 clips of 3x16x112x112, K=16384) through size-independent properties — the oracle
needs ~70 s per step at this size, so instead of a value comparison the kernels are held to identities that any correct
implementation satisfies at any size:

  * adjointness  <conv(x), dy> = <x, dgrad(dy)> = <w, wgrad(x, dy)>      (forward / dgrad / wgrad are one bilinear form)
  * linearity    conv(a x1 + b x2) = a conv(x1) + b conv(x2)
  * train-mode BN output has per-channel mean 0 / variance 1 (before the affine), running stats move by the batch moments
  * the step's bookkeeping: momentum update and SGD obey their formulas element-wise, the queue slab holds unit-norm keys,
    the pointer advances by B, |logits| <= 1/T, and the whole step is run-to-run deterministic.
"""
import pytest
import torch

from rspnet_amd.ops import ConvGeom, PoolGeom

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda", 0)
B = 32


@pytest.fixture(scope="module")
def hip():
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    return ops.backend()


def ddot(a, b):
    return float((a.double() * b.double()).sum())


# (name, batch, T, HW, Cin, Cout, k, s, p): C3D layers at B=32 (conv1 runs on the 4-channel padded clip = stem
# kernel) and the distinct conv shapes of the other backbones (§A.2-A.4): strided / factored / pointwise, stems via both paths
K3, S1, P1 = (3, 3, 3), (1, 1, 1), (1, 1, 1)
LAYERS = [
    ("c3d-conv1", 32, 16, 112, 4, 64, K3, S1, P1), ("c3d-conv2", 32, 16, 56, 64, 128, K3, S1, P1),
    ("c3d-conv3b", 32, 8, 28, 256, 256, K3, S1, P1), ("c3d-conv4b", 32, 4, 14, 512, 512, K3, S1, P1),
    ("c3d-conv5b", 32, 2, 7, 512, 512, K3, S1, P1),
    ("r3d-stem", 32, 16, 112, 4, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),            # 343 taps: implicit-GEMM path
    ("r3d-layer2-s2", 32, 8, 28, 64, 128, K3, (2, 2, 2), P1),                    # 8 dgrad parity classes
    ("r3d-downsample", 32, 8, 28, 64, 128, (1, 1, 1), (2, 2, 2), (0, 0, 0)),     # positions without gradient
    ("r21d-spatial", 32, 16, 56, 64, 144, (1, 3, 3), S1, (0, 1, 1)),             # factored pair, mid channels padded to x4
    ("r21d-temporal", 32, 16, 56, 144, 64, (3, 1, 1), S1, (1, 0, 0)),
    ("s3dg-stem", 16, 16, 224, 4, 64, (1, 7, 7), (1, 2, 2), (0, 3, 3)),           # 49 taps, stride 2: stem kernel
    ("s3dg-pointwise", 16, 8, 28, 192, 96, (1, 1, 1), S1, (0, 0, 0)),
]


@pytest.mark.parametrize("layer", LAYERS, ids=lambda l: l[0])
def test_conv_adjoint_identities_at_full_size(hip, layer):
    name, Bn, T, HW, cin, cout, k, st, pd = layer
    g = ConvGeom(Bn, T, HW, HW, cin, cout, k, st, pd)
    gen = torch.Generator(device=DEV).manual_seed(sum(map(ord, name)))
    x = torch.randn(Bn, T, HW, HW, cin, device=DEV, generator=gen)
    w = torch.randn(cout, cin, *k, device=DEV, generator=gen) * (k[0] * k[1] * k[2] * cin) ** -0.5
    dy = torch.randn(Bn, *g.out_dims, cout, device=DEV, generator=gen)
    y, _ = hip.conv_fwd(g, x, hip.conv_pack_fwd(g, w), None, False)
    lhs = ddot(y, dy)
    scale = (float(y.double().pow(2).sum()) * float(dy.double().pow(2).sum())) ** 0.5     # Cauchy-Schwarz bound of |<y,dy>|
    dw = torch.empty_like(w)
    hip.conv_wgrad(g, x, dy, dw)
    assert abs(ddot(w, dw) - lhs) <= 2e-6 * scale, (name, "wgrad", lhs, ddot(w, dw), scale)
    if cin > 4:
        dx = hip.conv_dgrad(g, dy, w)
        assert abs(ddot(x, dx) - lhs) <= 2e-6 * scale, (name, "dgrad", lhs, ddot(x, dx), scale)


def test_conv_linearity_and_stat_partials_at_full_size(hip):
    T, HW, cin, cout = 16, 56, 64, 128                                    # conv2: the largest GEMM of the step
    g = ConvGeom(B, T, HW, HW, cin, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    gen = torch.Generator(device=DEV).manual_seed(7)
    x1 = torch.randn(B, T, HW, HW, cin, device=DEV, generator=gen)
    x2 = torch.randn(B, T, HW, HW, cin, device=DEV, generator=gen)
    w = torch.randn(cout, cin, 3, 3, 3, device=DEV, generator=gen) * (27 * cin) ** -0.5
    wp = hip.conv_pack_fwd(g, w)
    y1, st1 = hip.conv_fwd(g, x1, wp, None, True)
    y2, _ = hip.conv_fwd(g, x2, wp, None, False)
    y12, _ = hip.conv_fwd(g, 0.5 * x1 - 2.0 * x2, wp, None, False)
    err = float((y12 - (0.5 * y1 - 2.0 * y2)).abs().max())
    assert err <= 2e-5 * float(y12.abs().max()), err
    # the BN partials written by the epilogue are the column sums / sums of squares of y
    s = st1.double().sum(0)
    ref = torch.stack([y1.double().sum((0, 1, 2, 3)), y1.double().pow(2).sum((0, 1, 2, 3))], dim=1)
    assert float((s - ref).abs().max() / ref.abs().max()) <= 1e-6


def test_batchnorm_normalises_at_full_size(hip):
    T, HW, cin, C = 16, 56, 64, 128                                       # conv2 + bn2 of C3D
    rows = B * T * HW * HW
    g = ConvGeom(B, T, HW, HW, cin, C, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    gen = torch.Generator(device=DEV).manual_seed(3)
    x = torch.randn(B, T, HW, HW, cin, device=DEV, generator=gen) + 0.3
    w = torch.randn(C, cin, 3, 3, 3, device=DEV, generator=gen) * (27 * cin) ** -0.5
    y, st = hip.conv_fwd(g, x, hip.conv_pack_fwd(g, w), None, True)
    gamma, beta = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    mi, ss = hip.bn_finalize(st, rows, None, gamma, beta, 1e-5, 0.1, rm, rv)
    out = hip.bn_act_pool_fwd(PoolGeom(B, T, HW, HW, C), y, ss, None, False)
    m = out.double().mean((0, 1, 2, 3))
    v = out.double().var((0, 1, 2, 3), unbiased=False)
    assert float(m.abs().max()) <= 1e-5 and float((v - 1).abs().max()) <= 1e-4
    bm = y.double().mean((0, 1, 2, 3))
    bv = y.double().var((0, 1, 2, 3), unbiased=True)
    assert float((rm.double() - 0.1 * bm).abs().max()) <= 1e-6                      # nn.BatchNorm3d momentum 0.1 (models/c3d.py:22)
    assert float((rv.double() - (0.9 + 0.1 * bv)).abs().max()) <= 1e-5
    # ReLU + MaxPool3d(2,2,2) fused behind it: equals pooling the normalised tensor
    pooled = hip.bn_act_pool_fwd(PoolGeom(B, T, HW, HW, C, (2, 2, 2), (2, 2, 2)), y, ss, None, True)
    ref = out.clamp_min(0).view(B, T // 2, 2, HW // 2, 2, HW // 2, 2, C).amax((2, 4, 6))
    assert torch.equal(pooled, ref)


# BASELINE.json configs 1-4: (arch, clips per GPU, H = W)
WORKLOADS = [("c3d", 32, 112), ("resnet18", 32, 112), ("r2plus1d-vcop", 32, 112), ("s3dg", 16, 224)]


@pytest.mark.parametrize("arch,B,HW", WORKLOADS, ids=[w[0] for w in WORKLOADS])
def test_full_size_step_bookkeeping_and_determinism(arch, B, HW):
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD
    K, T_, m_, lr, wd = 16384, 0.07, 0.999, 0.05, 1e-4
    cfg = {"model": {"arch": arch}, "moco": {"dim": 128, "k": K, "m": m_, "t": T_, "fc_type": "linear", "diff_speed": [2]}}
    gen = torch.Generator(device=DEV).manual_seed(11)
    im_q = torch.randn(B, 3, 32, HW, HW, device=DEV, generator=gen)
    im_k = im_q + 0.1 * torch.randn(B, 3, 32, HW, HW, device=DEV, generator=gen)

    def one_step():
        import random
        torch.manual_seed(5)
        random.seed(5)
        wrapped = ModelFactory(cfg).build_moco_diffloss(device=DEV)
        model = wrapped.module
        model.train()
        opt = SGD(wrapped.parameters(), lr=lr, momentum=0.9, dampening=0.0, weight_decay=wd, nesterov=False)
        q0 = {n: p.detach().clone() for n, p in model.encoder_q.named_parameters()}
        k0 = {n: p.detach().clone() for n, p in model.encoder_k.named_parameters()}
        queue0 = model.queue.clone()
        out, tgt, rl, rt = wrapped(im_q, im_k)
        loss, la, lm = Loss(margin=2.0, A=1.0, M=1.0)(out, tgt, rl, rt)
        opt.zero_grad()
        loss.backward()
        grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.encoder_q.named_parameters()}
        opt.step()
        torch.cuda.synchronize()
        return model, out, rl, (loss, la, lm), q0, k0, queue0, grads

    model, out, rl, losses, q0, k0, queue0, grads = one_step()
    assert all(torch.isfinite(v).all() for v in (*out, *rl, *losses))
    assert out[0].shape == (B, 1 + K) and float(out[0].detach().abs().max()) <= 1 / T_ * (1 + 1e-5)     # unit-norm dot products / T
    assert float(losses[0]) == pytest.approx(float(losses[1]) + float(losses[2]), rel=1e-6)
    # momentum update (builder_diffspeed_diffloss.py:337-343) happened BEFORE the optimizer step, on every key parameter
    for n, p in model.encoder_k.named_parameters():
        want = k0[n] * m_ + q0[n] * (1 - m_)
        assert float((p - want).abs().max()) <= 1e-7 * max(1.0, float(want.abs().max())), n
    # first SGD step with zero momentum buffers: p <- p - lr (g + wd p); parameters without a gradient are untouched
    for n, p in model.encoder_q.named_parameters():
        if grads[n] is None:
            assert torch.equal(p, q0[n]), n
        else:
            want = q0[n] - lr * (grads[n] + wd * q0[n])
            assert float((p - want).abs().max()) <= 2e-7 * max(1.0, float(want.abs().max())), n
    # queue (:345-359): the first B columns are the new unit-norm keys, the rest is untouched, ptr advanced by B
    assert int(model.queue_ptr) == B
    assert float((model.queue[:, :B].norm(dim=0) - 1).abs().max()) <= 1e-5
    assert torch.equal(model.queue[:, B:], queue0[:, B:])
    # the whole step is reproducible bit for bit (fixed-order reductions everywhere, no atomics)
    model2, out2, rl2, losses2, *_ = one_step()
    assert torch.equal(out[0], out2[0]) and torch.equal(out[1], out2[1]) and torch.equal(losses[0], losses2[0])
    for (n, p), (_, p2) in zip(model.encoder_q.named_parameters(), model2.encoder_q.named_parameters()):
        assert torch.equal(p, p2), n
    # every BatchNorm of both encoders saw finite statistics; the key encoder ran twice, the query encoder once
    for n, b in model.named_buffers():
        if n.endswith(("running_mean", "running_var")):
            assert torch.isfinite(b).all(), n
        elif n.endswith("num_batches_tracked"):
            assert int(b) == (2 if n.startswith("encoder_k.") else 1), n


def test_offsets_beyond_32_bits_are_refused_or_handled(hip):
    """The LDS-DMA kernels address with 32-bit byte offsets.  S3D-G's largest activation at B=16, 224^2 (stem output
    16x8x112x112x64 fp32 = 411 MB) is far below 4 GiB, and the library routes anything larger to the 64-bit scalar-gather kernel
    instead of wrapping around: a conv whose input exceeds 4 GiB must still be right."""
    N, D, H, W, Cin, Cout = 1, 33, 1024, 1024, 32, 8          # 33 * 2^20 * 32 * 4 B = 4.4 GB input
    g = ConvGeom(N, D, H, W, Cin, Cout, (1, 1, 1), (1, 1, 1), (0, 0, 0))
    x = torch.zeros((N, D, H, W, Cin), device=DEV)
    x[0, D - 1, H - 1, W - 1, :] = 1.0                          # the very last position, beyond the 32-bit range
    x[0, 0, 0, 0, :] = 2.0
    w = torch.arange(Cout * Cin, device=DEV, dtype=torch.float32).view(Cout, Cin, 1, 1, 1) * 1e-3
    y, _ = hip.conv_fwd(g, x, hip.conv_pack_fwd(g, w), None, False)
    want_last = w.view(Cout, Cin).sum(1)
    assert float((y[0, D - 1, H - 1, W - 1] - want_last).abs().max()) <= 1e-5
    assert float((y[0, 0, 0, 0] - 2 * want_last).abs().max()) <= 1e-5
    assert float(y[0, D - 1, H - 1, W - 2].abs().max()) == 0.0
    del x, y
    torch.cuda.empty_cache()
