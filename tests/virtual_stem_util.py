"""One 3-channel stride-2 stem unit (conv + BatchNorm + ReLU) through the engine twice — as the ordinary 4-channel-padded
convolution and on virtual pixels (rspnet_amd/engine.py VirtualStem) — on whatever backend is installed.  The two must agree
to rounding: same products, different summation order."""
import torch
from torch import nn

from rspnet_amd import engine


def run_unit(dev, virtual, cout, cout_pad, k, s, p, shape, seed=0):
    N, T, H, W = shape
    gen = torch.Generator().manual_seed(seed)
    conv = nn.Conv3d(3, cout, k, s, p, bias=False)
    bn = nn.BatchNorm3d(cout)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=gen) * 0.1)
        bn.weight.copy_(torch.rand(cout, generator=gen) + 0.5)
        bn.bias.copy_(torch.randn(cout, generator=gen) * 0.1)
    conv, bn = conv.to(dev), bn.to(dev)
    x = torch.zeros(N, T, H, W, 4)
    x[..., :3] = torch.randn(N, T, H, W, 3, generator=gen)
    node = engine.ConvBN(conv, bn, 0, 1, k, s, p, relu=True, cout_pad=cout_pad, virtual_w=virtual)
    plan = engine.Plan([node], input_slot=0, output_slot=1)
    packed = engine.PackedWeights()
    out, ctx = engine.run_forward(plan, x.to(dev), packed, keep=True, training=True)
    assert (len(packed._virtual) == 1) == virtual
    dout = torch.randn(out.shape, generator=gen).to(dev)
    grads = {id(q): torch.full_like(q, float("nan")) for q in (conv.weight, bn.weight, bn.bias)}
    engine.run_backward(plan, ctx, dout, lambda q: grads[id(q)])
    res = {"out": out, "dw": grads[id(conv.weight)], "dgamma": grads[id(bn.weight)], "dbeta": grads[id(bn.bias)],
           "running_mean": bn.running_mean, "running_var": bn.running_var}
    # second step with changed weights: the derived filters must follow the parameter
    with torch.no_grad():
        conv.weight.mul_(-0.5)
    packed.invalidate()
    out2, _ = engine.run_forward(plan, x.to(dev), packed, keep=False, training=True)
    res["out_after_update"] = out2
    return {n: t.detach().cpu() for n, t in res.items()}


CASES = [  # cout, cout_pad, k, s, p, (N, T, H, W)
    (64, 0, (7, 7, 7), (1, 2, 2), (3, 3, 3), (2, 4, 16, 16)),        # R3D-18 conv1 (models/resnet.py:124)
    (45, 48, (1, 7, 7), (1, 2, 2), (0, 3, 3), (2, 3, 12, 24)),       # R(2+1)D stem, spatial half, odd mid-channel count
    (16, 0, (3, 7, 7), (2, 2, 2), (1, 3, 3), (1, 5, 10, 8)),
]


def check_case(dev, case, tol):
    cout, cp, k, s, p, shape = case
    a = run_unit(dev, False, cout, cp, k, s, p, shape)
    b = run_unit(dev, True, cout, cp, k, s, p, shape)
    for n in a:
        scale = float(a[n].abs().max()) + 1e-12
        err = float((a[n] - b[n]).abs().max()) / scale
        assert err < tol, (n, err)
