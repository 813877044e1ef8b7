import pytest
import torch

from cpu_ops import CpuOps
from rspnet_amd import ops
from virtual_stem_util import CASES, check_case


@pytest.fixture()
def cpu_backend():
    prev = ops.set_backend(CpuOps())
    yield
    ops.set_backend(prev)


@pytest.mark.parametrize("case", CASES)
def test_virtual_pixel_stem_equals_padded_stem(cpu_backend, case):
    check_case(torch.device("cpu"), case, 2e-5)


def test_virtual_filters_hold_every_weight_once():
    from torch import nn
    from rspnet_amd import engine
    conv = nn.Conv3d(3, 5, (3, 7, 7), (1, 2, 2), (1, 3, 3), bias=False)
    node = engine.ConvBN(conv, nn.BatchNorm3d(5), 0, 1, (3, 7, 7), (1, 2, 2), (1, 3, 3), virtual_w=True)
    vs = engine.VirtualStem(node)
    for c in range(2):
        hv = vs.holders[c].conv.weight.view(-1)
        assert sorted(vs.dst[c].tolist()) == list(range(conv.weight.numel()))
        assert torch.equal(hv[vs.src[c]], conv.weight.data.view(-1)[vs.dst[c]])
        mask = torch.ones_like(hv, dtype=torch.bool)
        mask[vs.src[c]] = False
        assert float(hv[mask].abs().max()) == 0.0
