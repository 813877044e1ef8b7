"""CPU: rsp_conv3d_executed_fraction — the host replica of the kernels' padding-skip walks that bench.py prices `executed_frac` with
(no GPU call: planning arithmetic only).  Reference FLOP convention: nn.Conv3d as cuDNN / oneDNN count it, padded taps included
(models/c3d.py:21-52); the kernels skip the K chunks / row chunks that are zero padding for a whole tile."""
import ctypes as C

import pytest

from rspnet_amd import _lib
from rspnet_amd.ops import ConvGeom


def frac(g, which):
    d = g.desc()
    return float(_lib.load().rsp_conv3d_executed_fraction(C.byref(d), which))


def test_no_padding_means_everything_is_executed():
    g = ConvGeom(8, 8, 28, 28, 64, 128, (1, 1, 1), (1, 1, 1), (0, 0, 0))          # pointwise
    assert [frac(g, w) for w in (0, 1, 2)] == [1.0, 1.0, 1.0]
    g = ConvGeom(8, 8, 28, 28, 64, 128, (1, 3, 3), (1, 1, 1), (0, 1, 1))          # spatial padding only: never dead for a whole tile
    assert frac(g, 0) == 1.0 and frac(g, 1) == 1.0


@pytest.mark.parametrize("T,HW,cin,cout", [(8, 28, 256, 256), (4, 14, 512, 512), (2, 7, 512, 512)])
def test_c3d_depth_padding_share(T, HW, cin, cout):
    """3x3x3, padding 1 over T frames: 2/(3T) of the multiply-adds read the depth padding; depth-major rows make those taps dead for
    (nearly) every tile of the first / last frame, so the executed share sits just above 1 - 2/(3T)."""
    g = ConvGeom(32, T, HW, HW, cin, cout, (3, 3, 3), (1, 1, 1), (1, 1, 1))
    ideal = 1.0 - 2.0 / (3 * T)
    for which in (0, 1):
        f = frac(g, which)
        assert ideal - 1e-9 <= f <= ideal + 0.03, (which, f, ideal)
    assert ideal - 1e-9 <= frac(g, 2) <= 1.0


def test_r3d_last_stage_runs_one_depth_tap_of_three():
    g = ConvGeom(32, 1, 4, 4, 512, 512, (3, 3, 3), (1, 1, 1), (1, 1, 1))          # T = 1: two of three depth taps are padding everywhere
    assert abs(frac(g, 0) - 1.0 / 3.0) < 1e-9 and abs(frac(g, 1) - 1.0 / 3.0) < 1e-9


def test_fraction_is_a_share():
    for g in (ConvGeom(32, 16, 56, 56, 64, 128, (3, 3, 3), (1, 1, 1), (1, 1, 1)), ConvGeom(32, 16, 112, 112, 4, 64, (7, 7, 7), (1, 2, 2), (3, 3, 3)),
              ConvGeom(16, 8, 112, 112, 64, 64, (7, 1, 1), (1, 1, 1), (3, 0, 0)), ConvGeom(32, 16, 56, 56, 144, 64, (3, 1, 1), (2, 1, 1), (1, 0, 0))):
        for which in (0, 1, 2):
            assert 0.3 < frac(g, which) <= 1.0
