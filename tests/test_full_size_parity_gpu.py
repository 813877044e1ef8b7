"""GPU: value parity AT THE MEASURED SIZE.  One pretext step of every BASELINE.json workload (configs 2-5: C3D / R3D-18 /
R(2+1)D at B=32, 3x16x112x112; S3D-G at B=16, 3x16x224x224; K=16384, dim=128) through the HIP kernels, and the same step on the
oracle restatement (live, on this host's cores: ~20-60 s each) from the SAME seeded state, clips and injected permutations.
Bar (BASELINE.json north_star): loss / logits / features / queue within 1e-3 relative; gradient-derived tensors at the
per-architecture whole-step gate (tests/golden_util.py:grad_tol — ReLU / arg-max flips bound it).
Reference: /root/reference/moco/builder_diffspeed_diffloss.py:492-547, /root/reference/pretrain.py:157-165.

The full-size-only code paths this pins by value: BN reductions over 6.4 M elements per channel, 25 088-tile launches, the
single-LDS-buffer mode (>= 768 tiles), 49-round grids, 512-tile split-K tails, the K=16384 logits inside a real step."""
import pytest
import torch

from full_size_util import FWD_KEYS, step_vs_oracle
from golden_util import grad_tol

pytestmark = pytest.mark.gpu
TOL = 1e-3
# BASELINE.json configs 2-5: (arch, clips per GPU, H = W)
WORKLOADS = [("c3d", 32, 112), ("resnet18", 32, 112), ("r2plus1d-vcop", 32, 112), ("s3dg", 16, 224)]


@pytest.mark.parametrize("arch,B,HW", WORKLOADS, ids=[w[0] for w in WORKLOADS])
def test_full_size_step_matches_oracle(arch, B, HW):
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    errs, detail = step_vs_oracle(arch, B, HW, 16384, seed=1, device=torch.device("cuda", 0))
    print(f"\n{arch} B={B} {HW}x{HW} K=16384 vs oracle: " + ", ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    print("   worst tensors:", detail)
    for k in FWD_KEYS + ("queue", "bn_running_stats", "encoder_k_params"):
        assert errs[k] <= TOL, (k, errs[k], detail.get(k))
    # The floor of a whole gradient is a property of the STATE (tests/golden_util.py): at the measured size it is known for the
    # headline workload only — C3D: the oracle's fp32 gradient is 2.41e-3 from its fp64 gradient (profiles/grad_floor.json, a 150 s
    # fp64 replay), so the whole gradient is held to three floors = 7.2e-3 (measured 5.1e-3: tools/grad_census.py explains the
    # distance); the other backbones and every single-tensor figure keep 2e-2 / their fixture family's gate, whichever is larger.
    gt = max(grad_tol(arch), 2e-2)
    whole = 7.2e-3 if arch == "c3d" else gt
    assert errs["grad_whole"] <= whole, ("grad_whole", errs["grad_whole"])
    for k in ("grad_worst_tensor", "momentum_post", "encoder_q_params_post"):
        assert errs[k] <= gt, (k, errs[k], detail.get(k))


@pytest.mark.parametrize("arch,B,HW,K", [("c3d", 5, 32, 60), ("resnet18", 3, 64, 63), ("s3dg", 1, 64, 64)], ids=["c3d_B5", "resnet18_B3", "s3dg_B1"])
def test_odd_and_single_clip_batches_match_oracle(arch, B, HW, K):
    """Edge cases of the batch: an ODD batch (int(B * alpha) clips keep their speed, the rest are sub-sampled: unequal halves,
    /root/reference/moco/builder_diffspeed_diffloss.py:421-431; the restatement is pinned to the reference for B = 5 in
    tests/test_oracle_vs_reference.py) and a batch of ONE clip (int(0.5) = 0: every clip sub-sampled; BatchNorm statistics over one
    sample; a queue that takes one key per step), against the oracle on the same seeded inputs."""
    from rspnet_amd import ops
    assert ops.backend().name == "hip"
    errs, detail = step_vs_oracle(arch, B, HW, K, seed=2, device=torch.device("cuda", 0))
    print(f"\n{arch} B={B} {HW}x{HW} K={K} vs oracle: " + ", ".join(f"{k}={v:.2e}" for k, v in errs.items()))
    for k in FWD_KEYS + ("queue", "bn_running_stats", "encoder_k_params"):
        assert errs[k] <= TOL, (k, errs[k], detail.get(k))
    # (an unguarded random state with 8-48 positions per channel in its last layers: one flipped ReLU mask moves a small tensor by
    #  percents — tests/golden_util.py — so the gradient is held as a whole vector only)
    assert errs["grad_whole"] <= 5e-2, ("grad_whole", errs["grad_whole"])
