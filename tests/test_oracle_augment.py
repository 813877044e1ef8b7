"""CPU: the augmentation restatement (oracle/augment.py) against the fixtures generated from the reference's own transform
classes (oracle/gen_golden_augment.py -> tests/golden/augment.npz), and the product's host-side random draws against it."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment as A

GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment.npz"))
CASES = [tuple(int(v) for v in row) for row in GOLD["cases"]]


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"seed{c[0]}_{c[2]}x{c[3]}to{c[4]}")
def test_restatement_matches_reference_fixture(case):
    seed, T, h, w, size = case
    clip = A.synthetic_clip(seed, T, h, w)
    random.seed(seed)
    prm = A.draw_params()
    out = A.augment_clip(clip, size, prm, GOLD["mean"].tolist(), GOLD["std"].tolist())
    ref = torch.from_numpy(GOLD[f"out_{seed}"])
    assert out.shape == ref.shape == (3, T, size, size)
    assert (out - ref).abs().max().item() <= 1e-6          # bit-identical in the generating container


@pytest.mark.parametrize("seed", [int(v) for v in GOLD["plus_seeds"]])
def test_aug_plus_restatement_matches_reference_fixture(seed):
    _, T, h, w, size = CASES[0]
    clip = A.synthetic_clip(seed, T, h, w)
    random.seed(seed)
    out = A.augment_clip(clip, size, A.draw_params_plus(), GOLD["mean"].tolist(), GOLD["std"].tolist())
    assert (out - torch.from_numpy(GOLD[f"plus_{seed}"])).abs().max().item() <= 1e-6


def test_fixture_covers_every_branch():
    seen_ops, grays, flips = set(), 0, 0
    for seed, *_ in CASES:
        random.seed(seed)
        prm = A.draw_params()
        grays += prm.gray
        flips += prm.flip
        seen_ops.add(tuple(o for o, _ in prm.ops))
    assert grays >= 1 and flips >= 1 and len(seen_ops) >= 6     # distinct op orders


def test_product_draws_consume_random_like_the_reference_pipeline():
    from rspnet_amd.augment import FusedGPUCollateFn
    fn = FusedGPUCollateFn(16, [0.485, 0.456, 0.406], [0.229, 0.224, 0.225], device=torch.device("cpu"))
    for seed in range(20):
        random.seed(seed)
        want = [A.draw_params() for _ in range(3)]
        random.seed(seed)
        got = [fn.draw() for _ in range(3)]
        for w, (gray, flip, op_list) in zip(want, got):
            assert (int(w.gray), w.flip, w.ops) == (gray, flip, op_list)
    plus = FusedGPUCollateFn(16, [0, 0, 0], [1, 1, 1], aug_plus=True, device=torch.device("cpu"))
    for seed in range(30):
        random.seed(seed)
        w = A.draw_params_plus()
        random.seed(seed)
        gray, flip, op_list = plus.draw()
        assert gray == (2 if w.gray else 0) + (4 if w.blur else 0) and flip == w.flip and op_list == w.ops
    assert torch.equal(torch.tensor(plus.blur9).view(3, 3), A.gaussian_kernel2d())
    # a zero strength removes the op AND its random draw (ColorJitter._check_input, transforms_tensor.py:92-95)
    fn0 = FusedGPUCollateFn(16, [0, 0, 0], [1, 1, 1], hue=0, device=torch.device("cpu"))
    random.seed(3)
    want = A.draw_params(hue=0)
    random.seed(3)
    assert fn0.draw() == (int(want.gray), want.flip, want.ops) and len(want.ops) == 3
