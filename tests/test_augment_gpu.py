"""GPU: fused augmentation kernel (rsp_augment_batch through rspnet_amd.augment.FusedGPUCollateFn) against the reference
fixtures and the CPU restatement.  Tolerance 5e-6 absolute on normalised outputs (|x| <= 2.7): every step follows the
reference's evaluation order exactly; only the contrast mean is summed in a different order."""
import os
import random

import numpy as np
import pytest
import torch

from oracle import augment as A

pytestmark = pytest.mark.gpu
GOLD = np.load(os.path.join(os.path.dirname(__file__), "golden", "augment.npz"))
CASES = [tuple(int(v) for v in row) for row in GOLD["cases"]]
MEAN, STD = GOLD["mean"].tolist(), GOLD["std"].tolist()
TOL = 5e-6


def collate(size, **kw):
    from rspnet_amd.augment import FusedGPUCollateFn
    return FusedGPUCollateFn(size, MEAN, STD, device=torch.device("cuda", 0), **kw)


@pytest.mark.parametrize("case", CASES, ids=lambda c: f"seed{c[0]}_{c[2]}x{c[3]}to{c[4]}")
def test_matches_reference_fixture(case):
    seed, T, h, w, size = case
    clip = A.synthetic_clip(seed, T, h, w)
    random.seed(seed)
    (out,), label, extra = collate(size)([([clip], 7, "x")])
    assert label.tolist() == [7] and extra == ("x",)
    err = (out[0].cpu() - torch.from_numpy(GOLD[f"out_{seed}"])).abs().max().item()
    assert err <= TOL, err


@pytest.mark.parametrize("seed", [int(v) for v in GOLD["plus_seeds"]])
def test_aug_plus_matches_reference_fixture(seed):
    _, T, h, w, size = CASES[0]
    clip = A.synthetic_clip(seed, T, h, w)
    random.seed(seed)
    (out,), _ = collate(size, aug_plus=True)([([clip], 0)])
    err = (out[0].cpu() - torch.from_numpy(GOLD[f"plus_{seed}"])).abs().max().item()
    assert err <= TOL, err


def test_full_size_batch_against_restatement():
    """B=3 samples x 2 clips, T=32, ragged crops -> 112 (the shipped pretext geometry); draws replayed into the oracle."""
    B, T, size = 3, 32, 112
    shapes = [(171, 128), (240, 320), (97, 211), (256, 340), (112, 112), (130, 260)]
    batch = [([A.synthetic_clip(10 + 2 * b + ci, T, *shapes[2 * b + ci]) for ci in range(2)], b) for b in range(B)]
    random.seed(123)
    clips, label, = collate(size)(batch)
    assert len(clips) == 2 and clips[0].shape == (B, 3, T, size, size) and label.tolist() == [0, 1, 2]
    random.seed(123)
    for b in range(B):
        for ci in range(2):
            want = A.augment_clip(batch[b][0][ci], size, A.draw_params(), MEAN, STD)
            err = (clips[ci][b].cpu() - want).abs().max().item()
            assert err <= TOL, (b, ci, err)


def test_flip_is_an_exact_mirror_and_identity_pipeline_is_resize_plus_normalise():
    clip = A.synthetic_clip(5, 4, 37, 53)
    plain = collate(24, p_gray=0.0, brightness=0, contrast=0, saturation=0, hue=0, p_flip=0.0)
    flipped = collate(24, p_gray=0.0, brightness=0, contrast=0, saturation=0, hue=0, p_flip=1.0)
    (a,), _ = plain([([clip], 0)])
    (b,), _ = flipped([([clip], 0)])
    assert torch.equal(a.flip(-1), b)
    want = A.augment_clip(clip, 24, A.ClipParams(), MEAN, STD)
    assert (a[0].cpu() - want).abs().max().item() <= TOL     # bilinear + normalise alone
    (a2,), _ = plain([([clip], 0)])
    assert torch.equal(a, a2)                                # run-to-run deterministic


def test_rejects_wrong_input():
    fn = collate(16)
    with pytest.raises(ValueError):
        fn([([torch.zeros(4, 8, 8, 3)], 0)])                 # float clip
    with pytest.raises(ValueError):
        fn([([torch.zeros(4, 8, 8, 4, dtype=torch.uint8)], 0)])
