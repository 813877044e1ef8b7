"""Generate tests/golden/augment.npz by running the REFERENCE's augmentation classes (build container only).
TEST INFRASTRUCTURE ONLY.

The reference modules datasets/transforms_video/{functional_tensor,transforms_tensor,transforms_spatial}.py are imported
from /root/reference.  They import ``torchvision`` (absent here; requirements.txt pins 0.7.0) for Compose / RandomApply and
the three ``_transforms_video`` classes, so a stub module carrying the published definitions of exactly those names is
installed first (oracle/augment.py header).  Everything else -- Resize, RandomGrayScale, ColorJitter and the functional
colour math -- is the reference's own code.  For each case: random.seed(seed) -> gpu_transform(clip) -> output; the same
seed drives oracle.augment.draw_params in the tests.

    python -m oracle.gen_golden_augment
"""
import os
import random
import sys
import types

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REFERENCE_ROOT = "/root/reference"
MEAN, STD = [0.485, 0.456, 0.406], [0.229, 0.224, 0.225]       # config/dataset/normalization.libsonnet:3-4
# (seed, T, h, w, size)
CASES = [(1, 4, 20, 26, 16), (2, 3, 31, 17, 16), (3, 4, 16, 16, 16), (4, 2, 40, 56, 24), (5, 4, 12, 9, 16), (6, 3, 24, 24, 12),
         (7, 2, 33, 47, 20), (8, 4, 18, 30, 16)]


def install_torchvision_stub():
    tv = types.ModuleType("torchvision")
    tr = types.ModuleType("torchvision.transforms")
    tvid = types.ModuleType("torchvision.transforms._transforms_video")

    class Compose:
        def __init__(self, transforms):
            self.transforms = transforms

        def __call__(self, x):
            for t in self.transforms:
                x = t(x)
            return x

    class RandomApply:
        def __init__(self, transforms, p=0.5):
            self.transforms, self.p = transforms, p

        def __call__(self, x):
            if self.p < random.random():
                return x
            for t in self.transforms:
                x = t(x)
            return x

    class ToTensorVideo:
        def __call__(self, clip):
            return clip.float().permute(3, 0, 1, 2) / 255.0

    class RandomHorizontalFlipVideo:
        def __init__(self, p=0.5):
            self.p = p

        def __call__(self, clip):
            if random.random() < self.p:
                clip = clip.flip(-1)
            return clip

    class NormalizeVideo:
        def __init__(self, mean, std, inplace=False):
            self.mean, self.std = mean, std

        def __call__(self, clip):
            m = torch.as_tensor(self.mean, dtype=clip.dtype)
            s = torch.as_tensor(self.std, dtype=clip.dtype)
            return (clip - m[:, None, None, None]) / s[:, None, None, None]

    tr.Compose, tr.RandomApply = Compose, RandomApply
    tvid.ToTensorVideo, tvid.RandomHorizontalFlipVideo, tvid.NormalizeVideo = ToTensorVideo, RandomHorizontalFlipVideo, NormalizeVideo
    tfun = types.ModuleType("torchvision.transforms.functional")    # imported by functional_tensor.py:7, never used there
    tv.__path__, tr.__path__ = [], []
    tv.transforms = tr
    tr._transforms_video, tr.functional = tvid, tfun
    sys.modules.update({"torchvision": tv, "torchvision.transforms": tr, "torchvision.transforms._transforms_video": tvid,
                        "torchvision.transforms.functional": tfun})


def reference_gpu_transform(size):
    """The `not aug_plus` branch of datasets/classification/__init__.py:189-202, built from the reference's classes."""
    install_torchvision_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from datasets.transforms_video import transforms_spatial, transforms_tensor
    return transforms_tensor.Compose([
        transforms_spatial.ToTensor(),
        transforms_spatial.Resize(size),
        transforms_spatial.RandomGrayScale(p=0.2),
        transforms_spatial.ColorJitter(brightness=0.4, contrast=0.4, saturation=0.4, hue=0.4),
        transforms_spatial.RandomHorizontalFlip(),
        transforms_spatial.Normalize(MEAN, STD, inplace=True),
    ])


def reference_gpu_transform_plus(size):
    """The `aug_plus` branch (datasets/classification/__init__.py:203-218) from the reference's classes."""
    install_torchvision_stub()
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    from datasets.transforms_video import transforms_spatial, transforms_tensor
    from torchvision.transforms import RandomApply
    return transforms_tensor.Compose([
        transforms_spatial.ToTensor(),
        transforms_spatial.Resize(size),
        RandomApply([transforms_spatial.ColorJitter(0.4, 0.4, 0.4, 0.1)], p=0.8),
        transforms_spatial.RandomGrayScale(p=0.2),
        RandomApply([transforms_tensor.GaussianBlur((3, 3), (1.5, 1.5))], p=0.5),
        transforms_spatial.RandomHorizontalFlip(),
        transforms_spatial.Normalize(MEAN, STD, inplace=True),
    ])


PLUS_SEEDS = [11, 12, 13, 14, 15, 16, 17, 18, 19, 20]


def main():
    sys.path.insert(0, ROOT)
    from oracle import augment as A
    out = {"cases": np.array(CASES, dtype=np.int64), "mean": np.array(MEAN, np.float32), "std": np.array(STD, np.float32)}
    worst = 0.0
    for seed, T, h, w, size in CASES:
        clip = A.synthetic_clip(seed, T, h, w)
        tf = reference_gpu_transform(size)
        random.seed(seed)
        ref = tf(clip.clone())
        random.seed(seed)
        prm = A.draw_params()
        mine = A.augment_clip(clip, size, prm, MEAN, STD)
        err = (ref - mine).abs().max().item()
        worst = max(worst, err)
        print(f"seed {seed}: gray={prm.gray} flip={prm.flip} ops={[(o, round(f, 3)) for o, f in prm.ops]} |ref - restatement| = {err:.2e}")
        out[f"out_{seed}"] = ref.numpy().astype(np.float32)
    assert worst <= 2e-6, worst
    # aug_plus chain on the geometry of the first case
    _, T, h, w, size = CASES[0]
    out["plus_seeds"] = np.array(PLUS_SEEDS, dtype=np.int64)
    for seed in PLUS_SEEDS:
        clip = A.synthetic_clip(seed, T, h, w)
        tf = reference_gpu_transform_plus(size)
        random.seed(seed)
        ref = tf(clip.clone())
        random.seed(seed)
        prm = A.draw_params_plus()
        mine = A.augment_clip(clip, size, prm, MEAN, STD)
        err = (ref - mine).abs().max().item()
        print(f"plus seed {seed}: gray={prm.gray} blur={prm.blur} flip={prm.flip} n_ops={len(prm.ops)} |ref - restatement| = {err:.2e}")
        assert err <= 2e-6, err
        out[f"plus_{seed}"] = ref.numpy().astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "augment.npz"), **out)
    print("wrote tests/golden/augment.npz; worst restatement error", worst)


if __name__ == "__main__":
    main()
