"""Fused GPU augmentation of the pretext data path — the collate function the reference builds in
datasets/classification/__init__.py:146-202 (`SequentialGPUCollateFn(gpu_transform)`, transforms_tensor.py:207-233), with the
default (`moco.aug_plus = false`) per-clip transform chain

    ToTensor -> Resize(size) -> RandomGrayScale(0.2) -> ColorJitter(0.4, 0.4, 0.4, 0.4) -> RandomHorizontalFlip -> Normalize

(and, with aug_plus=True, the `moco.aug_plus` chain with RandomApply'd jitter and 3x3 Gaussian blur) executed by ONE batched HIP launch group (`rsp_augment_batch`, rspnet_amd/csrc/augment.hip) instead of ~25 small ATen
launches per clip in a Python loop.  The random decisions are drawn on the host from Python's `random` in exactly the
reference's order (clip by clip: gray test, the four uniform factors, the shuffle of the op list, flip test), so a seeded run
reproduces the reference's augmentations.  Same call contract as the reference collate: a list of
`([clip_0, clip_1, ...], label, *others)` samples with uint8 (T,h,w,3) clips already cropped on the CPU
(RawVideoRandomCrop) -> `([ (B,3,T,size,size) float32 per clip index ], label_tensor, *others)`.
"""
from __future__ import annotations

import ctypes as C
import random
from typing import List, Optional, Sequence

import torch

from . import _lib, ops

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


def _jitter_range(value, center=1.0, clip_first_on_zero=True):
    """ColorJitter._check_input (transforms_tensor.py:76-95): a number v -> [center - v, center + v]; None when it is a no-op."""
    if isinstance(value, (tuple, list)):
        lo, hi = float(value[0]), float(value[1])
    else:
        if value < 0:
            raise ValueError("If jitter strength is a single number, it must be non negative.")
        lo, hi = center - value, center + value
        if clip_first_on_zero:
            lo = max(lo, 0)
    return None if lo == hi == center else (lo, hi)


def _gaussian_kernel2d(ksize=(3, 3), sigma=(1.5, 1.5)) -> torch.Tensor:
    """The 3x3 kernel GaussianBlur registers (transforms_tensor.py:168-176, functional_tensor.py:420-500), in float32."""
    def win(n, s):
        g = torch.stack([torch.exp(torch.tensor(-(x - n // 2) ** 2 / float(2 * s ** 2))) for x in range(n)])
        return g / g.sum()
    return torch.matmul(win(ksize[0], sigma[0]).unsqueeze(-1), win(ksize[1], sigma[1]).unsqueeze(-1).t())


class FusedGPUCollateFn:
    def __init__(self, size: int, mean: Sequence[float], std: Sequence[float], p_gray: float = 0.2, brightness=0.4, contrast=0.4,
                 saturation=0.4, hue=None, p_flip: float = 0.5, target_transform: bool = True,
                 device: Optional[torch.device] = None, aug_plus: bool = False, p_jitter: float = 0.8, p_blur: float = 0.5):
        """aug_plus=False: the default chain above.  aug_plus=True (`moco.aug_plus`, datasets/classification/__init__.py:203-218):
        RandomApply([ColorJitter(.4,.4,.4,.1)], 0.8) -> RandomGrayScale(0.2) -> RandomApply([GaussianBlur((3,3),(1.5,1.5))], 0.5)
        -> flip -> normalise.  `hue` defaults to the chain's own value (0.4 / 0.1)."""
        self.aug_plus, self.p_jitter, self.p_blur = bool(aug_plus), p_jitter, p_blur
        if hue is None:
            hue = 0.1 if aug_plus else 0.4
        self.blur9 = _gaussian_kernel2d().reshape(-1).tolist() if aug_plus else None
        self.size = int(size)
        self.mean, self.std = [float(v) for v in mean], [float(v) for v in std]
        self.p_gray, self.p_flip = p_gray, p_flip
        self.brightness = _jitter_range(brightness)
        self.contrast = _jitter_range(contrast)
        self.saturation = _jitter_range(saturation)
        self.hue = _jitter_range(hue, center=0.0, clip_first_on_zero=False)
        self.target_transform = target_transform
        self.device = device or torch.device("cuda", torch.cuda.current_device())

    def _draw_jitter(self):
        op_list = []
        if self.brightness is not None:
            op_list.append((BRIGHTNESS, random.uniform(*self.brightness)))
        if self.contrast is not None:
            op_list.append((CONTRAST, random.uniform(*self.contrast)))
        if self.saturation is not None:
            op_list.append((SATURATION, random.uniform(*self.saturation)))
        if self.hue is not None:
            op_list.append((HUE, random.uniform(*self.hue)))
        random.shuffle(op_list)
        return op_list

    def draw(self):
        """One clip's random decisions, consuming `random` as the reference's Compose does (transforms_tensor.py:29,107-127;
        torchvision's RandomApply skips its transforms when ``p < random.random()``).  Returns (gray code, flip, ops):
        gray code = rsp_augment_clip_desc.gray (1 grayscale before the ops, 2 after, +4 blur)."""
        if not self.aug_plus:
            gray = int(random.random() < self.p_gray)
            op_list = self._draw_jitter()
            flip = random.random() < self.p_flip
            return gray, flip, op_list
        op_list = [] if self.p_jitter < random.random() else self._draw_jitter()
        gray = 2 if random.random() < self.p_gray else 0
        if not (self.p_blur < random.random()):
            gray += 4
        flip = random.random() < self.p_flip
        return gray, flip, op_list

    def __call__(self, batch):
        clips, label, *others = zip(*batch)
        label_tensor = None
        if self.target_transform:
            label_tensor = torch.as_tensor(label).to(self.device, non_blocking=True)
        B, num_clips = len(clips), len(clips[0])
        T = int(clips[0][0].shape[0])
        # one pinned staging buffer + one H2D copy for all crops (16-byte aligned starts)
        offs, total = [], 0
        for sample in clips:
            for clip in sample:
                if clip.dtype != torch.uint8 or clip.dim() != 4 or clip.shape[3] != 3 or clip.shape[0] != T:
                    raise ValueError(f"expected uint8 (T={T},h,w,3) clips, got {clip.dtype} {tuple(clip.shape)}")
                offs.append(total)
                total += (clip.numel() + 15) // 16 * 16
        stage = torch.empty(total, dtype=torch.uint8, pin_memory=True)
        descs = (_lib.AugmentClipDesc * (B * num_clips))()
        k = 0
        for b, sample in enumerate(clips):                      # reference order: batch index outer, clip index inner
            for ci, clip in enumerate(sample):
                h, w = int(clip.shape[1]), int(clip.shape[2])
                stage[offs[k]:offs[k] + clip.numel()] = clip.contiguous().view(-1)
                gray, flip, op_list = self.draw()
                d = descs[ci * B + b]
                d.src = offs[k]                                  # rebased to the device buffer below
                d.frame_pitch, d.row_pitch, d.h, d.w = h * w * 3, w * 3, h, w
                d.gray, d.flip, d.n_ops = int(gray), int(flip), len(op_list)
                for i, (op, f) in enumerate(op_list):
                    d.op[i], d.factor[i], d.one_minus[i] = op, f, 1.0 - f
                k += 1
        src = stage.to(self.device, non_blocking=True)
        base = src.data_ptr()
        for d in descs:
            d.src = base + (d.src or 0)
        dbytes = torch.frombuffer(bytearray(bytes(descs)), dtype=torch.uint8).pin_memory()
        ddev = dbytes.to(self.device, non_blocking=True)
        out = torch.empty((num_clips, B, 3, T, self.size, self.size), dtype=torch.float32, device=self.device)
        ops.backend().augment_batch(ddev, B * num_clips, T, self.size, self.mean, self.std, out.view(-1, 3, T, self.size, self.size),
                                    blur9=self.blur9)
        # `src` / `ddev` die here, but the caching allocator is stream-ordered: their blocks are only re-issued to work queued
        # behind these kernels on the same stream
        return ([out[i] for i in range(num_clips)], label_tensor, *others)
