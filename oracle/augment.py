"""CPU restatement of the reference's per-clip GPU augmentation (pretext path; the default chain and the ``moco.aug_plus``
chain, datasets/classification/__init__.py:189-218).
TEST INFRASTRUCTURE ONLY: imported by tests/, never by the product.

Pipeline restated (datasets/classification/__init__.py:189-202, applied per clip by
transforms_tensor.py:SequentialGPUCollateFn :207-233 after the CPU-side RawVideoRandomCrop, transforms_spatial.py:28-80):

    ToTensorVideo            uint8 (T,H,W,C) -> float32 (C,T,H,W) / 255            [torchvision, third party -- see below]
    Resize(size)             F.interpolate(bilinear, align_corners=False)         transforms_spatial.py:16-25
    RandomGrayScale(p=0.2)   0.2989 r + 0.5870 g + 0.1140 b on all 3 channels     transforms_tensor.py:12-34, functional_tensor.py:89-100
    ColorJitter(.4,.4,.4,.4) brightness / contrast / saturation / hue, shuffled   transforms_tensor.py:52-143, functional_tensor.py:103-162,254-417
    RandomHorizontalFlipVideo(p=0.5)  flip of W                                   [torchvision]
    NormalizeVideo(mean,std) (x - mean[c]) / std[c]                               [torchvision]

Third-party pieces absent from /root/reference: ``torchvision.transforms._transforms_video`` (requirements.txt pins
torchvision==0.7.0).  Their published definitions are restated here: ToTensorVideo = ``clip.float().permute(3,0,1,2)/255``,
RandomHorizontalFlipVideo = ``clip.flip(-1)`` when ``random.random() < p``, NormalizeVideo = ``(clip - mean[:,None,None,None])
/ std[:,None,None,None]``.

Pinned: oracle/gen_golden_augment.py runs the reference's own Resize / RandomGrayScale / ColorJitter classes (imported from
/root/reference with a torchvision stub carrying the three restated classes) under ``random.seed(s)`` and commits inputs'
seeds + outputs to tests/golden/augment.npz; tests/test_oracle_augment.py checks this file against them and checks that
``draw_params`` consumes Python's ``random`` exactly as the reference pipeline does; tests/test_oracle_vs_reference.py
re-runs the reference classes live on other seeds whenever /root/reference is present.
"""
from __future__ import annotations

import random
from dataclasses import dataclass, field
from typing import List, Sequence, Tuple

import torch

BRIGHTNESS, CONTRAST, SATURATION, HUE = 0, 1, 2, 3


@dataclass
class ClipParams:
    gray: bool = False
    flip: bool = False
    ops: List[Tuple[int, float]] = field(default_factory=list)     # (opcode, factor) in application order
    gray_after: bool = False      # aug_plus chain: RandomGrayScale sits behind the colour ops
    blur: bool = False            # aug_plus chain: GaussianBlur((3,3),(1.5,1.5)) hit


def draw_params(p_gray=0.2, brightness=0.4, contrast=0.4, saturation=0.4, hue=0.4, p_flip=0.5) -> ClipParams:
    """Consume ``random`` in the reference's order for ONE clip: RandomGrayScale.__call__ (transforms_tensor.py:29),
    ColorJitter.get_params (:107-127: uniform b, c, s, h then random.shuffle of the op list), RandomHorizontalFlipVideo."""
    gray = random.random() < p_gray
    ops = []
    if brightness:
        ops.append((BRIGHTNESS, random.uniform(max(0.0, 1 - brightness), 1 + brightness)))
    if contrast:
        ops.append((CONTRAST, random.uniform(max(0.0, 1 - contrast), 1 + contrast)))
    if saturation:
        ops.append((SATURATION, random.uniform(max(0.0, 1 - saturation), 1 + saturation)))
    if hue:
        ops.append((HUE, random.uniform(-hue, hue)))
    random.shuffle(ops)
    flip = random.random() < p_flip
    return ClipParams(gray, flip, ops)


def draw_params_plus(p_jitter=0.8, p_gray=0.2, p_blur=0.5, p_flip=0.5) -> ClipParams:
    """The `moco.aug_plus` chain (datasets/classification/__init__.py:203-218): RandomApply([ColorJitter(.4,.4,.4,.1)], 0.8)
    -> RandomGrayScale(0.2) -> RandomApply([GaussianBlur], 0.5) -> flip.  torchvision's RandomApply skips when
    ``p < random.random()``."""
    ops = []
    if not (p_jitter < random.random()):
        ops = [(BRIGHTNESS, random.uniform(0.6, 1.4)), (CONTRAST, random.uniform(0.6, 1.4)), (SATURATION, random.uniform(0.6, 1.4)),
               (HUE, random.uniform(-0.1, 0.1))]
        random.shuffle(ops)
    gray = random.random() < p_gray
    blur = not (p_blur < random.random())
    flip = random.random() < p_flip
    return ClipParams(gray, flip, ops, gray_after=True, blur=blur)


def gaussian_kernel2d(ksize=(3, 3), sigma=(1.5, 1.5)) -> torch.Tensor:
    """functional_tensor.py:420-500 (get_gaussian_kernel2d): normalised 1-D windows, outer product."""
    def win(n, s):
        g = torch.stack([torch.exp(torch.tensor(-(x - n // 2) ** 2 / float(2 * s ** 2))) for x in range(n)])
        return g / g.sum()
    kx, ky = win(ksize[0], sigma[0]), win(ksize[1], sigma[1])
    return torch.matmul(kx.unsqueeze(-1), ky.unsqueeze(-1).t())


def _gray(img: torch.Tensor) -> torch.Tensor:                       # functional_tensor.py:89-100
    g = 0.2989 * img[0] + 0.5870 * img[1] + 0.1140 * img[2]
    return g.expand_as(img).contiguous()


def _blend(a: torch.Tensor, b, ratio: float) -> torch.Tensor:       # functional_tensor.py:103-106
    return (ratio * a + (1 - ratio) * b).clamp(0, 1)


def _rgb_to_hsv(img: torch.Tensor) -> torch.Tensor:                 # functional_tensor.py:304-345
    flat = img.reshape(3, -1)
    r, g, b = flat[0], flat[1], flat[2]
    maxc, idx = flat.max(0)
    minc = flat.min(0).values
    delta = maxc - minc
    s = torch.where(maxc == 0, torch.zeros(1), delta / maxc)
    allh = torch.stack([(g - b) / delta, (b - r) / delta + 2.0, (r - g) / delta + 4.0])
    h = torch.gather(allh, 0, idx.unsqueeze(0)).squeeze(0)
    h = h.masked_fill(delta == 0, 0.0)
    h = (h / 6.0) % 1.0
    return torch.stack([h, s, maxc]).view_as(img)


def _hsv_to_rgb(img: torch.Tensor) -> torch.Tensor:                 # functional_tensor.py:254-300
    flat = img.reshape(3, -1)
    h, s, v = flat[0], flat[1], flat[2]
    hi = torch.floor(h * 6)
    f = h * 6 - hi
    vtpq = torch.stack([v, v * (1 - (1 - f) * s), v * (1 - s), v * (1 - f * s)])
    index = hi.long() % 6
    cmap = torch.tensor([[0, 3, 2, 2, 1, 0], [1, 0, 0, 3, 2, 2], [2, 2, 1, 0, 0, 3]])
    gi = torch.gather(cmap, 1, index.expand(3, -1))
    return torch.gather(vtpq, 0, gi).view_as(img)


def augment_clip(clip_u8: torch.Tensor, size: int, params: ClipParams, mean: Sequence[float], std: Sequence[float]) -> torch.Tensor:
    """clip_u8: (T,h,w,3) uint8, already cropped.  Returns (3,T,size,size) float32."""
    x = clip_u8.float().permute(3, 0, 1, 2) / 255.0                                   # ToTensorVideo
    x = torch.nn.functional.interpolate(x, size=size, mode="bilinear", align_corners=False)   # transforms_spatial.py:21-25
    if params.gray and not params.gray_after:
        x = _gray(x)
    for op, f in params.ops:
        if op == BRIGHTNESS:
            x = _blend(x, torch.zeros_like(x), f)                                      # functional_tensor.py:110-125
        elif op == CONTRAST:
            x = _blend(x, torch.mean(_gray(x)), f)                                     # :128-145
        elif op == SATURATION:
            x = _blend(x, _gray(x), f)                                                 # :148-162
        elif op == HUE:
            hsv = _rgb_to_hsv(x)                                                       # :376-417
            hsv = torch.cat([((hsv[0] + f) % 1.0).unsqueeze(0), hsv[1:]])
            x = _hsv_to_rgb(hsv)
    if params.gray and params.gray_after:
        x = _gray(x)
    if params.blur:                                                                   # transforms_tensor.py:146-204
        k = gaussian_kernel2d().repeat(3, 1, 1, 1)
        x = torch.nn.functional.conv2d(x.transpose(0, 1), k, padding=(1, 1), stride=1, groups=3).transpose(0, 1).contiguous()
    if params.flip:
        x = x.flip(-1)
    m = torch.tensor(mean, dtype=torch.float32)[:, None, None, None]
    s = torch.tensor(std, dtype=torch.float32)[:, None, None, None]
    return (x - m) / s


def synthetic_clip(seed: int, T: int, h: int, w: int) -> torch.Tensor:
    """Portable uint8 test clip: integer colour ramps + hash noise (no transcendental functions, so it is re-derived
    bit-identically on any host); exercises bilinear taps, hue sectors and saturation."""
    from . import portable as P
    import numpy as np
    noise = (P.uniform01("augclip", seed, T * h * w * 3).astype(np.float64) * 52).astype(np.int64).reshape(T, h, w, 3)
    t, y, x, c = np.meshgrid(np.arange(T), np.arange(h), np.arange(w), np.arange(3), indexing="ij")
    ramp = (x * 5 + y * 3 * (c + 1) + t * 7 + c * 40 + seed * 11) % 128
    ramp = np.where(ramp < 64, ramp, 127 - ramp)                     # triangle wave 0..63
    v = ramp * 204 // 63 + noise                                     # 0..204 + 0..51
    return torch.from_numpy(v.astype(np.uint8))
