"""Portable deterministic generators for oracle / parity tests.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
anything under ``oracle/``.  The product (``rspnet_amd``) never does.

The reference initialises weights with torch's RNG (``nn.Conv3d`` default init, ``torch.randn`` for
the queue: /root/reference/moco/builder_diffspeed_diffloss.py:329-330), which is neither portable
across torch versions/devices nor small enough to commit (SURVEY.md §8c "weights / inputs").  The
oracle therefore fills every state-dict entry and every input clip from a counter-based integer
hash (splitmix64 → 24-bit mantissa float), which is bit-identical on any numpy.
"""
from __future__ import annotations

import zlib
from typing import Dict, Iterable, Tuple

import numpy as np

_GOLD = np.uint64(0x9E3779B97F4A7C15)
_M1 = np.uint64(0xBF58476D1CE4E5B9)
_M2 = np.uint64(0x94D049BB133111EB)


def _splitmix(z: np.ndarray) -> np.ndarray:
    with np.errstate(over="ignore"):
        z = (z ^ (z >> np.uint64(30))) * _M1
        z = (z ^ (z >> np.uint64(27))) * _M2
        z = z ^ (z >> np.uint64(31))
    return z


def key_id(name: str) -> int:
    return zlib.crc32(name.encode()) & 0xFFFFFFFF


def _uniform01_range(base: np.uint64, lo: int, hi: int) -> np.ndarray:
    with np.errstate(over="ignore"):
        idx = np.arange(lo, hi, dtype=np.uint64)
        z = _splitmix(_splitmix(idx + base) + _GOLD)
    return ((z >> np.uint64(40)).astype(np.float64) / float(1 << 24)).astype(np.float32)


def uniform01(name: str, seed: int, n: int) -> np.ndarray:
    """n floats in [0,1), exactly representable in fp32 (24-bit), from hash(seed, name, index).  Pure function of the index, so
    large requests are hashed in 1M-element slices on a thread pool (numpy releases the GIL): bit-identical, several times
    faster — the 28-46M-parameter states of the fixtures are rebuilt from it in every parity test."""
    with np.errstate(over="ignore"):
        base = (np.uint64(seed) * _GOLD) ^ (np.uint64(key_id(name)) << np.uint64(32))
    step = 1 << 20
    if n <= 2 * step:
        return _uniform01_range(base, 0, n)
    import os
    from concurrent.futures import ThreadPoolExecutor
    out = np.empty(n, dtype=np.float32)

    def work(lo):
        out[lo:min(n, lo + step)] = _uniform01_range(base, lo, min(n, lo + step))

    with ThreadPoolExecutor(max_workers=min(8, os.cpu_count() or 1)) as ex:
        list(ex.map(work, range(0, n, step)))
    return out


def uniform(name: str, seed: int, shape: Tuple[int, ...], lo: float, hi: float) -> np.ndarray:
    n = int(np.prod(shape)) if len(shape) else 1
    u = uniform01(name, seed, n).astype(np.float64)
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def permutation(name: str, seed: int, n: int) -> np.ndarray:
    """Deterministic permutation of range(n) (argsort of hashed keys, ties impossible at 24 bits + index)."""
    u = uniform01(name, seed, n).astype(np.float64) + np.arange(n) * 1e-12
    return np.argsort(u, kind="stable").astype(np.int64)


def sample_index(name: str, numel: int, count: int = 16) -> np.ndarray:
    """Fixed pseudo-random flat indices used to summarise big tensors in golden files."""
    u = uniform01("idx:" + name, 7, count).astype(np.float64)
    return np.minimum((u * numel).astype(np.int64), numel - 1)


def summarise(name: str, arr: np.ndarray, count: int = 16) -> np.ndarray:
    """[l2-norm, sum, count samples] as float64 — what golden files keep for large tensors."""
    flat = np.asarray(arr, dtype=np.float64).reshape(-1)
    idx = sample_index(name, flat.size, count)
    return np.concatenate([[np.sqrt((flat * flat).sum()), flat.sum()], flat[idx]])


def projections(name: str, arr: np.ndarray, count: int = 64) -> np.ndarray:
    """Count-sketch of a tensor: element i is added with a pseudo-random sign into one of `count` (<= 64) buckets, scaled by
    1/sqrt(numel); float64.  E|S x|^2 = |x|^2, so with the golden sketch g and a measured one m,  |m - g| / |g|  estimates the
    relative L2 distance of the full tensors to about 1/sqrt(count) of itself — unlike the plain element sum, whose error can
    exceed the L2 error by sqrt(numel) when the difference has a coherent component.  Bucket and sign come from one byte per
    element of numpy's PCG64 raw stream (stable by numpy's bit-generator policy), seeded from the tensor's name: one cheap pass."""
    flat = np.asarray(arr, dtype=np.float64).reshape(-1)
    n = flat.size
    raw = np.random.PCG64(0x5EED0000 + key_id("sketch:" + name)).random_raw((n + 7) // 8).view(np.uint8)[:n]
    sign = ((raw >> 6) & 1).astype(np.float64) * 2.0 - 1.0
    return np.bincount(raw & (count - 1), weights=flat * sign, minlength=count) / np.sqrt(n)


def proj_rel_err(name: str, mine: np.ndarray, golden_proj: np.ndarray) -> float:
    m = projections(name, mine, len(golden_proj))
    den = float(np.sqrt((golden_proj ** 2).sum()))
    return float(np.sqrt(((m - golden_proj) ** 2).sum()) / max(den, 1e-12))


def fill_state(spec: Dict[str, Tuple[Tuple[int, ...], str]], seed: int) -> Dict[str, np.ndarray]:
    """Fill a state dict described by {key: (shape, dtype)} with well-conditioned portable values.

    Rules (chosen so BN'd activations and the projected embeddings are well spread, SURVEY.md §8c
    "conditioning"):
      * BN running_var U(0.5,1.5), running_mean U(-0.1,0.1), num_batches_tracked = seed % 5,
        BN weight U(0.5,1.5), BN bias U(-0.2,0.2);
      * conv weight (5-D) U(±sqrt(6/fan_in)); linear weight (2-D) U(±sqrt(3/fan_in)); other 1-D
        (conv / linear bias) U(-0.1,0.1);
      * queue U(-1,1) L2-normalised over dim 0; queue_ptr = 0 (callers override).
    """
    keys = set(spec)
    out: Dict[str, np.ndarray] = {}
    for key, (shape, dtype) in spec.items():
        shape = tuple(shape)
        if key == "queue":
            q = uniform(key, seed, shape, -1.0, 1.0).astype(np.float64)
            q /= np.maximum(np.sqrt((q * q).sum(axis=0, keepdims=True)), 1e-12)
            out[key] = q.astype(np.float32)
        elif key == "queue_ptr":
            out[key] = np.zeros(shape, dtype=np.int64)
        elif key.endswith("num_batches_tracked"):
            out[key] = np.full(shape, seed % 5, dtype=np.int64)
        elif key.endswith("running_var"):
            out[key] = uniform(key, seed, shape, 0.5, 1.5)
        elif key.endswith("running_mean"):
            out[key] = uniform(key, seed, shape, -0.1, 0.1)
        elif len(shape) == 1 and key.rsplit(".", 1)[0] + ".running_mean" in keys:
            if key.endswith(".weight"):
                out[key] = uniform(key, seed, shape, 0.5, 1.5)
            else:
                out[key] = uniform(key, seed, shape, -0.2, 0.2)
        elif len(shape) == 5:
            fan_in = shape[1] * shape[2] * shape[3] * shape[4]
            a = float(np.sqrt(6.0 / fan_in))
            out[key] = uniform(key, seed, shape, -a, a)
        elif len(shape) == 2:
            a = float(np.sqrt(3.0 / shape[1]))
            out[key] = uniform(key, seed, shape, -a, a)
        else:
            out[key] = uniform(key, seed, shape, -0.1, 0.1)
        assert str(out[key].dtype) == dtype.replace("torch.", ""), (key, out[key].dtype, dtype)
    return out


def fill_momentum(spec_keys: Iterable[Tuple[str, Tuple[int, ...]]], seed: int, scale: float = 0.01):
    return {k: uniform("mom:" + k, seed, tuple(s), -scale, scale) for k, s in spec_keys}


def clips(seed: int, rank: int, shape: Tuple[int, ...]):
    """Synthetic (im_q, im_k) pair, NCDHW fp32 as the reference takes them (B,3,T,H,W).

    i.i.d. noise clips give sample-independent pooled features (near-degenerate
    logits), so each sample is a moving plane wave with per-sample frequency / per-channel phase
    plus noise; im_k is the "same video, other augmentation": same wave, shifted phase, new noise.
    """
    B, C, T, H, W = shape
    tag = f"{rank}"
    f = uniform("clip_f:" + tag, seed, (B, 3), 0.5, 3.0).astype(np.float64)      # fx, fy, ft
    amp = uniform("clip_a:" + tag, seed, (B, C), 0.5, 1.5).astype(np.float64)
    ph = uniform("clip_p:" + tag, seed, (B, C), 0.0, 6.2831853).astype(np.float64)
    dc = uniform("clip_d:" + tag, seed, (B, C), -0.5, 0.5).astype(np.float64)
    t = (np.arange(T) / T)[None, None, :, None, None]
    y = (np.arange(H) / H)[None, None, None, :, None]
    x = (np.arange(W) / W)[None, None, None, None, :]
    arg = 6.283185307179586 * (f[:, 0, None, None, None, None] * x + f[:, 1, None, None, None, None] * y
                               + f[:, 2, None, None, None, None] * t)
    out = []
    for which, shift in (("q", 0.0), ("k", 0.7)):
        wave = amp[:, :, None, None, None] * np.sin(arg + ph[:, :, None, None, None] + shift) \
            + dc[:, :, None, None, None]
        noise = uniform(f"im_{which}:{tag}", seed, shape, -0.5, 0.5).astype(np.float64)
        out.append(np.round((wave + noise) * 4096.0) / 4096.0)   # quantise: kills libm ulp differences
    return out[0].astype(np.float32), out[1].astype(np.float32)
