#!/usr/bin/env python
"""How accurate would fp32 products emulated on the bf16 matrix pipe be?  (a decision input, not a product path.)

a = a_hi + a_mid + a_lo with three bf16 terms (round-to-nearest-even splits), a.b ~ sum of the 6 cross terms of order <= 2
(hi.hi, hi.mid, mid.hi, hi.lo, lo.hi, mid.mid), each term an exact bf16 x bf16 product accumulated in fp32 -- what a
v_mfma_f32_32x32x16_bf16 chain computes.  Compared on conv-like dot products (K = 3456 = C3D conv2, and 13824 = conv4b) against
float64, next to the plain fp32 fma chain the current kernels run and a 3-term (hi.hi, hi.lo', lo'.hi with a 2-way split) variant."""
import numpy as np
import torch


def bf16(x):
    return x.to(torch.bfloat16).to(torch.float32)


def split3(x):
    hi = bf16(x)
    mid = bf16(x - hi)
    lo = bf16(x - hi - mid)
    return hi, mid, lo


def run(K, rows=512, seed=0):
    g = torch.Generator().manual_seed(seed)
    a = torch.relu(torch.randn(rows, K, generator=g)) * 1.3          # post-ReLU activations
    b = torch.randn(rows, K, generator=g) * (2.0 / K) ** 0.5         # He-scaled weights
    exact = (a.double() * b.double()).sum(1)
    scale = (a.double().abs() * b.double().abs()).sum(1)             # condition-independent error scale sum |a||b|
    f32 = torch.zeros(rows)
    for k in range(K):                                               # sequential fp32 fma chain, as the MFMA accumulates
        f32 = torch.addcmul(f32, a[:, k], b[:, k])
    ah, am, al = split3(a)
    bh, bm, bl = split3(b)
    six = torch.zeros(rows)
    for ta, tb in ((al, bh), (ah, bl), (am, bm), (am, bh), (ah, bm), (ah, bh)):      # small terms first
        six = six + (ta * tb).sum(1, dtype=torch.float32)
    a2h, a2l = bf16(a), bf16(a - bf16(a))
    b2h, b2l = bf16(b), bf16(b - bf16(b))
    three = (a2l * b2h).sum(1, dtype=torch.float32) + (a2h * b2l).sum(1, dtype=torch.float32) + (a2h * b2h).sum(1, dtype=torch.float32)
    one = (a2h * b2h).sum(1, dtype=torch.float32)
    for name, v in (("fp32 fma chain (current kernels)", f32), ("bf16 x6 (3-way split)", six), ("bf16 x3 (2-way split)", three),
                    ("plain bf16", one)):
        err = ((v.double() - exact).abs() / scale).max().item()
        print(f"  K={K:6d}  {name:34s} max |err| / sum|a||b| = {err:.2e}")


if __name__ == "__main__":
    for K in (3456, 13824):
        run(K)
