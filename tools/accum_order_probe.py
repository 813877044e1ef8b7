#!/usr/bin/env python
"""Why does the HIP path decide ~2x as many ReLU / max-pool knife edges differently from fp64 as the CPU reference does (tools/grad_census.py)?
CPU-only experiment: the rounding error of ONE fp32 accumulation chain over all K products of an output — what a
v_mfma_f32_32x32x2_f32 tile does: acc += a0*b0 + a1*b1, K/2 times — against torch's CPU conv3d (oneDNN: K summed in panels, partial
sums combined: a two-level summation), both against fp64, for the K of C3D's layers (conv2 1 728, conv3b / conv4a 6 912, conv4b /
conv5 13 824).  The chain's error grows with sqrt(K); the panel sum's does not."""
import numpy as np
import torch

torch.manual_seed(0)
torch.set_num_threads(8)
print("K      torch CPU conv3d fp32 vs fp64 (rel rms)   one fp32 chain, MFMA-style (rel rms)   ratio")
for Cin, Cout in ((3, 64), (64, 128), (256, 256), (512, 512)):
    x = torch.randn(2, Cin, 4, 14, 14)
    w = torch.randn(Cout, Cin, 3, 3, 3) * (Cin * 27) ** -0.5
    y64 = torch.nn.functional.conv3d(x.double(), w.double(), padding=1)
    y32 = torch.nn.functional.conv3d(x, w, padding=1)
    e = float(((y32.double() - y64) ** 2).mean().sqrt() / (y64 ** 2).mean().sqrt())
    K = Cin * 27 + (Cin * 27) % 2
    n = 3000
    a = np.random.default_rng(1).standard_normal((n, K)).astype(np.float32)
    b = (np.random.default_rng(2).standard_normal((n, K)) / np.sqrt(K)).astype(np.float32)
    ref = (a.astype(np.float64) * b).sum(1)
    acc = np.zeros(n, np.float32)
    for k in range(0, K, 2):      # (the pair sum taken exactly: the most favourable reading of the instruction)
        acc = (acc.astype(np.float64) + (a[:, k].astype(np.float64) * b[:, k] + a[:, k + 1].astype(np.float64) * b[:, k + 1])).astype(np.float32)
    es = np.sqrt(((acc - ref) ** 2).mean()) / np.sqrt((ref ** 2).mean())
    print(f"{Cin * 27:6d} {e:20.2e} {es:40.2e} {es / e:25.2f}")
