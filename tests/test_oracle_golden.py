"""CPU: the oracle restatement reproduces every golden fixture generated from the reference."""
import numpy as np
import pytest

from golden_util import ALL_CASES, build_inputs, compare_to_golden, load_case, run_restatement, worst_grad_err
from oracle import portable as P


@pytest.mark.parametrize("arch,ws,seed", ALL_CASES)
def test_restatement_matches_golden(arch, ws, seed):
    z, meta = load_case(arch, ws, seed)
    spec, inputs = build_inputs(arch, meta)
    outs, states, moms = run_restatement(arch, meta, inputs)
    for r in range(ws):
        post = {k: v.detach().numpy() for k, v in states[r].items()}
        mom_post = {k: v.numpy() for k, v in outs[r]["momentum_post"].items()}
        out = {k: (v.numpy() if hasattr(v, "numpy") else v) for k, v in outs[r].items()
               if k not in ("grads", "momentum_post")}
        errs = compare_to_golden(z, r, out, post, mom_post, tol=2e-5, tol_grad=2e-4)
        # gradient summaries
        grads = {k: (None if g is None else g.numpy()) for k, g in outs[r]["grads"].items()}
        wkey, worst = worst_grad_err(z, r, grads)
        assert worst <= 2e-4, (wkey, worst)


def test_portable_generator_is_stable():
    # pins the hash so fixtures stay reproducible on any numpy
    u = P.uniform01("x", 3, 4)
    assert u.dtype == np.float32
    assert np.allclose(P.uniform01("x", 3, 4), u)
    p = P.permutation("perm:0", 1, 8)
    assert sorted(p.tolist()) == list(range(8))


@pytest.mark.parametrize("arch,ws,seed", ALL_CASES)
def test_fixture_carries_its_guard_band(arch, ws, seed):
    """Every pretext fixture's pre-step state holds the guard band of oracle/guard.py: rebuilt from the portable seeds + the bias
    values stored in the file, no ReLU input of the query pass (any rank) lies closer to zero than 0.6 of the band's floor (2e-5 ...
    2e-4 channel-sigmas by channel size) in the restatement's fp32 forward — the decisions a whole-step gradient comparison is
    discontinuous in are not left to anyone's rounding."""
    import torch
    from oracle import guard
    from oracle import restatement as S
    z, meta = load_case(arch, ws, seed)
    assert meta["nudges"], "no guard band in the fixture"
    spec, (state, mom, clips, perms_B, sh) = build_inputs(arch, meta)
    q = [S.diff_speed(torch.from_numpy(clips[r][0]), torch.from_numpy(clips[r][1]), torch.from_numpy(perms_B[r]), meta["speed"])[0]
         for r in range(ws)]
    st = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    evs = guard._trace_query(meta["arch"], meta.get("fc_type", "linear"), [st] * ws, q)
    margins = guard.relu_margins(evs)
    worst = min(margins, key=lambda t: t[2])
    assert len(margins) >= 8 and worst[2] >= 0.6, worst


def test_guard_shift_is_the_smallest_admissible_one():
    """oracle/guard.py:_shift_for — the bias movement of one channel: after it no value lies inside the band, and no smaller movement
    (either sign) would do (checked against a dense scan)."""
    from oracle.guard import _shift_for, band_eps
    rng = np.random.default_rng(5)
    assert band_eps(32) == band_eps(4096) > band_eps(4097) == band_eps(1 << 16) > band_eps((1 << 16) + 1)
    for n, band in ((32, 2e-4), (256, 2e-3), (2048, 3e-4), (20000, 5e-5)):
        vals = rng.standard_normal(n)
        vals[rng.integers(0, n, 3)] = rng.uniform(-band, band, 3)            # a few values inside the band
        d = _shift_for(vals, band)
        assert np.abs(vals + d).min() >= band * (1 - 1e-12)
        scan = np.linspace(-abs(d), abs(d), 4001)[1:-1]                        # every strictly smaller |movement|
        ok = np.abs(vals[None, :] + scan[:, None]).min(axis=1) >= band
        assert not ok[np.abs(scan) < abs(d) * (1 - 1e-3)].any(), (n, band, d)
    assert _shift_for(np.array([0.5, -0.7, 1.2]), 1e-3) == 0.0                 # nothing inside the band: nothing moves
