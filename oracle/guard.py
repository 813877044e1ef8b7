"""Guard band around the ReLU decisions of a fixture's query pass.  TEST INFRASTRUCTURE ONLY (build container, fixture generation).

Why.  The backward pass of a ReLU network is discontinuous in the forward values: an element of a ReLU's input within rounding
distance of zero gets its mask — and with it an O(1) share of that layer's gradient — from the summation order of whoever
evaluates it.  On a fixture whose late layers hold a few dozen positions per channel ONE such element moves a weight gradient by
percents between two correct fp32 implementations (rounds 3-5: whole-step gates of 5e-2 ... 1e-1 on S3D-G / ResNet-34 / -50, and
a second tile plan as a witness).  Rejecting seeds cannot fix that: a fixture has 10^6 ... 10^7 ReLU inputs, a forward error of
1e-6 relative puts a handful of them inside the band on EVERY seed.

What.  The fixture's pre-step state stays what oracle/portable.py:fill_state draws — except that the additive per-channel term in
front of each ReLU of encoder_q (a BatchNorm bias; the bias of the 'mlp' / 'conv' heads' first layer) is moved, channel by channel
and only where needed, by the smallest amount that leaves NO element of that channel's ReLU input closer to zero than
`band_eps` x (the channel's standard deviation) — for every rank's query clips at once.  A bias shifts its channel's ReLU input
exactly and leaves the layer's normalised values alone, so the layers are settled one after the other in evaluation order, each
with one fp64 forward of the restatement (pinned to the reference at 2e-5; fp64 makes the band independent of anyone's rounding).
Typically 1-5 % of the channels of a layer move, by ~1e-4 of their scale; the moved values are kept in the fixture
(`nudge.idx.<key>` / `nudge.val.<key>`: the NEW fp32 values) and applied by oracle/gen_golden.py:case_inputs.

What it does not cover: max-pool arg-max decisions (a per-channel shift moves both candidates) — those stay guarded by seed
selection on the small layers (oracle/ref_harness.py) and are light on the large ones; the key passes need no guard (nothing is
differentiated through them)."""
from __future__ import annotations

from typing import Dict, List, Tuple

import numpy as np
import torch

from oracle import restatement as S

# FLOOR of the band half-width in units of the channel's standard deviation, by the number of values a channel holds (all ranks
# together); the band of a ReLU is this or DRIFT_FACTOR x the measured fp32 drift at its depth, whichever is larger (guard_band).
# The forward values of two correct fp32 evaluations differ by 1e-6 ... 2e-6 channel-sigmas per layer (1.4e-6 for
# one fp32 chain at K = 13 824) and the differences accumulate with depth: measured on S3D-G's 77 units, the reference's own fp32
# forward sits up to 3.6e-5 sigmas from the restatement's fp64 values at a late layer.  The band therefore widens where it can — the
# small late layers, which are also the ones where a single flipped element weighs most: 2e-4 up to 4k values per channel, 5e-5 up to
# 64k, 2e-5 beyond (the first layers, in front of which nothing has accumulated yet and where a flipped element weighs
# 1 / sqrt(elements) of its layer's gradient).
def band_eps(values_per_channel: int) -> float:
    return 2e-4 if values_per_channel <= 4096 else (5e-5 if values_per_channel <= (1 << 16) else 2e-5)


def _trace_query(arch: str, fc_type: str, states64: List[dict], q_clips: List[torch.Tensor], forward=None):
    """One traced forward of encoder_q (or of `forward(state, clips)`: the fine-tune model) per rank -> per-rank event lists."""
    out = []
    for sd, x in zip(states64, q_clips):
        scratch = {k: (v.clone() if not v.is_floating_point() or k.endswith(("running_mean", "running_var")) else v) for k, v in sd.items()}
        S._TRACE[0] = ev = []
        try:
            with torch.no_grad():
                if forward is not None:
                    forward(scratch, x)
                else:
                    S.encoder_forward(arch, scratch, "encoder_q", x, fc_type)
        finally:
            S._TRACE[0] = None
        out.append(ev)
    return out


def _channel_rows(z: torch.Tensor) -> np.ndarray:
    """(C, elements per channel) view of a ReLU input: (B,C,T,H,W) or (B,C)."""
    if z.dim() == 2:
        return z.t().contiguous().numpy()
    return z.transpose(0, 1).reshape(z.shape[1], -1).numpy()


def _shift_for(vals: np.ndarray, band: float) -> float:
    """Smallest |d| such that no element of vals + d lies inside (-band, band)."""
    u = np.sort(-vals)                                   # forbidden centres: d must keep `band` away from each of them
    lo = np.concatenate([[-np.inf], u + band])           # admissible intervals [u_i + band, u_{i+1} - band]
    hi = np.concatenate([u - band, [np.inf]])
    ok = hi >= lo
    cand = np.clip(0.0, lo[ok], hi[ok])                  # the point of each admissible interval closest to 0
    return float(cand[np.argmin(np.abs(cand))])


def relu_margins(events_per_rank) -> List[Tuple[int, str, float]]:
    """[(event index, bias key, smallest |z| / std over the channels and ranks, in units of the layer's band)] for every ReLU."""
    out = []
    ev0 = events_per_rank[0]
    last_shift = None
    for i, e in enumerate(ev0):
        if e[0] == "shift":
            last_shift = e[1]
        elif e[0] == "relu":
            rows = np.concatenate([_channel_rows(ev[i][1]) for ev in events_per_rank], axis=1)
            sd = np.maximum(rows.std(axis=1), 1e-30)
            out.append((i, last_shift, float((np.abs(rows).min(axis=1) / sd).min() / band_eps(rows.shape[1]))))
    return out


DRIFT_FACTOR = 6.0


def guard_band(arch: str, fc_type: str, state: Dict[str, np.ndarray], q_clips: List[np.ndarray], verbose: bool = False, forward=None):
    """Returns ({bias key: (channel indices int32, NEW values float32)}, report).  `state`: the fixture's pre-step state (numpy,
    not modified); q_clips: the query clips of every rank as encoder_q sees them (after _diff_speed).  forward(state, clips):
    another traced forward than encoder_q's — the fine-tune model of oracle/gen_golden_finetune.py.

    Band of a ReLU = max(band_eps(values per channel), DRIFT_FACTOR x drift), drift = the largest distance, in channel sigmas,
    between this ReLU's input in the restatement's fp32 forward and in its fp64 forward — the MEASURED size of what separates two
    correct evaluations at this depth (1e-5 at the first layers, 1e-4 ... 5e-4 at the end of ResNet-50's 49 units: the differences
    of every layer pass through all the BatchNorms behind it)."""
    st = {k: torch.from_numpy(np.array(v)) for k, v in state.items()}
    st64 = {k: (v.double() if v.dtype == torch.float32 else v) for k, v in st.items()}
    clips64 = [torch.from_numpy(np.asarray(c)).double() for c in q_clips]
    clips32 = [c.float() for c in clips64]
    ws = len(clips64)
    prev = torch.get_default_dtype()
    moved: Dict[str, Dict[int, float]] = {}
    settled = 0          # ReLU events before this index are settled
    passes = 0
    bands: Dict[int, float] = {}

    def traces():
        torch.set_default_dtype(torch.float64)
        e64 = _trace_query(arch, fc_type, [st64] * ws, clips64, forward)
        torch.set_default_dtype(torch.float32)
        st32 = {k: (v.float() if v.dtype == torch.float64 else v) for k, v in st64.items()}
        e32 = _trace_query(arch, fc_type, [st32] * ws, clips32, forward)
        return e64, e32

    try:
        while True:
            evs, evs32 = traces()
            passes += 1
            ev0 = evs[0]
            last_shift, first_fixed = None, None
            for i, e in enumerate(ev0):
                if e[0] == "shift":
                    last_shift = e[1]
                    continue
                if e[0] != "relu" or i < settled:
                    continue
                rows = np.concatenate([_channel_rows(ev[i][1]) for ev in evs], axis=1)
                rows32 = np.concatenate([_channel_rows(ev[i][1]) for ev in evs32], axis=1).astype(np.float64)
                sd = np.maximum(rows.std(axis=1), 1e-30)
                drift = float((np.abs(rows32 - rows).max(axis=1) / sd).max())
                eps = max(band_eps(rows.shape[1]), DRIFT_FACTOR * drift)
                bands[i] = eps
                bad = np.nonzero(np.abs(rows).min(axis=1) < eps * sd)[0]
                if bad.size == 0:
                    if first_fixed is None:
                        settled = i + 1
                    continue
                assert last_shift is not None, "a ReLU without an additive per-channel term in front of it"
                key = last_shift
                assert st64[key].shape[0] == rows.shape[0], (key, st64[key].shape, rows.shape)
                for c in bad:
                    d = _shift_for(rows[c], 1.25 * eps * sd[c])          # (a quarter more: the new value is rounded to fp32)
                    new = np.float32(float(st64[key][c]) + d)
                    st64[key][c] = float(new)
                    moved.setdefault(key, {})[int(c)] = float(new)
                if verbose:
                    print(f"  guard: {key}: {bad.size} of {rows.shape[0]} channels moved ({rows.shape[1]} values each, band {eps:.1e} sigma, "
                          f"fp32 drift {drift:.1e})", flush=True)
                # Every violating ReLU of this pass is moved with the values of this pass: units that do not read one another (the
                # branches of an inception block) settle together; a unit downstream of a moved one is simply checked — and, where its
                # inputs have shifted onto a new near-zero element, moved — again in the next pass, which starts at the first moved one.
                if first_fixed is None:
                    first_fixed = i
            fixed = first_fixed is not None
            if fixed:
                settled = first_fixed
            if not fixed:
                break
            assert passes < 400, "guard band does not converge"
        # smallest |z| over the fp32 forward, in units of each ReLU's band: what another fp32 evaluation has left of the band
        worst32 = 1e9
        for i, e in enumerate(ev0):
            if e[0] == "relu":
                rows = np.concatenate([_channel_rows(ev[i][1]) for ev in evs], axis=1)
                rows32 = np.concatenate([_channel_rows(ev[i][1]) for ev in evs32], axis=1).astype(np.float64)
                sd = np.maximum(rows.std(axis=1), 1e-30)
                worst32 = min(worst32, float((np.abs(rows32).min(axis=1) / sd).min() / bands[i]))
    finally:
        torch.set_default_dtype(prev)
    nudges = {k: (np.array(sorted(v), dtype=np.int32), np.array([v[c] for c in sorted(v)], dtype=np.float32)) for k, v in moved.items()}
    report = {"passes": passes, "relus": len(bands), "band_min": min(bands.values()), "band_max": max(bands.values()),
              "fp32_margin_in_bands": worst32, "channels_moved": int(sum(len(v) for v in moved.values())), "biases_touched": len(moved)}
    return nudges, report


def apply_nudges(state: Dict[str, np.ndarray], nudges) -> None:
    """In place: the guard band's bias values over fill_state's (nudges: {key: (indices, values)} or None)."""
    if not nudges:
        return
    for key, (idx, val) in nudges.items():
        state[key][np.asarray(idx, dtype=np.int64)] = np.asarray(val, dtype=np.float32)
