"""Teacher-forced, per-op replay of a whole pretext step (test infrastructure).

`Recorder` wraps the CPU checker backend (tests/cpu_ops.py — itself pinned to the reference goldens by the CPU suite) and
records every op call of one full step — arguments before the call, every tensor the call changed, and its results — while
the step runs through rspnet_amd's real host logic (plan executor, concat slices, residual fan-out, gate, heads, losses).
`replay` then feeds each recorded call's inputs to another backend (the HIP kernels on the GPU) and compares what it
produces.  Every kernel is thereby checked inside the real composition of the backbone's backward (gate bwd -> concat slice
-> fan-out add -> overlapping max-pool bwd -> BN bwd -> dgrad/wgrad ...) on exactly the tensors the reference path would
hand it, so rounding differences cannot accumulate from unit to unit and the comparison can be tight (2e-5).

Knife edges.  A ReLU mask / pool arg-max decided on a value within fp32 rounding of zero / of its neighbour is made
differently by any two correct implementations, and one re-routed element moves a small layer's gradient by percents.  The
recorder therefore zeroes the incoming gradient (`dout`) of exactly those elements (|z| or top-2 gap below 1e-5 of the
layer's range) BEFORE the checker op runs: the chain stays self-consistent, nothing depends on an undecidable mask, and the
remaining elements must agree to the last digits."""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Any, Dict, List, Tuple

import numpy as np
import torch
import torch.nn.functional as F

KNIFE = 1e-5


def _is_t(x):
    return isinstance(x, torch.Tensor)


@dataclass
class Snap:
    data: torch.Tensor          # contiguous clone of the values
    stride: Tuple[int, ...]     # original strides (channel-slice views keep their pitch)
    contiguous: bool


def _snap(t: torch.Tensor) -> Snap:
    return Snap(t.detach().clone(memory_format=torch.contiguous_format).cpu(), tuple(t.stride()), t.is_contiguous())


def _materialise(s: Snap, dev) -> torch.Tensor:
    if s.contiguous:
        return s.data.to(dev)
    out = torch.empty_strided(tuple(s.data.shape), s.stride, dtype=s.data.dtype, device=dev)
    out.copy_(s.data)
    return out


@dataclass
class Call:
    name: str
    args: List[Any]                       # Snap for tensors, plain values otherwise
    kwargs: Dict[str, Any]
    changed: Dict[Any, Snap]              # arg position / kw name -> value after the call (in-place outputs)
    results: Any                          # Snap / tuple of (Snap | None | value)
    aliases: Dict[str, int] = field(default_factory=dict)   # kw name -> position of the positional arg it is the same tensor as


def _pool_unsafe(z_ncdhw, pg, relu):
    """Pooled outputs whose arg-max (or the ReLU mask of the max) is not robustly decided.  z = pre-activation values."""
    m1, idx = F.max_pool3d(z_ncdhw, pg.k, pg.s, pg.p, return_indices=True)
    z2 = z_ncdhw.flatten(2).scatter(2, idx.flatten(2), float("-inf")).view_as(z_ncdhw)
    m2 = F.max_pool3d(z2, pg.k, pg.s, pg.p)
    d = KNIFE * (float(z_ncdhw.abs().max()) or 1.0)
    if relu:      # all-negative windows pass no gradient under either routing; only a positive max needs a clear runner-up
        unsafe = (m1.abs() < d) | ((m1 >= d) & ((m1 - m2) < d))
    else:
        unsafe = (m1 - m2) < d
    return unsafe.permute(0, 2, 3, 4, 1)


class Recorder:
    """Op backend that forwards to `inner` and records each call."""
    name = "recorder"
    gate_pool_keep = False         # (the kept forward of a gated front-end unit is recorded as its primitives: pool apart)

    def __init__(self, inner):
        self.inner = inner
        self.calls: List[Call] = []
        self.event_log = None
        self.neutralised = 0

    def _neutralise(self, name, args, kwargs):
        if name != "bn_act_pool_bwd":
            return args
        pg, y, residual, dout, gamma, mi, ss, relu = args[:8]
        z = y * ss[0] + ss[1]
        if residual is not None:
            z = z + residual
        pooled = pg.k != (1, 1, 1) or pg.s != (1, 1, 1)
        if not relu and not pooled:
            return args
        scale = float(z.abs().max()) or 1.0
        if pooled:
            unsafe = _pool_unsafe(z.permute(0, 4, 1, 2, 3), pg, relu)
        else:
            unsafe = z.abs() < KNIFE * scale
        n = int(unsafe.sum())
        if n:
            self.neutralised += n
            dout = dout.clone()
            dout[unsafe] = 0
            args = tuple(args[:3]) + (dout,) + tuple(args[4:])
        return args

    def bn_act_gate_bwd(self, pg, y, dout, gamma, mean_invstd, scale_shift, relu, w, mean, gate, dgamma_out, dbeta_out, dw_out,
                        db_out):
        """The fused backward is recorded as the three primitives it stands for (each through this recorder, so the ReLU knife
        edges of the BatchNorm backward are neutralised as everywhere else); the fused device path itself is pinned to those
        primitives by tests/test_kernels_gpu.py::test_bn_act_gate_bwd_fused_matches_the_two_ops."""
        a = self.bn_act_pool_fwd(pg, y, scale_shift, None, relu)
        dx = self.gate_bwd(a, dout, w, mean, gate, dw_out, db_out)
        dy, _ = self.bn_act_pool_bwd(pg, y, None, dx, gamma, mean_invstd, scale_shift, relu, False, dgamma_out, dbeta_out)
        return dy

    def bn_act_maxpool_fwd(self, pg, y, scale_shift, relu, keep):
        """BatchNorm apply + overlapping max-pool in one pass (the ResNet stems) is recorded as the two primitives it stands for; the
        fused device path is pinned to them bit for bit by tests/test_kernels_gpu.py (overlapping-window test)."""
        from rspnet_amd.ops import PoolGeom
        a = self.bn_act_pool_fwd(PoolGeom(pg.N, pg.Di, pg.Hi, pg.Wi, pg.C), y, scale_shift, None, relu)
        return self.maxpool_fwd(pg, a, keep)

    def clip_gather_multi(self, jobs, T_out, c_out=None):
        """The step's three gathers in one launch are recorded as the single gathers they stand for; the fused device path is pinned
        to them bit for bit by tests/test_kernels_gpu.py::test_clip_gather."""
        return [self.clip_gather(im, src, step, T_out, c_out) for im, src, step in jobs]

    def __getattr__(self, name):
        fn = getattr(self.inner, name)
        if not callable(fn) or name in ("pack_signature", "bn_ema_set"):      # (host queries / set builders: nothing to replay)
            return fn

        def wrapped(*args, **kwargs):
            args = self._neutralise(name, args, kwargs)
            pre_a = [_snap(a) if _is_t(a) else a for a in args]
            pre_k = {k: (_snap(v) if _is_t(v) else v) for k, v in kwargs.items()}
            aliases = {k: i for k, v in kwargs.items() if _is_t(v) for i, a in enumerate(args)
                       if _is_t(a) and a.data_ptr() == v.data_ptr() and a.shape == v.shape}
            out = fn(*args, **kwargs)
            changed = {}
            for i, a in enumerate(args):
                if _is_t(a) and not torch.equal(a.detach().cpu(), pre_a[i].data.view_as(a)):
                    changed[i] = _snap(a)
            for k, v in kwargs.items():
                if _is_t(v) and not torch.equal(v.detach().cpu(), pre_k[k].data.view_as(v)):
                    changed[k] = _snap(v)
            if isinstance(out, tuple):
                res = tuple(_snap(o) if _is_t(o) else o for o in out)
            else:
                res = _snap(out) if _is_t(out) else out
            self.calls.append(Call(name, pre_a, pre_k, changed, res, aliases))
            return out
        return wrapped


def _err(got: torch.Tensor, exp: torch.Tensor) -> float:
    got = got.detach().double().cpu()
    exp = exp.detach().double().cpu()
    if exp.numel() == 0:
        return 0.0
    return float((got - exp).abs().max() / max(float(exp.abs().max()), 1e-6))


def replay(calls: List[Call], backend, dev, tol=2e-5, skip=("conv_pack_fwd", "pack_set")):
    """Run every recorded call on `backend` with the recorded inputs; returns {op name: worst rel err}; asserts <= tol."""
    worst: Dict[str, float] = {}
    where: Dict[str, int] = {}

    def note(name, e, ci, what):
        if e > worst.get(name, -1.0):
            worst[name] = e
            where[name] = ci
        assert e <= tol, f"call #{ci} {name}: {what} rel err {e:.3e} > {tol}"

    for ci, c in enumerate(calls):
        if c.name in skip:
            continue
        args = [_materialise(a, dev) if isinstance(a, Snap) else a for a in c.args]
        kwargs = {k: (_materialise(v, dev) if isinstance(v, Snap) else v) for k, v in c.kwargs.items()}
        for k, i in c.aliases.items():
            kwargs[k] = args[i]
        if c.name == "conv_fwd":
            # the checker's "packed" weight is the reference layout; the device backend packs it its own way
            args[2] = backend.conv_pack_fwd(args[0], args[2].contiguous())
        if c.name == "conv_dgrad_packed":
            # checker "packed" = zero-padded reference layout: re-pack it through the backend's own batched path
            ps = backend.pack_set([(args[0], 1, args[2].contiguous())])
            ps.run()
            args[2] = ps.packed[0]
        if c.name == "bn_finalize":
            args[0] = args[0].float().contiguous()          # checker keeps its single stat tile in fp64
        out = getattr(backend, c.name)(*args, **kwargs)
        # in-place outputs
        for key, snap in c.changed.items():
            t = args[key] if isinstance(key, int) else kwargs[key]
            if t.dtype.is_floating_point:
                note(c.name, _err(t, snap.data), ci, f"in-place arg {key}")
            else:
                assert torch.equal(t.cpu(), snap.data), (ci, c.name, key)
        outs = out if isinstance(out, tuple) else (out,)
        exps = c.results if isinstance(c.results, tuple) else (c.results,)
        assert len(outs) == len(exps), (c.name, len(outs), len(exps))
        for oi, (o, e) in enumerate(zip(outs, exps)):
            if not isinstance(e, Snap):
                assert (o is None) == (e is None), (ci, c.name, oi)
                continue
            if c.name == "conv_fwd" and oi == 1:
                # stat partials: tile decomposition is the backend's own; the per-channel totals must agree.  Scale of a
                # channel's sum is sqrt(rows * sumsq) (the sum itself may cancel to ~0).
                got = o.double().sum(dim=0).cpu()
                exp = e.data.double().sum(dim=0)
                rows = args[0].rows
                scale = torch.sqrt(rows * exp[:, 1].abs().max()).clamp_min(1e-6)
                note(c.name, float((got[:, 0] - exp[:, 0]).abs().max() / scale), ci, "stat sums")
                note(c.name, _err(got[:, 1], exp[:, 1]), ci, "stat sums of squares")
                continue
            if c.name == "maxpool_fwd" and oi == 1:
                # arg-max: ties (post-ReLU zeros) may resolve differently; the selected element must hold the max
                x = args[1]
                N, D, H, W, Cc = x.shape
                flat = x.permute(0, 4, 1, 2, 3).reshape(N, Cc, -1)
                picked = flat.gather(2, o.permute(0, 4, 1, 2, 3).reshape(N, Cc, -1).long())
                assert torch.equal(picked.view(N, Cc, *o.shape[1:4]).permute(0, 2, 3, 4, 1).contiguous(), outs[0]), (ci, c.name)
                continue
            if o.dtype.is_floating_point:
                note(c.name, _err(o, e.data), ci, f"result {oi}")
            else:
                assert torch.equal(o.cpu(), e.data), (ci, c.name, oi)
    return worst, where
