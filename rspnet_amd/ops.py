"""Tensor-level gateway to the HIP kernels (one method per C-ABI entry point family).

Activations are torch tensors of shape (N, D, H, W, C) (NDHWC, contiguous, fp32); torch is used only as the
owner of device memory and of the current stream.  ``HipOps`` is the only product backend: it requires CUDA/HIP
tensors and the built ``librspnet_hip.so`` and raises otherwise.  ``set_backend`` exists so that the host logic
(graph executor, DDP exchange, optimizer) can be exercised by CPU tests with a checker backend living under
``tests/``; nothing in this package ever selects another backend by itself.
"""
from __future__ import annotations

import collections
import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib


@dataclass(frozen=True)
class ConvGeom:
    """Geometry of one nn.Conv3d application (groups=1, dilation=1)."""
    N: int
    Di: int
    Hi: int
    Wi: int
    Cin: int
    Cout: int
    k: Tuple[int, int, int]
    s: Tuple[int, int, int]
    p: Tuple[int, int, int]
    Cin_alg: float = 0  # algorithmic input channels when Cin includes zero padding (FLOP / byte accounting only; may be fractional)

    @property
    def out_dims(self):
        return tuple((i + 2 * p - k) // s + 1 for i, k, s, p in zip((self.Di, self.Hi, self.Wi), self.k, self.s, self.p))

    @property
    def rows(self):
        d, h, w = self.out_dims
        return self.N * d * h * w

    @property
    def flops(self):
        """Algorithmic FLOPs of one pass (2*MACs, padded taps included — SURVEY.md §8d)."""
        return int(2 * self.rows * self.Cout * (self.Cin_alg or self.Cin) * self.k[0] * self.k[1] * self.k[2])

    @property
    def bytes(self):
        """Algorithmic bytes of one pass (forward, dgrad or wgrad alike): input + output activations + weights, each moved
        once — what a launch must touch at least; the measured L2-miss traffic is priced against it."""
        taps = self.k[0] * self.k[1] * self.k[2]
        cin = self.Cin_alg or self.Cin
        return int(4 * (self.N * self.Di * self.Hi * self.Wi * cin + self.rows * self.Cout + self.Cout * cin * taps))

    def desc(self, in_ld=None, out_ld=None) -> _lib.ConvDesc:
        do, ho, wo = self.out_dims
        return _lib.ConvDesc(self.N, self.Di, self.Hi, self.Wi, self.Cin, do, ho, wo, self.Cout, *self.k, *self.s, *self.p,
                             in_ld or self.Cin, out_ld or self.Cout)


@dataclass(frozen=True)
class PoolGeom:
    """Geometry of the (optional) MaxPool3d fused behind BN+ReLU. k=s=(1,1,1), p=0 means no pooling."""
    N: int
    Di: int
    Hi: int
    Wi: int
    C: int
    k: Tuple[int, int, int] = (1, 1, 1)
    s: Tuple[int, int, int] = (1, 1, 1)
    p: Tuple[int, int, int] = (0, 0, 0)

    @property
    def out_dims(self):
        return tuple((i + 2 * p - k) // s + 1 for i, k, s, p in zip((self.Di, self.Hi, self.Wi), self.k, self.s, self.p))

    def desc(self, in_ld=None, out_ld=None, res_ld=None) -> _lib.PoolDesc:
        do, ho, wo = self.out_dims
        return _lib.PoolDesc(self.N, self.Di, self.Hi, self.Wi, self.C, do, ho, wo, *self.k, *self.s, *self.p,
                             in_ld or self.C, out_ld or self.C, res_ld or self.C)


import functools


@functools.lru_cache(maxsize=4096)
def _conv_plan(lib_id, g: "ConvGeom", in_ld, out_ld):
    """ctypes descriptor + geometry-derived sizes of one conv application, built once per distinct geometry (the
    per-launch host cost matters for the small-layer backbones: ~700-2500 launches per step)."""
    lib = _lib.load()
    d = g.desc(in_ld=in_ld, out_ld=out_ld)
    ref = C.byref(d)
    names = tuple(lib.rsp_conv3d_kernel_name(ref, which).decode() for which in (0, 1, 2))
    return (d, ref, lib.rsp_conv3d_stat_tiles(ref), lib.rsp_conv3d_fwd_workspace(ref), lib.rsp_conv3d_dgrad_workspace(ref),
            lib.rsp_conv3d_wgrad_workspace(ref), g.out_dims, names)


@functools.lru_cache(maxsize=4096)
def executed_fraction(g: "ConvGeom", which: int) -> float:
    """Share of the algorithmic MACs of one pass (0 forward, 1 dgrad, 2 wgrad) the kernels execute after skipping padding chunks
    (rsp_conv3d_executed_fraction; host arithmetic, measurement bookkeeping only)."""
    d = g.desc()
    return float(_lib.load().rsp_conv3d_executed_fraction(C.byref(d), which))


@functools.lru_cache(maxsize=8192)
def _pack_signature(g: "ConvGeom", which: int, cout_src: int, cin_src: int):
    lib = _lib.load()
    d = g.desc()
    buf = (_lib.PackJob * 64)()
    cnt = lib.rsp_conv3d_pack_jobs(C.byref(d), which, cout_src, cin_src, C.c_void_p(1 << 20), C.c_void_p(1 << 30), buf, 64)
    lib.rsp_conv3d_pack_forget(C.c_void_p(1 << 30))      # (a probe with dummy pointers: nothing to remember about that address)
    if cnt < 0:
        _lib.check(cnt, "rsp_conv3d_pack_jobs")
    n = (lib.rsp_conv3d_packed_dgrad_elems if which else lib.rsp_conv3d_packed_fwd_elems)(C.byref(d))
    return (int(n), bytes(buf)[:cnt * C.sizeof(_lib.PackJob)])


@functools.lru_cache(maxsize=4096)
def _pool_plan(lib_id, pg: "PoolGeom", in_ld, out_ld, res_ld):
    lib = _lib.load()
    d = pg.desc(in_ld=in_ld, out_ld=out_ld, res_ld=res_ld)
    ref = C.byref(d)
    return d, ref, lib.rsp_bn_bwd_workspace(ref), pg.out_dims


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_get_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current HIP stream of the current device as a raw handle.  (torch.cuda.current_stream() builds a Stream object
    through three Python layers: 7 us a call, 8 ms of host time per S3D-G step at ~1 150 kernel calls; the raw query is the same
    lookup without the wrapper.)"""
    if _raw_stream is not None and _get_device is not None:
        return C.c_void_p(_raw_stream(_get_device()))
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _chk(t: torch.Tensor, name: str, dtype=torch.float32):
    if not t.is_cuda:
        raise _lib.RspError(f"{name}: expected a HIP device tensor (rspnet_amd has no CPU path)")
    if t.dtype != dtype:
        raise _lib.RspError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise _lib.RspError(f"{name}: expected a contiguous tensor")
    return t


def _rows_ld(t: torch.Tensor, name: str) -> int:
    """Channel pitch of an NDHWC tensor that is either contiguous or a channel slice of a contiguous one."""
    if not t.is_cuda or t.dtype != torch.float32:
        raise _lib.RspError(f"{name}: expected a float32 HIP device tensor (rspnet_amd has no CPU path)")
    if t.is_contiguous():
        return t.shape[-1]
    ld = t.stride(-2)
    exp, ok = ld, t.stride(-1) == 1
    for dim in range(t.dim() - 2, -1, -1):
        ok = ok and (t.stride(dim) == exp or t.shape[dim] == 1)
        exp *= t.shape[dim]
    if not ok:
        raise _lib.RspError(f"{name}: expected a dense NDHWC tensor or a channel slice of one")
    return ld


class HipOps:
    """Calls into librspnet_hip.so on the current HIP stream of the current device."""

    name = "hip"

    def __init__(self):
        self.lib = _lib.load(init_gpu=True)
        self._ws = {}
        # geometry -> [row-geometry table of the weight-gradient kernels, event behind its fill | None, raw handle of the fill
        # stream, pinned by a captured graph?] (see _rowgeom_table), least recently used first
        self._rowgeom = collections.OrderedDict()
        self._rowgeom_bytes = 0
        self._rowgeom_streams = {}      # raw handle -> stream object of every stream a table was handed to (for eviction)
        # bench.py sets this to a list to collect (kind, algorithmic_flops, start_event, end_event, kernel name, algorithmic
        # bytes, geometry) per MFMA launch
        # group, recorded on the stream the kernels run on (torch's current stream).
        self.event_log = None
        self.hbm_log = None      # (kind, algorithmic bytes, start, end) per streaming launch group, with event_log

    def _ev(self):
        if self.event_log is None:
            return None
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        return e

    def _log(self, kind, g, e0, kernel=""):
        # kernel = the template instance the library dispatches for this launch (rsp_conv3d_kernel_name)
        if e0 is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            which = {"conv_fwd": 0, "conv_dgrad": 1, "conv_wgrad": 2}[kind]
            self.event_log.append((kind, g.flops, e0, e1, kernel, g.bytes, g, g.flops * executed_fraction(g, which)))

    def _log_hbm(self, kind, nbytes, e0):
        """bench.py's roofline pass: a streaming (HBM-bound) launch group with its algorithmic bytes — every operand tensor moved
        once."""
        if e0 is not None and self.hbm_log is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.hbm_log.append((kind, int(nbytes), e0, e1))

    # one grow-only scratch buffer per (device, stream): kernels on one stream are serialised, so they can share it; work issued
    # on different streams (independent branches of a layer graph) must not
    def _workspace(self, dev, nbytes: int) -> torch.Tensor:
        if torch.cuda.is_current_stream_capturing():
            # inside a HIP-graph capture the scratch comes from the graph's own memory pool, per call: it lives exactly as long as
            # the graph (a cached buffer would outlive a destroyed graph's pool, or be freed under a live graph when it grows),
            # and the pool's stream-aware reuse keeps kernels on forked streams apart
            return torch.empty(max(int(nbytes), 256), dtype=torch.uint8, device=dev)
        key = (dev, _stream().value)
        buf = self._ws.get(key)
        if buf is None or buf.numel() < nbytes:
            if buf is not None:
                buf.record_stream(torch.cuda.current_stream(dev))
            buf = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=dev)
            self._ws[key] = buf
        return buf

    def set_option(self, name: str, value: int) -> int:
        """rsp_conv3d_set_option: a planning option of the convolution launchers (process-wide; e.g. "narrow_max_tiles").  Returns
        the previous value.  The per-geometry plan cache (kernel names, workspace sizes) is dropped with it."""
        prev = self.lib.rsp_conv3d_set_option(name.encode(), int(value))
        if prev < 0:
            _lib.check(prev, "rsp_conv3d_set_option")
        _conv_plan.cache_clear()
        executed_fraction.cache_clear()
        return prev

    # ---- conv -------------------------------------------------------------------------------------------------
    def conv_pack_fwd(self, g: ConvGeom, w_ref: torch.Tensor) -> torch.Tensor:
        _chk(w_ref, "w_ref")
        d = g.desc()
        n = self.lib.rsp_conv3d_packed_fwd_elems(C.byref(d))
        out = torch.empty(n, dtype=torch.float32, device=w_ref.device)
        _lib.check(self.lib.rsp_conv3d_pack_fwd(C.byref(d), _ptr(w_ref), _ptr(out), _stream()), "rsp_conv3d_pack_fwd")
        return out

    def conv_fwd(self, g: ConvGeom, x, w_packed, bias, want_stats: bool, out: Optional[torch.Tensor] = None,
                 out_ld: Optional[int] = None, in_ld: Optional[int] = None):
        _chk(x, "x")
        d, dref, tiles, wsb, _, _, (do, ho, wo), names = _conv_plan(0, g, in_ld, out_ld)
        if out is None:
            out = torch.empty((g.N, do, ho, wo, g.Cout), dtype=torch.float32, device=x.device)
        stats = None
        if want_stats:
            stats = torch.empty((tiles, g.Cout, 2), dtype=torch.float32, device=x.device)
        ws = self._workspace(x.device, wsb) if wsb else None
        e0 = self._ev()
        _lib.check(self.lib.rsp_conv3d_fwd(dref, _ptr(x), _ptr(w_packed), _ptr(bias), _ptr(out), _ptr(stats),
                                           _ptr(ws), wsb, _stream()), "rsp_conv3d_fwd")
        self._log("conv_fwd", g, e0, names[0])
        return out, stats

    def conv_dgrad(self, g: ConvGeom, dy, w_ref):
        _chk(dy, "dy")
        _chk(w_ref, "w_ref")
        d, dref, _, _, wsb, _, _, names = _conv_plan(0, g, None, None)
        dx = torch.empty((g.N, g.Di, g.Hi, g.Wi, g.Cin), dtype=torch.float32, device=dy.device)
        ws = self._workspace(dy.device, wsb)
        e0 = self._ev()
        _lib.check(self.lib.rsp_conv3d_dgrad(dref, _ptr(dy), _ptr(w_ref), _ptr(dx), _ptr(ws), wsb, _stream()),
                   "rsp_conv3d_dgrad")
        self._log("conv_dgrad", g, e0, names[1])
        return dx

    def conv_dgrad_packed(self, g: ConvGeom, dy, w_packed):
        """dgrad over weights re-packed by a PackSet (which=1 entry)."""
        _chk(dy, "dy")
        d, dref, _, _, wsb, _, _, names = _conv_plan(0, g, None, None)
        dx = torch.empty((g.N, g.Di, g.Hi, g.Wi, g.Cin), dtype=torch.float32, device=dy.device)
        ws = self._workspace(dy.device, wsb)
        e0 = self._ev()
        _lib.check(self.lib.rsp_conv3d_dgrad_packed(dref, _ptr(dy), _ptr(_chk(w_packed, "w_packed")), _ptr(dx), _ptr(ws), wsb,
                                                    _stream()), "rsp_conv3d_dgrad_packed")
        self._log("conv_dgrad", g, e0, names[1])
        return dx

    def pack_signature(self, g: ConvGeom, which: int, w_ref):
        """Hashable identity of the packed layout `pack_set` would build for (g, which): the library's own job records with the
        pointers replaced by fixed dummies — equal for every input geometry that shares the layout."""
        return _pack_signature(g, which, int(w_ref.shape[0]), int(w_ref.shape[1]))

    def pack_set(self, entries):
        """entries: list of (ConvGeom, which, w_ref) with which = 0 (forward layout) / 1 (dgrad layouts) and w_ref the LIVE
        reference-layout weight tensor (it may have fewer channels than the geometry: zero padding).  Allocates the packed
        buffers and the device job table once; PackSet.run() re-packs all of them with one launch."""
        return PackSet(self, entries)

    def conv_wgrad(self, g: ConvGeom, x, dy, dw_out: torch.Tensor, dbias_out: Optional[torch.Tensor] = None):
        """dw_out (Cout,Cin,kT,kH,kW) and dbias_out are written in place (they are views of the flat grad buffer)."""
        _chk(x, "x")
        _chk(dw_out, "dw_out")
        dy_ld = _rows_ld(dy, "dy")                # dy may be a channel slice of a wider gradient tensor
        d, dref, _, _, _, wsb, _, names = _conv_plan(0, g, None, None if dy_ld == g.Cout else dy_ld)
        ws = self._workspace(x.device, wsb)
        if dw_out.shape[0] > g.Cout or dw_out.shape[1] > g.Cin or (dbias_out is not None and tuple(dw_out.shape[:2]) != (g.Cout, g.Cin)):
            raise _lib.RspError("conv_wgrad: dw_out has more channels than the geometry, or a bias gradient with padded channels")
        table = self._rowgeom_table(g, d, dref, x.device)
        e0 = self._ev()
        # (a geometry with zero-padded channels the parameter does not have: their gradients are dropped by the reduce)
        _lib.check(self.lib.rsp_conv3d_wgrad_t(dref, _ptr(x), _ptr(dy), _ptr(dw_out), _ptr(dbias_out), dw_out.shape[0], dw_out.shape[1],
                                               _ptr(table), _ptr(ws), wsb, _stream()), "rsp_conv3d_wgrad_t")
        self._log("conv_wgrad", g, e0, names[2])

    ROWGEOM_MAX_TABLES = 128
    ROWGEOM_MAX_BYTES = 2 << 30

    def _rowgeom_table(self, g: ConvGeom, d, dref, dev):
        """The weight-gradient kernels' per-row geometry table of this geometry (rsp_conv3d_rowgeom): computed once and kept — it
        depends on the positions only, not on channels or data.  None where the table-free path runs (kernel dims above 8).

        The table is filled on the stream that first needs it (often the engine's side-task stream) and then handed to weight
        gradients on ANY stream: the fill's event travels with the entry, and a caller on another stream waits for it until it has
        completed once.  The cache is bounded (count and bytes, least recently used first; ~8 bytes per output position, 50 MB for
        C3D's conv1): an evicted table is released behind every stream a table was ever handed to; tables baked into a captured
        graph's kernel arguments are never evicted."""
        if max(g.k) > 8:
            return None
        key = (dev, g.N, g.Di, g.Hi, g.Wi, g.k, g.s, g.p, d.in_ld)
        ent = self._rowgeom.get(key)
        capturing = torch.cuda.is_current_stream_capturing()
        here = _stream().value
        if ent is None:
            if capturing:
                return None                    # (first seen inside a capture: this call computes its own, the cache fills on the next eager step)
            nbytes = int(self.lib.rsp_conv3d_rowgeom_bytes(dref))
            t = torch.empty(nbytes, dtype=torch.uint8, device=dev)
            _lib.check(self.lib.rsp_conv3d_rowgeom(dref, _ptr(t), _stream()), "rsp_conv3d_rowgeom")
            ev = torch.cuda.Event()
            ev.record()
            ent = self._rowgeom[key] = [t, ev, here, False]
            self._rowgeom_bytes += nbytes
            self._rowgeom_evict(dev)
        else:
            self._rowgeom.move_to_end(key)
            if ent[1] is not None and not capturing:      # (a capture starts behind a device-wide synchronize; event queries are illegal inside)
                if ent[1].query():
                    ent[1] = None              # the fill is complete: visible to every stream from now on
                elif here != ent[2]:
                    torch.cuda.current_stream(dev).wait_event(ent[1])
            if capturing:
                ent[3] = True
        if here not in self._rowgeom_streams and not capturing:
            self._rowgeom_streams[here] = torch.cuda.current_stream(dev)
        return ent[0]

    def _rowgeom_evict(self, dev):
        if len(self._rowgeom) <= self.ROWGEOM_MAX_TABLES and self._rowgeom_bytes <= self.ROWGEOM_MAX_BYTES:
            return
        for key in list(self._rowgeom)[:-1]:           # never the entry just added
            if len(self._rowgeom) <= self.ROWGEOM_MAX_TABLES and self._rowgeom_bytes <= self.ROWGEOM_MAX_BYTES:
                break
            t, _, _, pinned = self._rowgeom[key]
            if pinned:
                continue
            for st in self._rowgeom_streams.values():  # a weight gradient on any of them may still be reading it
                t.record_stream(st)
            self._rowgeom_bytes -= t.numel()
            del self._rowgeom[key]

    # ---- batch norm -------------------------------------------------------------------------------------------
    def bn_finalize(self, stats, count: int, conv_bias, gamma, beta, eps: float, momentum: float, running_mean,
                    running_var, batch_stats_out=None):
        """batch_stats_out ([2][c_valid] view of a BnEmaSet): report this pass's batch moments there instead of moving the
        running statistics (deferred update, rsp_bn_running_update)."""
        tiles, Cc, _ = stats.shape
        if stats.is_contiguous():
            _chk(stats, "stats")
            stat_ld = Cc
        else:      # channel slice of a wider convolution's partials
            if not stats.is_cuda or stats.dtype != torch.float32 or stats.stride(2) != 1 or stats.stride(1) != 2 or stats.stride(0) % 2:
                raise _lib.RspError("stats: expected [tiles][C][2] float32 partials or a channel slice of them")
            stat_ld = stats.stride(0) // 2
        mi = torch.empty((2, Cc), dtype=torch.float32, device=stats.device)
        ss = torch.empty((2, Cc), dtype=torch.float32, device=stats.device)
        wsb = int(self.lib.rsp_bn_finalize_workspace(tiles, Cc))
        ws = self._workspace(stats.device, wsb)
        # the parameter vectors may be shorter than the convolution's (zero-padded) channel count: gamma's length says how many
        c_valid = Cc if gamma is None else int(gamma.shape[0])
        _lib.check(self.lib.rsp_bn_finalize_x(_ptr(stats), tiles, Cc, c_valid, stat_ld, count, _ptr(conv_bias), _ptr(gamma), _ptr(beta),
                                              eps, momentum, _ptr(running_mean), _ptr(running_var), _ptr(batch_stats_out), _ptr(mi),
                                              _ptr(ss), _ptr(ws), wsb, _stream()), "rsp_bn_finalize")
        return mi, ss

    def bn_ema_set(self, entries):
        """entries: list of (running_mean, running_var, momentum) of BatchNorm layers whose running-statistics update is deferred;
        returns a BnEmaSet: `.stats[i]` is the [2][C] buffer layer i's bn_finalize reports into, `.run()` applies all updates."""
        return BnEmaSet(self, entries)

    def bn_act_pool_fwd(self, pg: PoolGeom, y, scale_shift, residual, relu: bool, out=None):
        in_ld = _rows_ld(y, "y")             # y may be a channel slice of a wider conv output
        if out is None:
            do, ho, wo = pg.out_dims
            out = torch.empty((pg.N, do, ho, wo, pg.C), dtype=torch.float32, device=y.device)
        d, dref, _, _ = _pool_plan(0, pg, in_ld, _rows_ld(out, "out"), None if residual is None else _rows_ld(residual, "residual"))
        e0 = self._ev()
        _lib.check(self.lib.rsp_bn_act_pool_fwd(dref, _ptr(y), _ptr(scale_shift), _ptr(residual), int(relu), _ptr(out),
                                                _stream()), "rsp_bn_act_pool_fwd")
        n_in = pg.N * pg.Di * pg.Hi * pg.Wi * pg.C
        self._log_hbm("bn_act_pool_fwd", 4 * (n_in * (1 if residual is None else 2) + out.numel()), e0)
        return out

    def bn_act_maxpool_fwd(self, pg: PoolGeom, y, scale_shift, relu: bool, keep: bool):
        """BatchNorm apply (+ ReLU) and an OVERLAPPING max-pool (the ResNet stems' 3x3x3 / 2 / 1) in one pass over the convolution
        output, with the arg-max a backward needs (keep) — (out, idx | None), or None when the shape is not covered
        (rsp_bn_act_maxpool_applicable; the caller runs bn_act_pool_fwd + maxpool_fwd then).  Same bits as those two."""
        in_ld = _rows_ld(y, "y")
        do, ho, wo = pg.out_dims
        d = pg.desc(in_ld=in_ld)
        if not self.lib.rsp_bn_act_maxpool_applicable(C.byref(d)) or y.data_ptr() % 16 or scale_shift.data_ptr() % 16:
            return None
        out = torch.empty((pg.N, do, ho, wo, pg.C), dtype=torch.float32, device=y.device)
        idx = torch.empty((pg.N, do, ho, wo, pg.C), dtype=torch.int32, device=y.device) if keep else None
        e0 = self._ev()
        _lib.check(self.lib.rsp_bn_act_maxpool_fwd(C.byref(d), _ptr(y), _ptr(scale_shift), int(relu), _ptr(out), _ptr(idx), _stream()),
                   "rsp_bn_act_maxpool_fwd")
        self._log_hbm("bn_act_pool_fwd", 4 * (pg.N * pg.Di * pg.Hi * pg.Wi * pg.C + out.numel() * (2 if keep else 1)), e0)
        return out, idx

    def bn_act_pool_bwd(self, pg: PoolGeom, y, residual, dout, gamma, mean_invstd, scale_shift, relu: bool,
                        want_dres: bool, dgamma_out, dbeta_out, dy_out=None):
        """dy_out: where to write dy (same channel pitch as y — a slice of a wider gradient tensor when y is a slice)."""
        in_ld = _rows_ld(y, "y")
        d, dref, wsb, _ = _pool_plan(0, pg, in_ld, _rows_ld(dout, "dout"), None if residual is None else _rows_ld(residual, "residual"))
        if dy_out is not None:
            if _rows_ld(dy_out, "dy_out") != in_ld or dy_out.shape != y.shape:
                raise _lib.RspError("dy_out: expected the shape and channel pitch of y")
            dy = dy_out
        else:
            if not y.is_contiguous():
                raise _lib.RspError("y is a channel slice: pass dy_out with the same pitch")
            dy = torch.empty_like(y)
        dres = torch.empty_like(y) if want_dres else None
        ws = self._workspace(y.device, wsb)
        c_valid = pg.C if gamma is None else int(gamma.shape[0])      # < C: zero-padded channels (gamma / dgamma / dbeta are short)
        e0 = self._ev()
        _lib.check(self.lib.rsp_bn_act_pool_bwd_v(dref, _ptr(y), _ptr(residual), _ptr(dout), _ptr(gamma),
                                                  _ptr(mean_invstd), _ptr(scale_shift), int(relu), _ptr(dy), _ptr(dres),
                                                  _ptr(dgamma_out), _ptr(dbeta_out), c_valid, _ptr(ws), wsb, _stream()),
                   "rsp_bn_act_pool_bwd")
        # reduce pass: y (+ residual) + dout read; apply pass: the same again, dy (+ dres) written
        n_in = y.numel()
        self._log_hbm("bn_bwd(reduce+apply)", 4 * (2 * (n_in * (1 if residual is None else 2) + dout.numel()) + n_in * (2 if want_dres else 1)), e0)
        return dy, dres

    # ---- stand-alone pooling / gating --------------------------------------------------------------------------
    def maxpool_fwd(self, pg: PoolGeom, x, keep: bool):
        do, ho, wo = pg.out_dims
        out = torch.empty((pg.N, do, ho, wo, pg.C), dtype=torch.float32, device=x.device)
        idx = torch.empty((pg.N, do, ho, wo, pg.C), dtype=torch.int32, device=x.device) if keep else None
        d = pg.desc(in_ld=_rows_ld(x, "x"))
        _lib.check(self.lib.rsp_maxpool3d_fwd(C.byref(d), _ptr(x), _ptr(out), _ptr(idx), _stream()), "rsp_maxpool3d_fwd")
        return out, idx

    def maxpool_bwd(self, pg: PoolGeom, dout, idx):
        dx = torch.empty((pg.N, pg.Di, pg.Hi, pg.Wi, pg.C), dtype=torch.float32, device=dout.device)
        d = pg.desc(out_ld=_rows_ld(dout, "dout"))
        _lib.check(self.lib.rsp_maxpool3d_bwd(C.byref(d), _ptr(dout), _ptr(idx), _ptr(dx), _stream()), "rsp_maxpool3d_bwd")
        return dx

    def gate_fwd(self, x, w, b, out=None):
        _chk(x, "x")
        N, D, H, W, Cc = x.shape
        P = D * H * W
        if out is None:
            out = torch.empty_like(x)
        mean = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        gate = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        wsb = self.lib.rsp_gate_fwd_workspace(N, P, Cc)
        ws = self._workspace(x.device, wsb)
        _lib.check(self.lib.rsp_gate_fwd(_ptr(x), N, P, Cc, Cc, _ptr(_chk(w, "w")), _ptr(_chk(b, "b")), _ptr(out),
                                         _rows_ld(out, "out"), _ptr(mean), _ptr(gate), _ptr(ws), wsb, _stream()),
                   "rsp_gate_fwd")
        return out, mean, gate

    gate_pool_keep = True      # bn_act_gate_fwd(pool=..., pool_idx=True): the pooled form can also write the pool's arg-max

    def bn_act_gate_fwd(self, pg: PoolGeom, y, scale_shift, relu: bool, w, b, keep_act: bool, pool: Optional[PoolGeom] = None,
                        out=None, pool_idx: bool = False):
        """BatchNorm-apply (+ReLU) and the self-gating unit behind it as one op (models/s3dg.py:52-72; pg: the unit-window
        geometry of y).  a = act(y*scale + shift); gate = sigmoid(W mean(a) + b); out = a * gate — through `pool` (a max-pool
        geometry over a's dims, models/s3dg.py:105-109) when given.  keep_act: also materialise a for the backward (then no pool).
        Returns (out, a | None, mean, gate).  Two passes over y instead of the five tensor passes of bn_act_pool_fwd + gate_fwd."""
        N, P, Cc = pg.N, pg.Di * pg.Hi * pg.Wi, pg.C
        y_ld = _rows_ld(y, "y")
        if keep_act and pool is not None:
            raise _lib.RspError("bn_act_gate_fwd: the pooled form keeps no activation")
        a = torch.empty((N, pg.Di, pg.Hi, pg.Wi, Cc), dtype=torch.float32, device=y.device) if keep_act else None
        mean = torch.empty((N, Cc), dtype=torch.float32, device=y.device)
        gate = torch.empty((N, Cc), dtype=torch.float32, device=y.device)
        wsb = self.lib.rsp_gate_fwd_workspace(N, P, Cc)
        ws = self._workspace(y.device, wsb)
        _lib.check(self.lib.rsp_bn_gate_sums(_ptr(y), N, P, Cc, y_ld, _ptr(scale_shift), int(relu), _ptr(a), Cc, _ptr(_chk(w, "w")),
                                             _ptr(_chk(b, "b")), _ptr(mean), _ptr(gate), _ptr(ws), wsb, _stream()), "rsp_bn_gate_sums")
        og = pool if pool is not None else pg
        if out is None:
            do, ho, wo = og.out_dims
            out = torch.empty((N, do, ho, wo, Cc), dtype=torch.float32, device=y.device)
        if keep_act:
            _lib.check(self.lib.rsp_gate_apply(_ptr(a), N, P, Cc, Cc, _ptr(gate), _ptr(out), _rows_ld(out, "out"), _stream()),
                       "rsp_gate_apply")
        elif pool_idx:
            # the pooled form for a forward a backward follows: the max-pool's arg-max written by the same pass (returned fifth);
            # the caller has checked bn_act_gate_pool_idx_ok
            d, dref, _, _ = _pool_plan(0, og, y_ld, _rows_ld(out, "out"), None)
            idx = torch.empty(tuple(out.shape), dtype=torch.int32, device=y.device)
            _lib.check(self.lib.rsp_bn_act_maxpool_gate_fwd(dref, _ptr(y), _ptr(scale_shift), int(relu), _ptr(gate), _ptr(out), _ptr(idx),
                                                            _stream()), "rsp_bn_act_maxpool_gate_fwd")
            return out, a, mean, gate, idx
        else:
            d, dref, _, _ = _pool_plan(0, og, y_ld, _rows_ld(out, "out"), None)
            _lib.check(self.lib.rsp_bn_act_pool_gate_fwd(dref, _ptr(y), _ptr(scale_shift), None, int(relu), _ptr(gate), _ptr(out),
                                                         _stream()), "rsp_bn_act_pool_gate_fwd")
        return out, a, mean, gate

    def bn_act_gate_pool_idx_ok(self, pool: PoolGeom, y, scale_shift) -> bool:
        """Whether bn_act_gate_fwd(pool=pool, pool_idx=True) is available for this shape (rsp_bn_act_maxpool_applicable)."""
        d = pool.desc(in_ld=_rows_ld(y, "y"))
        return bool(self.lib.rsp_bn_act_maxpool_applicable(C.byref(d))) and y.data_ptr() % 16 == 0 and scale_shift.data_ptr() % 16 == 0

    def bn_act_gate_bwd(self, pg: PoolGeom, y, dout, gamma, mean_invstd, scale_shift, relu: bool, w, mean, gate,
                        dgamma_out, dbeta_out, dw_out, db_out):
        """Backward of bn_act_gate_fwd(keep_act=False): dout is the gradient of the gated output (possibly a channel slice of
        a concat gradient).  Gate parameter gradients into dw_out / db_out, BatchNorm's into dgamma_out / dbeta_out; returns dy.
        The activation is recomputed from y, and the gate's data gradient dout*gate + dmean/P is formed inside the BatchNorm
        backward kernels: seven tensor passes instead of the nine of gate_bwd + bn_act_pool_bwd."""
        N, P, Cc = pg.N, pg.Di * pg.Hi * pg.Wi, pg.C
        _chk(y, "y")
        dout_ld = _rows_ld(dout, "dout")
        dmean = torch.empty((N, Cc), dtype=torch.float32, device=y.device)
        wsb = self.lib.rsp_gate_bwd_workspace(N, P, Cc)
        ws = self._workspace(y.device, wsb)
        _lib.check(self.lib.rsp_gate_bwd_params(_ptr(y), _ptr(scale_shift), int(relu), _ptr(dout), N, P, Cc, Cc, dout_ld, _ptr(_chk(w, "w")),
                                                _ptr(mean), _ptr(gate), _ptr(dw_out), _ptr(db_out), _ptr(dmean), _ptr(ws), wsb,
                                                _stream()), "rsp_gate_bwd_params")
        d, dref, wsb2, _ = _pool_plan(0, pg, Cc, dout_ld, None)
        ws2 = self._workspace(y.device, wsb2)
        dy = torch.empty_like(y)
        c_valid = pg.C if gamma is None else int(gamma.shape[0])
        _lib.check(self.lib.rsp_bn_act_pool_bwd_g(dref, _ptr(y), None, _ptr(dout), _ptr(gamma), _ptr(mean_invstd), _ptr(scale_shift),
                                                  int(relu), _ptr(dy), None, _ptr(dgamma_out), _ptr(dbeta_out), c_valid, _ptr(gate),
                                                  _ptr(dmean), _ptr(ws2), wsb2, _stream()), "rsp_bn_act_pool_bwd_g")
        return dy

    def gate_bwd(self, x, dout, w, mean, gate, dw_out, db_out):
        _chk(x, "x")
        N, D, H, W, Cc = x.shape
        P = D * H * W
        dx = torch.empty_like(x)
        wsb = self.lib.rsp_gate_bwd_workspace(N, P, Cc)
        ws = self._workspace(x.device, wsb)
        _lib.check(self.lib.rsp_gate_bwd(_ptr(x), _ptr(dout), N, P, Cc, Cc, _rows_ld(dout, "dout"), _ptr(w), _ptr(mean),
                                         _ptr(gate), _ptr(dx), Cc, _ptr(dw_out), _ptr(db_out), _ptr(ws), wsb, _stream()),
                   "rsp_gate_bwd")
        return dx

    # ---- heads / contrastive ----------------------------------------------------------------------------------
    def head_fwd(self, feat, w1, b1, w2, b2):
        _chk(feat, "feat")
        N, D, H, W, Cc = feat.shape
        P = D * H * W
        dim = w1.shape[0]
        dev = feat.device
        o1 = torch.empty((N, dim), dtype=torch.float32, device=dev)
        o2 = torch.empty((N, dim), dtype=torch.float32, device=dev)
        pooled = torch.empty((N, Cc), dtype=torch.float32, device=dev)
        raw = torch.empty((2, N, dim), dtype=torch.float32, device=dev)
        _lib.check(self.lib.rsp_head_fwd(_ptr(feat), N, P, Cc, Cc, _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), dim, _ptr(o1),
                                         _ptr(o2), _ptr(pooled), _ptr(raw), _stream()), "rsp_head_fwd")
        return o1, o2, pooled, raw

    def head_bwd(self, d1, d2, pooled, raw, w1, w2, feat_shape, dw1, db1, dw2, db2):
        N, D, H, W, Cc = feat_shape
        P = D * H * W
        dim = w1.shape[0]
        dfeat = torch.empty(feat_shape, dtype=torch.float32, device=d1.device)
        wsb = self.lib.rsp_head_bwd_workspace(N, dim)
        ws = self._workspace(d1.device, wsb)
        _lib.check(self.lib.rsp_head_bwd(_ptr(_chk(d1, "d1")), _ptr(_chk(d2, "d2")), _ptr(pooled), _ptr(raw), _ptr(w1),
                                         _ptr(w2), N, P, Cc, Cc, dim, _ptr(dw1), _ptr(db1), _ptr(dw2), _ptr(db2),
                                         _ptr(dfeat), _ptr(ws), wsb, _stream()), "rsp_head_bwd")
        return dfeat

    # generic pieces of the 'mlp' head
    def spatial_mean_fwd(self, x):
        _chk(x, "x")
        N, D, H, W, Cc = x.shape
        mean = torch.empty((N, Cc), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.rsp_spatial_mean_fwd(_ptr(x), N, D * H * W, Cc, Cc, _ptr(mean), _stream()), "rsp_spatial_mean_fwd")
        return mean

    def spatial_mean_bwd(self, dmean, shape):
        N, D, H, W, Cc = shape
        dx = torch.empty(shape, dtype=torch.float32, device=dmean.device)
        _lib.check(self.lib.rsp_spatial_mean_bwd(_ptr(_chk(dmean, "dmean")), N, D * H * W, Cc, Cc, _ptr(dx), _stream()),
                   "rsp_spatial_mean_bwd")
        return dx

    def linear_fwd(self, x, w, b, relu: bool):
        B, Cin = x.shape
        y = torch.empty((B, w.shape[0]), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.rsp_linear_fwd(_ptr(_chk(x, "x")), B, Cin, _ptr(_chk(w, "w")), _ptr(b), w.shape[0], int(relu), _ptr(y),
                                           _stream()), "rsp_linear_fwd")
        return y

    def linear_bwd(self, x, y, dy, w, relu: bool, dw_out, db_out, want_dx=True):
        B, Cin = x.shape
        Cout = w.shape[0]
        dx = torch.empty_like(x) if want_dx else None
        wsb = self.lib.rsp_linear_bwd_workspace(B, Cout)
        ws = self._workspace(x.device, wsb)
        _lib.check(self.lib.rsp_linear_bwd(_ptr(x), _ptr(y), _ptr(_chk(dy, "dy")), _ptr(w), B, Cin, Cout, int(relu), _ptr(dx),
                                           _ptr(dw_out), _ptr(db_out), _ptr(ws), wsb, _stream()), "rsp_linear_bwd")
        return dx

    def l2norm_fwd(self, x):
        y = torch.empty_like(x)
        _lib.check(self.lib.rsp_l2norm_fwd(_ptr(_chk(x, "x")), x.shape[0], x.shape[1], _ptr(y), _stream()), "rsp_l2norm_fwd")
        return y

    def l2norm_bwd(self, x, dy):
        dx = torch.empty_like(x)
        _lib.check(self.lib.rsp_l2norm_bwd(_ptr(x), _ptr(_chk(dy, "dy")), x.shape[0], x.shape[1], _ptr(dx), _stream()),
                   "rsp_l2norm_bwd")
        return dx

    def logits_fwd(self, qA, qM, kA, kM, knegA, knegM, queue, inv_T: float):
        B, dim = qA.shape
        K = queue.shape[1]
        dev = qA.device
        for n, t in (("qA", qA), ("qM", qM), ("kA", kA), ("kM", kM), ("knegA", knegA), ("knegM", knegM), ("queue", queue)):
            _chk(t, n)
        l1 = torch.empty((B, K + 1), dtype=torch.float32, device=dev)
        l2 = torch.empty((B, K + 1), dtype=torch.float32, device=dev)
        lp = torch.empty((B, 1), dtype=torch.float32, device=dev)
        ln = torch.empty((B, 1), dtype=torch.float32, device=dev)
        _lib.check(self.lib.rsp_logits_fwd(_ptr(qA), _ptr(qM), _ptr(kA), _ptr(kM), _ptr(knegA), _ptr(knegM), _ptr(queue), B,
                                           dim, qM.shape[1], K, inv_T, _ptr(l1), _ptr(l2), _ptr(lp), _ptr(ln), _stream()),
                   "rsp_logits_fwd")
        return l1, l2, lp, ln

    def logits_bwd(self, dl1, dl2, dlp, dln, kA, kM, knegA, knegM, queue, inv_T: float):
        B, dim = kA.shape
        K = queue.shape[1]
        dev = kA.device
        dqA = torch.empty((B, dim), dtype=torch.float32, device=dev)
        dqM = torch.empty((B, kM.shape[1]), dtype=torch.float32, device=dev)
        wsb = self.lib.rsp_logits_bwd_workspace(B, dim, K)
        ws = self._workspace(dev, wsb)
        _lib.check(self.lib.rsp_logits_bwd(_ptr(_chk(dl1, "dl1")), _ptr(_chk(dl2, "dl2")), _ptr(_chk(dlp, "dlp")),
                                           _ptr(_chk(dln, "dln")), _ptr(kA), _ptr(kM), _ptr(knegA), _ptr(knegM), _ptr(queue),
                                           B, dim, kM.shape[1], K, inv_T, _ptr(dqA), _ptr(dqM), _ptr(ws), wsb, _stream()),
                   "rsp_logits_bwd")
        return dqA, dqM

    def loss_fwd_bwd(self, l1, l2, lp, ln, margin: float, A: float, M: float):
        B, K1 = l1.shape
        dev = l1.device
        for n, t in (("logits1", l1), ("logits2", l2), ("l_pos_M", lp), ("l_neg_M", ln)):
            _chk(t, n)
        losses = torch.empty(3, dtype=torch.float32, device=dev)
        d1 = torch.empty_like(l1)
        d2 = torch.empty_like(l2)
        dp = torch.empty_like(lp)
        dn = torch.empty_like(ln)
        scratch = torch.empty(2 * B, dtype=torch.float32, device=dev)
        _lib.check(self.lib.rsp_loss_fwd_bwd(_ptr(l1), _ptr(l2), _ptr(lp), _ptr(ln), B, K1, margin, A, M, _ptr(losses),
                                             _ptr(d1), _ptr(d2), _ptr(dp), _ptr(dn), _ptr(scratch), _stream()),
                   "rsp_loss_fwd_bwd")
        return losses, d1, d2, dp, dn

    def queue_enqueue(self, queue, ptr: int, keys):
        dim, K = queue.shape
        _lib.check(self.lib.rsp_queue_enqueue(_ptr(_chk(queue, "queue")), dim, K, ptr, _ptr(_chk(keys, "keys")),
                                              keys.shape[0], _stream()), "rsp_queue_enqueue")

    def queue_enqueue_dev(self, queue, queue_ptr, keys):
        """queue[:, ptr:ptr+n] = keys.T with ptr read from / advanced in the device buffer `queue_ptr` (int64, 1 element)."""
        dim, K = queue.shape
        _chk(queue_ptr, "queue_ptr", torch.int64)
        _lib.check(self.lib.rsp_queue_enqueue_dev(_ptr(_chk(queue, "queue")), dim, K, _ptr(queue_ptr), _ptr(_chk(keys, "keys")),
                                                  keys.shape[0], _stream()), "rsp_queue_enqueue_dev")

    # ---- glue -------------------------------------------------------------------------------------------------
    def clip_gather(self, im, src, step, T_out: int, c_out: Optional[int] = None):
        _chk(im, "im")
        _chk(src, "src", torch.int32)
        _chk(step, "step", torch.int32)
        B_in, Cc, T_in, H, W = im.shape
        B_out = src.shape[0]
        c_out = c_out or Cc
        out = torch.empty((B_out, T_out, H, W, c_out), dtype=torch.float32, device=im.device)
        e0 = self._ev()
        _lib.check(self.lib.rsp_clip_gather(_ptr(im), B_in, Cc, T_in, H, W, _ptr(src), _ptr(step), B_out, T_out, c_out, _ptr(out),
                                            _stream()), "rsp_clip_gather")
        self._log_hbm("clip_gather", 4 * (B_out * T_out * H * W * Cc + out.numel()), e0)
        return out

    def clip_gather_multi(self, jobs, T_out: int, c_out: Optional[int] = None):
        """jobs: [(im, src, step)] of ONE geometry (same clip shape, same number of output clips) -> their gathered clips, from one
        launch (rsp_clip_gather_multi)."""
        im0, src0, _ = jobs[0]
        B_in, Cc, T_in, H, W = im0.shape
        B_out = src0.shape[0]
        c_out = c_out or Cc
        outs = []
        for im, src, step in jobs:
            _chk(im, "im")
            _chk(src, "src", torch.int32)
            _chk(step, "step", torch.int32)
            if tuple(im.shape) != tuple(im0.shape) or src.shape[0] != B_out or step.shape[0] != B_out:
                raise _lib.RspError("clip_gather_multi: the jobs must share one geometry")
            outs.append(torch.empty((B_out, T_out, H, W, c_out), dtype=torch.float32, device=im.device))
        n = len(jobs)
        arr = C.c_void_p * n
        e0 = self._ev()
        _lib.check(self.lib.rsp_clip_gather_multi(n, arr(*[j[0].data_ptr() for j in jobs]), arr(*[j[1].data_ptr() for j in jobs]),
                                                  arr(*[j[2].data_ptr() for j in jobs]), arr(*[o.data_ptr() for o in outs]), B_in, Cc,
                                                  T_in, H, W, B_out, T_out, c_out, _stream()), "rsp_clip_gather_multi")
        self._log_hbm("clip_gather", 4 * n * (B_out * T_out * H * W * Cc + outs[0].numel()), e0)
        return outs

    def momentum_update(self, k_flat, q_flat, m: float):
        e0 = self._ev()
        _lib.check(self.lib.rsp_momentum_update(_ptr(_chk(k_flat, "k")), _ptr(_chk(q_flat, "q")), k_flat.numel(), m,
                                                _stream()), "rsp_momentum_update")
        self._log_hbm("momentum_update", 4 * 3 * k_flat.numel(), e0)

    def sgd_step(self, p, g, buf, lr: float, mu: float, wd: float, gscale: float, first: bool):
        e0 = self._ev()
        _lib.check(self.lib.rsp_sgd_step(_ptr(_chk(p, "p")), _ptr(_chk(g, "g")), _ptr(_chk(buf, "buf")), p.numel(), lr, mu,
                                         wd, gscale, int(first), _stream()), "rsp_sgd_step")
        self._log_hbm("sgd_step", 4 * 5 * p.numel(), e0)

    def rows_gather(self, x, idx):
        _chk(x, "x")
        _chk(idx, "idx", torch.int32)
        n, width = idx.shape[0], x.shape[1]
        out = torch.empty((n, width), dtype=torch.float32, device=x.device)
        _lib.check(self.lib.rsp_rows_gather(_ptr(x), _ptr(idx), n, width, _ptr(out), _stream()), "rsp_rows_gather")
        return out

    def eltwise(self, op: str, a, b=None, out=None):
        """op in relu_fwd / relu_bwd / sigmoid_fwd / sigmoid_bwd / add; dense tensors, `out` may alias an input."""
        code = {"relu_fwd": 0, "relu_bwd": 1, "sigmoid_fwd": 2, "sigmoid_bwd": 3, "add": 4}[op]
        _chk(a, "a")
        if b is not None:
            _chk(b, "b")
            if b.shape != a.shape:
                raise _lib.RspError(f"eltwise {op}: shape mismatch {tuple(a.shape)} vs {tuple(b.shape)}")
        if out is None:
            out = torch.empty_like(a)
        _lib.check(self.lib.rsp_eltwise(code, _ptr(a), _ptr(b), _ptr(_chk(out, "out")), a.numel(), _stream()), "rsp_eltwise")
        return out

    # ---- clip augmentation ---------------------------------------------------------------------------
    def augment_batch(self, descs, n_clips, T, size, mean, std, out, blur9=None):
        """descs: uint8 device tensor holding n_clips rsp_augment_clip_desc records; out: (n_clips, 3, T, size, size) f32."""
        _chk(descs, "descs", torch.uint8)
        _chk(out, "out")
        assert descs.numel() >= n_clips * C.sizeof(_lib.AugmentClipDesc) and out.is_contiguous()
        assert out.numel() == n_clips * 3 * T * size * size
        nbytes = int(self.lib.rsp_augment_workspace(n_clips, T, size))
        ws = self._workspace(out.device, nbytes)
        m3, s3 = (C.c_float * 3)(*mean), (C.c_float * 3)(*std)
        b9 = None if blur9 is None else (C.c_float * 9)(*blur9)
        _lib.check(self.lib.rsp_augment_batch(_ptr(descs), n_clips, T, size, m3, s3, b9, _ptr(out), 3 * T * size * size, _ptr(ws),
                                              ws.numel(), _stream()), "rsp_augment_batch")
        return out


class BnEmaSet:
    """Batch moments of a list of BatchNorm layers (one flat buffer) + the device-resident job table of their deferred
    running-statistics update (rsp_bn_running_update: one launch for all layers)."""

    def __init__(self, be: "HipOps", entries):
        self.be = be
        dev = entries[0][0].device
        sizes = [int(rm.shape[0]) for rm, _, _ in entries]
        self.flat = torch.zeros(2 * sum(sizes), dtype=torch.float32, device=dev)
        self.stats, self.ptrs, jobs, off = [], [], [], 0
        for (rm, rv, mom), c in zip(entries, sizes):
            _chk(rm, "running_mean")
            _chk(rv, "running_var")
            v = self.flat[off:off + 2 * c].view(2, c)
            self.stats.append(v)
            self.ptrs.append((rm.data_ptr(), rv.data_ptr()))
            jobs.append(bytes(_lib.BnEmaJob(rm.data_ptr(), rv.data_ptr(), v.data_ptr(), c, float(mom))))
            off += 2 * c
        self.keep = [e[:2] for e in entries]
        self.max_c, self.n = max(sizes), len(sizes)
        self.table = torch.frombuffer(bytearray(b"".join(jobs)), dtype=torch.uint8).to(dev)

    def run(self):
        _lib.check(self.be.lib.rsp_bn_running_update(_ptr(self.table), self.n, self.max_c, _stream()), "rsp_bn_running_update")


class PackSet:
    """Persistent packed copies of a set of convolution weights + the device-resident job tables that rebuild them.
    Forward-layout and dgrad-layout jobs run as two launches: packing both layouts of the SAME weight concurrently reads it in
    two conflicting patterns and is several times slower than the two passes back to back (tools/pack_probe.py: 31 + 62 us
    apart, 221 us together on a 512x512x27 filter)."""

    def __init__(self, be: "HipOps", entries):
        self.be = be
        self.packed = []
        jobs = {0: [], 1: []}
        self.max_blocks = {0: 0, 1: 0}
        dev = None
        for g, which, w_ref in entries:
            _chk(w_ref, "w_ref")
            dev = w_ref.device
            d = g.desc()
            n = (be.lib.rsp_conv3d_packed_dgrad_elems if which else be.lib.rsp_conv3d_packed_fwd_elems)(C.byref(d))
            out = torch.empty(n, dtype=torch.float32, device=dev)
            buf = (_lib.PackJob * 64)()
            cnt = be.lib.rsp_conv3d_pack_jobs(C.byref(d), which, w_ref.shape[0], w_ref.shape[1], _ptr(w_ref), _ptr(out), buf, 64)
            if cnt < 0:
                _lib.check(cnt, "rsp_conv3d_pack_jobs")
            kind = 1 if which else 0
            jobs[kind].extend(bytes(buf)[i * C.sizeof(_lib.PackJob):(i + 1) * C.sizeof(_lib.PackJob)] for i in range(cnt))
            self.max_blocks[kind] = max([self.max_blocks[kind]] + [int(buf[i].blocks) for i in range(cnt)])
            self.packed.append(out)
        self.sources = [e[2] for e in entries]            # keep the weights (and their storage) alive
        self.tables = {}
        for kind, js in jobs.items():
            if js:
                self.tables[kind] = (torch.frombuffer(bytearray(b"".join(js)), dtype=torch.uint8).to(dev), len(js))

    def run(self):
        for kind, (table, n) in self.tables.items():
            _lib.check(self.be.lib.rsp_pack_run(_ptr(table), n, self.max_blocks[kind], _stream()), "rsp_pack_run")

    def __del__(self):
        # the library keeps a per-address note about packed stem buffers (three real input channels): withdrawn with the buffers
        try:
            for t in self.packed:
                self.be.lib.rsp_conv3d_pack_forget(_ptr(t))
        except Exception:      # noqa: BLE001 - interpreter shutdown: the library may be gone
            pass


_backend = None


def backend():
    """The active op backend; instantiates HipOps (and therefore requires the built library) on first use."""
    global _backend
    if _backend is None:
        _backend = HipOps()
    return _backend


def set_backend(b):
    """TEST HOOK ONLY: swap the op backend (tests/cpu_ops.py).  Returns the previous one."""
    global _backend
    prev, _backend = _backend, b
    return prev
