"""Plan executor: runs a backbone's layer plan forward/backward through the HIP ops.

A backbone (rspnet_amd/models/*) is an ``nn.Module`` tree that only *holds* parameters under the reference's
state-dict names; its ``plan()`` lists fused units over numbered tensor slots:

  ConvBN   conv3d → BatchNorm3d(train) [→ + residual] [→ ReLU] [→ disjoint MaxPool3d]     (all four backbones)
  ConvBias conv3d + bias [→ ReLU], no BatchNorm                                            ('conv' projection head)
  Pool     stand-alone MaxPool3d with overlapping windows                                  (ResNet stem, S3D-G)
  Gate     S3D-G self-gating  x * sigmoid(W·mean(x) + b)
  Concat   channel concat of branch outputs (S3D-G inception)

Forward keeps, per ConvBN, only the conv input, the raw conv output y and the per-channel (mean, invstd, scale,
shift); the activation / ReLU mask / pool arg-max are recomputed from y in backward.
Activations are (N, D, H, W, C) tensors.
"""
from __future__ import annotations

import contextlib
import os
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import ops as _ops
from .ops import ConvGeom, PoolGeom

Triple = Tuple[int, int, int]


@dataclass
class ConvBN:
    conv: nn.Module                 # holds .weight (Cout,Cin,kT,kH,kW) [, .bias]
    bn: nn.Module                   # holds weight/bias/running_mean/running_var/num_batches_tracked, .eps, .momentum
    src: int
    dst: int
    k: Triple
    s: Triple = (1, 1, 1)
    p: Triple = (0, 0, 0)
    relu: bool = True
    pool: Optional[Tuple[Triple, Triple]] = None   # (kernel, stride), disjoint windows only
    residual: Optional[int] = None                 # slot added before the ReLU
    into: Optional[Tuple[int, int, int]] = None    # (concat slot, channel offset, total channels): write a slice
    branch: int = 0                                # > 0: node of a side branch that may run beside its siblings (BranchStreams)
    virtual_w: bool = False                        # 3-channel stem with W stride 2: run on "virtual pixels" (VirtualStem)
    cout_pad: int = 0                              # run with Cout zero-padded to this many channels (0: as is).  R(2+1)D's
    #   mid-channel counts (83, 230, 921 ...) are not multiples of 4; padded, this conv's output and the next conv's input are
    #   16-byte rows and both take the LDS-DMA kernels instead of the scalar gather.  Pad channels carry zero weights and
    #   gamma = beta = 0, so they are exactly 0 after BN+ReLU and contribute nothing downstream; their gradients are dropped.


@dataclass
class ConvBNGroup:
    """ConvBN units that read the SAME tensor with the same geometry (S3D-G's inception siblings branch0 / branch1.0 /
    branch2.0, models/s3dg.py:80-88): when their filters sit back to back in memory (rspnet_amd/flat.py `adjacent`) they run
    as ONE convolution over the concatenated filters — one GEMM instead of three in forward, one dgrad whose K runs over all
    members (the fan-out sum of their input gradients comes out of the GEMM, no adds) and one wgrad in backward.  BatchNorm
    stays per member (channel slices of the shared conv output / statistics).  With any other parameter layout (fine-tune
    path, stand-alone use) the members simply run one by one."""
    members: List[ConvBN]

    def _cat_node(self):
        """What PackedWeights keys on: an object whose .conv.weight is the concatenated filter tensor."""
        holder = getattr(self, "_holder", None)
        if holder is None:
            holder = self._holder = _CatHolder(self._cat)
        return holder


class _CatHolder:
    def __init__(self, conv):
        self.conv = conv


@dataclass
class ConvBias:
    """nn.Conv3d(bias=True) [+ ReLU] without BatchNorm: the two convs of ConvFc (moco/split_wrapper.py:18-39)."""
    conv: nn.Module
    src: int
    dst: int
    k: Triple
    s: Triple = (1, 1, 1)
    p: Triple = (0, 0, 0)
    relu: bool = False


@dataclass
class Pool:
    """Stand-alone MaxPool3d (overlapping / padded windows allowed)."""
    src: int
    dst: int
    k: Triple
    s: Triple
    p: Triple = (0, 0, 0)
    branch: int = 0


@dataclass
class Gate:
    """S3D-G self-gating: x * sigmoid(conv1x1x1(mean(x)))   (models/s3dg.py:63-72)."""
    conv: nn.Module                                # excitation: weight (C,C,1,1,1), bias (C)
    src: int
    dst: int
    into: Optional[Tuple[int, int, int]] = None
    branch: int = 0


@dataclass
class Plan:
    nodes: List[object]
    input_slot: int = 0
    output_slot: int = 0


@dataclass
class _Saved:
    x: torch.Tensor
    y: torch.Tensor
    mi: torch.Tensor
    ss: torch.Tensor
    cg: ConvGeom
    pg: PoolGeom
    res: Optional[torch.Tensor] = None


@dataclass
class ForwardCtx:
    saved: Dict[int, _Saved] = field(default_factory=dict)
    feat_shape: Optional[Tuple[int, ...]] = None
    packed: Optional["PackedWeights"] = None       # the encoder's weight cache (dgrad layouts are fetched from it in backward)


class PackedWeights:
    """Packed copies of one encoder's convolution weights: the forward layout of every conv and, for convs whose input needs
    a gradient, the dgrad layouts.  The weights change once per step, so all copies are rebuilt together by ONE batched launch
    (ops.PackSet) the first time one of them is needed after `invalidate()`; channel padding (3 -> 4 channel stems, R(2+1)D's
    odd mid-channel counts) is applied by the re-pack itself.  A conv seen for the first time is packed on its own and joins
    the batch at the next rebuild."""

    def __init__(self):
        self._entries = []        # (geometry, which, conv module): ONE per distinct packed layout of a conv
        self._index = {}          # (id(conv), which, layout signature) -> (set number, position)
        self._sets = []
        self._ptrs = []
        self._dirty = True
        self._virtual = {}        # id(conv) -> VirtualStem: derived filters that are re-gathered before every re-pack

    def invalidate(self):
        self._dirty = True

    def refresh_now(self):
        """Re-pack on the CURRENT stream if the weights changed: for callers that are about to use the copies from several
        streams (two passes through one encoder side by side) and need the re-pack ordered before the fork."""
        if self._dirty and self._entries:
            self._refresh()

    def _refresh(self):
        be = _ops.backend()
        for vs in self._virtual.values():
            vs.refresh()
        ptrs = [c.weight.data_ptr() for _, _, c in self._entries]
        if len(self._sets) > 1 or ptrs != self._ptrs:      # new members since the last rebuild, or the parameters moved
            self._sets = [be.pack_set([(g, which, c.weight.data) for g, which, c in self._entries])] if self._entries else []
            self._index = {(id(c), which, be.pack_signature(g, which, c.weight.data)): (0, i)
                           for i, (g, which, c) in enumerate(self._entries)}
            self._ptrs = ptrs
        for ps in self._sets:
            ps.run()
        self._dirty = False

    def _lookup(self, conv, cg: ConvGeom, which: int):
        if self._dirty:
            self._refresh()
        # the packed layout depends on channels / kernel / stride / padding / stem applicability only: every input geometry a
        # conv is applied to (diff_speed's T_real values, a partial last batch, eval sizes) shares one packed copy
        key = (id(conv), which, _ops.backend().pack_signature(cg, which, conv.weight.data))
        at = self._index.get(key)
        if at is None:
            ps = _ops.backend().pack_set([(cg, which, conv.weight.data)])
            ps.run()
            self._sets.append(ps)
            self._entries.append((cg, which, conv))
            self._ptrs.append(conv.weight.data_ptr())
            at = self._index[key] = (len(self._sets) - 1, 0)
        return self._sets[at[0]].packed[at[1]]

    def get(self, node, cg: ConvGeom):
        return self._lookup(node.conv, cg, 0)

    def get_dgrad(self, node, cg: ConvGeom):
        return self._lookup(node.conv, cg, 1)


class _CatConv:
    """Stands in for a conv module where the engine / PackedWeights expect `.weight`: the concatenated filters of a group."""
    weight: torch.Tensor = None


def _adjacent_cat(ts):
    """If the tensors sit back to back in one storage (flat parameter / gradient buffers, rspnet_amd/flat.py), the single
    tensor that covers them — dim 0 concatenated — else None."""
    t0 = ts[0]
    if not all(t.is_contiguous() and t.shape[1:] == t0.shape[1:] and t.dtype == t0.dtype for t in ts):
        return None
    end = t0.data_ptr() + t0.numel() * t0.element_size()
    for t in ts[1:]:
        if t.data_ptr() != end or t.untyped_storage().data_ptr() != t0.untyped_storage().data_ptr():
            return None
        end += t.numel() * t.element_size()
    shape = (sum(t.shape[0] for t in ts),) + tuple(t0.shape[1:])
    stride = tuple(t0.stride())
    return torch.as_strided(t0, shape, stride, t0.storage_offset())


def _pad_vec(v: torch.Tensor, n: int, fill: float = 0.0) -> torch.Tensor:
    out = torch.full((n,), fill, dtype=v.dtype, device=v.device)
    out[:v.shape[0]] = v
    return out


INPUT_CHANNEL_PAD = 4
GATE_BWD_FUSED = not os.environ.get("RSP_NO_GATE_BWD_FUSION")      # (A/B switch for measurements)


class VirtualStem:
    """A 3-channel stem convolution with kernel width 7 and stride 2 along W (R3D-18's 7x7x7 conv1, models/resnet.py:124; the
    (1,7,7) spatial half of R(2+1)D's conv1, models/r2plus1d_vcop.py:57) without the zero fourth channel.

    The implicit-GEMM kernel wants 16-byte pixels, so the clip is gathered with its 3 channels padded to 4 and a quarter of the
    matrix work multiplies zeros.  But an output column wo reads 7 pixels = 21 CONTIGUOUS floats of a packed 3-channel row, starting
    at float 6 wo - 9; seen as 16-byte "virtual pixels" (4 floats, 84 per 112-pixel row) that window starts 3 floats into virtual
    pixel 3g - 3 for wo = 2g and 1 float into virtual pixel 3g - 1 for wo = 2g + 1.  Each parity of wo is therefore an ORDINARY
    convolution over the virtual row — kernel width 6, stride 3, 4 "channels", its own weight tensor with the 21 real values shifted
    by 3 (even) or 1 (odd) floats inside the 24 — and writes every second output column (channel pitch 2 Cout).  K per output drops
    from 7 * 4 = 28 to 24 floats per kernel row: 14 % less matrix work on the layer that is a third of R3D-18's step.
    The packed row carries 3 zero virtual pixels on the left and 2 on the right, so both parities run without padding along W
    (the odd one from a base address 2 virtual pixels in) and the output-size formula gives exactly Wo / 2 columns."""

    def __init__(self, node):
        self.conv = node.conv
        w = node.conv.weight
        Cout, cin, kT, kH, kW = w.shape
        assert cin == 3 and kW == 7 and node.s[2] == 2 and node.p[2] == 3
        self.holders, self.idx, self.src, self.dst = [], [], [], []
        co = torch.arange(Cout).view(Cout, 1, 1, 1, 1)
        c4 = torch.arange(4).view(1, 4, 1, 1, 1)
        kt = torch.arange(kT).view(1, 1, kT, 1, 1)
        kh = torch.arange(kH).view(1, 1, 1, kH, 1)
        j = torch.arange(6).view(1, 1, 1, 1, 6)
        for off in (3, 1):                               # even wo, odd wo
            r = 4 * j + c4 - off                         # float inside the 21-float window
            valid = ((r >= 0) & (r < 21)).expand(Cout, 4, kT, kH, 6)
            real = ((((co * 3 + r % 3) * kT + kt) * kH + kh) * 7 + torch.div(r, 3, rounding_mode="floor")).expand(Cout, 4, kT, kH, 6)
            idx = torch.where(valid, real, torch.full_like(real, w.numel())).reshape(-1)      # sentinel -> the appended zero
            flat_valid = valid.reshape(-1).nonzero().view(-1)
            self.idx.append(idx.to(w.device))
            self.src.append(flat_valid.to(w.device))
            self.dst.append(real.reshape(-1)[flat_valid].to(w.device))
            h = _CatConv()
            h.weight = torch.zeros((Cout, 4, kT, kH, 6), dtype=w.dtype, device=w.device)
            self.holders.append(_CatHolder(h))
        self.device = w.device
        self.refresh()

    enabled = not os.environ.get("RSP_NO_VIRTUAL_STEM")      # (A/B switch for measurements)

    @staticmethod
    def applies(node, xin) -> bool:
        w = node.conv.weight
        W = xin.shape[3]
        return (VirtualStem.enabled and node.virtual_w and w.shape[1] == 3 and xin.shape[4] == 4 and w.shape[4] == 7 and node.s[2] == 2 and node.p[2] == 3
                and (3 * W) % 4 == 0 and W % 2 == 0 and getattr(node.conv, "bias", None) is None and node.residual is None
                and node.into is None)

    def refresh(self):
        """class weights <- current parameter values (two gathers of 8/7 the filter size), called by PackedWeights ahead of every
        re-pack: the pack set then lays them out with the rest"""
        w = self.conv.weight
        ext = torch.cat([w.data.reshape(-1), w.data.new_zeros(1)])
        for h, idx in zip(self.holders, self.idx):
            torch.index_select(ext, 0, idx, out=h.conv.weight.view(-1))

    def pack_rows(self, xin):
        """(N,D,H,W,4) clip with a zero 4th channel -> packed 3-channel rows as virtual pixels: (N,D,H,Wv+5,4) for the even outputs and
        the same memory from 2 virtual pixels in for the odd ones."""
        N, D, H, W, _ = xin.shape
        Wp = 3 * W // 4 + 5
        flat = torch.zeros(N * D * H * Wp * 4 + 8, dtype=xin.dtype, device=xin.device)
        xv = flat[:N * D * H * Wp * 4].view(N, D, H, Wp, 4)
        xv.view(N, D, H, Wp * 4)[..., 12:12 + 3 * W].view(N, D, H, W, 3).copy_(xin[..., :3])
        return xv, torch.as_strided(flat, (N, D, H, Wp, 4), xv.stride(), 8)


class BranchStreams:
    """Side branches of a block (S3D-G's inception branches 1-3, models/s3dg.py:80-99) on their own HIP streams — while the step is
    being CAPTURED into a HIP graph (rspnet_amd/graph_step.py).  The late blocks launch a few dozen workgroups per kernel; as
    graph nodes without a common stream order four of them run side by side (tools/stream_overlap_probe.py: x1.14-1.31 on one
    block).  Issued eagerly the same forks gain nothing (x1.0: the host feeds one kernel at a time), so outside a capture
    every node stays on the caller's stream.

    Forks are FLAT: only the stream the capture began on (`origin`) forks, and everything joins back into it.  ROCm 7.2's
    hipStreamEndCapture dies with a segmentation fault on a stream forked from a forked stream (tools/graph_nest_probe.py), so a
    pass that itself runs on a side stream (the query forward beside the key passes) keeps its branches in line.

    Ordering: a side stream first waits for the trunk (its inputs — and every earlier reader of memory its allocator pool may
    recycle — are complete), the trunk waits for all side streams when the next trunk node comes up; a tensor produced on a
    side stream is consumed there or, after that join, on the trunk."""
    _streams: Dict[Tuple, "torch.cuda.Stream"] = {}
    origin = None          # raw handle of the capturing stream, set by GraphedPretextStep around the capture
    SMALL_WGRAD_FLOPS = float(os.environ.get("RSP_WGRAD_ASIDE_GFLOP", "50")) * 1e9      # weight gradients below this size run beside the input gradient (side_task); swept 50 / 120 /
    #   300 / 1000 GFLOP: larger ones compete with the input gradient for the matrix pipe (R(2+1)D 77.9 -> 80.2 ms at 300)

    # ... and mid-sized ones whose operands are small enough not to fight the input gradient for HBM: since tiles differ in length
    #  a second matrix kernel fills the ragged ends of the launches — C3D 93.7 -> 92.9 ms with conv3a / conv4 included;
    # R(2+1)D's 266-GFLOP weight gradients stream 1.3 GB each and stay in line (77.9 -> 80.2 ms beside the input gradient)
    MID_WGRAD_FLOPS = float(os.environ.get("RSP_WGRAD_MID_GFLOP", "400")) * 1e9
    MID_WGRAD_BYTES = float(os.environ.get("RSP_WGRAD_MID_MB", "450")) * 1e6
    EAGER_TASKS = not os.environ.get("RSP_NO_EAGER_OVERLAP")
    # Issued eagerly, a weight gradient below AHEAD_MAX_FLOPS is handed to the task lane WITHOUT joining the previous one first (the
    # lane keeps its own order; what the tasks read stays alive until the join): the trunk no longer stalls where a weight gradient
    # outlasts the trunk's work up to the next one.  profiles/r06/experiments_r6.txt r6u-v: R3D-18 1 321-1 326 -> 1 344-1 346 clips/s,
    # R(2+1)D 446.3 -> 448.3-450.3, C3D 352.7-353.3 -> 353.6-354.7; without the cap C3D loses 0.5 % (its 177-355 GFLOP weight gradients
    # queue up beside the input gradients and fight them for the whole backward instead of filling their ragged ends).
    RUN_AHEAD = bool(int(os.environ.get("RSP_TASK_RUN_AHEAD", "1")))
    AHEAD_MAX_FLOPS = float(os.environ.get("RSP_TASK_AHEAD_MAX_GFLOP", "100")) * 1e9
    # a list while rspnet_amd/graph_step.py captures a piece of the backward as a LINEAR graph: side tasks are not run but collected
    # there as (fn, keepalive, FLOPs) — the stepper captures them as a graph of their own and replays it on the weight-gradient lane
    deferred = None

    def __init__(self, x: torch.Tensor):
        self.dev = x.device
        self.on = bool(x.is_cuda and _ops.backend().name == "hip" and BranchStreams.origin is not None
                       and torch.cuda.is_current_stream_capturing()
                       and torch.cuda.current_stream(self.dev).cuda_stream == BranchStreams.origin)
        self.origin_h = BranchStreams.origin
        # Inception branches as parallel graph branches: x1.14-1.31 on a block of its own and 51 -> 47 ms per S3D-G step when they were
        # the only concurrency in the graph; with the three forward passes and the weight gradients on streams of their own the
        # machine is already fed and the extra forks cost more in joins than they fill (41.6 -> 40.1 ms without them): off,
        # RSP_BRANCH_FORKS=1 brings them back
        self.fork_branches = self.on and bool(os.environ.get("RSP_BRANCH_FORKS"))
        # issued eagerly, the weight-gradient side task still pays (the host is far ahead of those kernels); the inception branches
        # do not (x1.00: dozens of tiny launches, the host feeds one at a time) and stay in line
        if (not self.on and BranchStreams.EAGER_TASKS and x.is_cuda and _ops.backend().name == "hip"
                and not torch.cuda.is_current_stream_capturing()):
            self.on = True
            self.origin_h = torch.cuda.current_stream(self.dev).cuda_stream
            # (measured: S3D-G issued eagerly 394 clips/s with the branches in line, 355-358 forked)
        self.active: Dict[int, "torch.cuda.Stream"] = {}
        self.task = None       # (task stream, tensors its kernels still read) of the outstanding side task

    def _get(self, key):
        key = (self.dev.index if self.dev.index is not None else torch.cuda.current_device(), key)
        st = BranchStreams._streams.get(key)
        if st is None:
            # the weight-gradient side task shares the process's "w" lane: one of three streams MEASURED to overlap with the
            # main stream and with one another (rspnet_amd/streams.py: HIP multiplexes streams onto a few hardware queues)
            if key[1] == "task":
                from . import streams as _streams
                st = _streams.lane(self.dev, "w")
            else:
                st = torch.cuda.Stream(device=self.dev)
            BranchStreams._streams[key] = st
        return st

    def side_task(self, fn, keepalive, cost: float = 0.0):
        """Run `fn` — a kernel sequence whose results nobody reads before the end of the pass: a SMALL weight gradient, which on
        its own leaves most of the machine idle — on a task stream beside the trunk (R3D-18 +1.2 %, R(2+1)D +1.1 %).  Only from
        the trunk (flat forks); one task outstanding, the previous one is joined first.  `keepalive`: the tensors it reads,
        held until the join so that the graph's memory pool does not hand their blocks out again underneath it."""
        if BranchStreams.deferred is not None:
            BranchStreams.deferred.append((fn, keepalive, cost))      # (cost: FLOPs — rspnet_amd/graph_step.py places its cuts by it)
            return
        cur = torch.cuda.current_stream(self.dev) if self.on else None
        if cur is None or cur.cuda_stream != self.origin_h:
            return fn()
        held = []
        if BranchStreams.RUN_AHEAD and cost < BranchStreams.AHEAD_MAX_FLOPS and not torch.cuda.is_current_stream_capturing():
            # issued eagerly the task stream is a lane of its own (rspnet_amd/streams.py) and keeps its own order: the trunk does not
            # wait for the previous task before it hands over the next one — a weight gradient longer than the trunk's work up to
            # the next one no longer stalls the trunk — and everything the tasks read stays alive until the join
            if self.task is not None:
                held = self.task[1]
        else:
            self.join_task()
        ts = self._get("task")
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            fn()
        held.append(keepalive)
        self.task = (ts, held)

    @contextlib.contextmanager
    def grads_ready(self):
        """Stream context ordered behind everything issued so far on the trunk AND on the side-task stream — for work that reads
        this pass's gradients but that the trunk need not wait for (an asynchronous all-reduce of a finished gradient bucket:
        RCCL orders it behind the stream it is issued from).  The trunk itself is not held up, unlike with join_task()."""
        cur = torch.cuda.current_stream(self.dev) if self.on else None
        if cur is None or cur.cuda_stream != self.origin_h or torch.cuda.is_current_stream_capturing():
            self.join_task()
            yield
            return
        ts = self._get("task")
        ts.wait_stream(cur)
        with torch.cuda.stream(ts):
            yield

    def join_task(self):
        if self.task is not None:
            torch.cuda.current_stream(self.dev).wait_stream(self.task[0])
            self.task = None

    def run(self, node, fn):
        br = getattr(node, "branch", 0) if (self.on and self.fork_branches) else 0
        if br == 0:
            self.join()
            return fn()
        st = self.active.get(br)
        if st is None:
            st = self._get(br)
            st.wait_stream(torch.cuda.current_stream(self.dev))
            self.active[br] = st
        with torch.cuda.stream(st):
            return fn()

    def join(self):
        if self.active:
            main = torch.cuda.current_stream(self.dev)
            for st in self.active.values():
                main.wait_stream(st)
            self.active = {}

    def finish(self):
        self.join()
        if self.on:
            self.join_task()


def _gate_fusion(plan: "Plan"):
    """{index of a ConvBN: (index of the Gate that is the only reader of its output, index of the Pool that is the only reader of
    the gate's output | None)} — S3D-G's sep_conv units (models/s3dg.py:36-72) and, for the two front-end ones, the max-pool
    behind them (:105-109).  The three run as one op (ops.bn_act_gate_fwd)."""
    cached = getattr(plan, "_gate_fusion", None)
    if cached is not None:
        return cached
    readers: Dict[int, List[int]] = {}
    for i, n in enumerate(plan.nodes):
        for m in (n.members if isinstance(n, ConvBNGroup) else [n]):
            for slot in (getattr(m, "src", None), getattr(m, "residual", None)):
                if slot is not None:
                    readers.setdefault(slot, []).append(i)
    table = {}
    if not os.environ.get("RSP_NO_GATE_FUSION"):
        for i, n in enumerate(plan.nodes[:-1]):
            g = plan.nodes[i + 1]
            if not (isinstance(n, ConvBN) and isinstance(g, Gate) and g.src == n.dst and g.branch == n.branch):
                continue
            if n.residual is not None or n.pool or n.into is not None or n.cout_pad or n.virtual_w or n.dst == plan.output_slot:
                continue
            if readers.get(n.dst) != [i + 1]:
                continue
            pi = None
            if i + 2 < len(plan.nodes):
                q = plan.nodes[i + 2]
                if (isinstance(q, Pool) and g.into is None and q.src == g.dst and q.branch == n.branch and readers.get(g.dst) == [i + 2]
                        and g.dst != plan.output_slot):
                    pi = i + 2
            table[i] = (i + 1, pi)
    plan._gate_fusion = table
    return table


GATE_POOL_APART = bool(os.environ.get("RSP_NO_GATE_POOL_KEEP"))      # (A/B switch: the kept forward of a gated unit runs its max-pool apart)


def _pool_fusion(plan: "Plan"):
    """{index of a ConvBN: index of the Pool that is the only reader of its output} — the ResNet stems (conv1 -> bn1 -> relu ->
    MaxPool3d(3, 2, 1), models/resnet.py:124-139,203-207).  A forward that keeps nothing for a backward (the two key passes) applies
    the BatchNorm and takes the window maximum in ONE pass over the convolution output (rsp_bn_act_pool_fwd takes any window): the
    activated tensor — 411 MB per pass on R3D-18 — is neither written nor read back.  Same arithmetic per element, same bits."""
    cached = getattr(plan, "_pool_fusion", None)
    if cached is not None:
        return cached
    readers: Dict[int, List[int]] = {}
    for i, n in enumerate(plan.nodes):
        for m in (n.members if isinstance(n, ConvBNGroup) else [n]):
            for slot in (getattr(m, "src", None), getattr(m, "residual", None)):
                if slot is not None:
                    readers.setdefault(slot, []).append(i)
    table = {}
    if not os.environ.get("RSP_NO_POOL_FUSION"):
        gated = _gate_fusion(plan)
        for i, n in enumerate(plan.nodes[:-1]):
            q = plan.nodes[i + 1]
            if not (isinstance(n, ConvBN) and isinstance(q, Pool) and q.src == n.dst and q.branch == n.branch) or i in gated:
                continue
            if n.pool or n.into is not None or n.cout_pad or n.dst == plan.output_slot or readers.get(n.dst) != [i + 1]:
                continue
            table[i] = i + 1
    plan._pool_fusion = table
    return table


def _slice_of(slots, into, lead_shape, device):
    """Channel-slice view of a concat tensor, allocating the tensor on first use."""
    slot, off, total = into
    if slot not in slots:
        slots[slot] = torch.empty(tuple(lead_shape) + (total,), dtype=torch.float32, device=device)
    return slots[slot]


def _view(t, into, C):
    return t if into is None else t[..., into[1]:into[1] + C]


def _eval_scale_shift(bn, bias) -> torch.Tensor:
    """Eval-mode BatchNorm folded into the (scale, shift) pair the fused BN kernel consumes: y = conv + b,
    out = (y - running_mean) / sqrt(running_var + eps) * gamma + beta  =  conv * scale + shift."""
    scale = bn.weight.data * torch.rsqrt(bn.running_var + float(bn.eps))
    shift = bn.bias.data - bn.running_mean * scale
    if bias is not None:
        shift = shift + bias.data * scale
    return torch.stack([scale, shift]).contiguous()


def run_forward(plan: Plan, x: torch.Tensor, packed: PackedWeights, keep: bool, training: bool = True,
                deferred: Optional[Dict[int, torch.Tensor]] = None) -> Tuple[torch.Tensor, Optional[ForwardCtx]]:
    """Execute `plan` on x (N,D,H,W,C).  keep=True records what backward needs.  training=True: batch-statistics BN (the
    pretext step never runs anything else: pretrain.py:225); training=False: running-statistics BN for the fine-tune /
    validation forward (finetune.py:333-345), no backward."""
    be = _ops.backend()
    assert training or not keep, "eval-mode forward keeps nothing for backward"

    def finalize(bn, stats, rows, bias_d):
        # deferred: {id(BatchNorm module): [2][C] buffer} — this pass reports its batch moments there and leaves the running
        # statistics to a later rsp_bn_running_update (ops.BnEmaSet); otherwise bn_finalize moves them itself
        bso = deferred.get(id(bn)) if deferred is not None else None
        if bso is not None:
            return be.bn_finalize(stats, rows, bias_d, bn.weight.data, bn.bias.data, float(bn.eps), float(bn.momentum), None, None,
                                  batch_stats_out=bso)
        return be.bn_finalize(stats, rows, bias_d, bn.weight.data, bn.bias.data, float(bn.eps), float(bn.momentum),
                              bn.running_mean, bn.running_var)

    slots: Dict[int, torch.Tensor] = {plan.input_slot: x}
    ctx = ForwardCtx(packed=packed) if keep else None
    pool_fusion = _pool_fusion(plan)
    skipped = set()

    def bn_apply(node, y, ss, cg_cout, N, do, ho, wo, xin, key=None):
        pk, ps = node.pool if node.pool else ((1, 1, 1), (1, 1, 1))
        pg = PoolGeom(N, do, ho, wo, cg_cout, pk, ps, (0, 0, 0))
        res = slots[node.residual] if node.residual is not None else None
        pi = pool_fusion.get(key) if key is not None else None
        if pi is not None:      # the max-pool behind this unit taken by the kernel that applies the BatchNorm (_pool_fusion)
            pnode = plan.nodes[pi]
            pgf = PoolGeom(N, do, ho, wo, cg_cout, pnode.k, pnode.s, pnode.p)
            if not keep:
                slots[pnode.dst] = be.bn_act_pool_fwd(pgf, y, ss, res, node.relu)
                skipped.add(pi)
                return pgf, res
            # kept for a backward: the same pass also writes the arg-max (the Pool node's saved state); the BatchNorm backward
            # recomputes the activation from y as everywhere, so the activated tensor is never materialised
            fused = be.bn_act_maxpool_fwd(pgf, y, ss, node.relu, True) if (res is None and hasattr(be, "bn_act_maxpool_fwd")) else None
            if fused is not None:
                slots[pnode.dst], idx = fused
                ctx.saved[pi] = (pgf, idx)
                skipped.add(pi)
                return pg, res
        if node.into is not None:
            pdo, pho, pwo = pg.out_dims
            out = _view(_slice_of(slots, node.into, (N, pdo, pho, pwo), xin.device), node.into, cg_cout)
            be.bn_act_pool_fwd(pg, y, ss, res, node.relu, out=out)
        else:
            slots[node.dst] = be.bn_act_pool_fwd(pg, y, ss, res, node.relu)
        return pg, res

    def convbn_virtual(node, key):
        """ConvBN of a 3-channel stride-2 stem as two ordinary convolutions over virtual pixels (see VirtualStem)."""
        xin = slots[node.src]
        N, D, H, W, _ = xin.shape
        w = node.conv.weight
        Cout = w.shape[0]
        Cp = node.cout_pad if node.cout_pad > Cout else Cout
        vs = packed._virtual.get(id(node.conv))
        if vs is None or vs.device != w.device:
            vs = packed._virtual[id(node.conv)] = VirtualStem(node)
        x_e, x_o = vs.pack_rows(xin)
        (kT, kH, _), (sT, sH, _), (pT, pH, _) = node.k, node.s, node.p
        cg = ConvGeom(N, D, H, x_e.shape[3], 4, Cp, (kT, kH, 6), (sT, sH, 3), (pT, pH, 0), Cin_alg=3 * 7 / 6)
        do, ho, g = cg.out_dims
        y = torch.empty((N, do, ho, 2 * g, Cp), dtype=torch.float32, device=xin.device)
        yv = y.view(N, do, ho, g, 2 * Cp)
        parts = []
        for c, xc in enumerate((x_e, x_o)):
            _, st = be.conv_fwd(cg, xc, packed.get(vs.holders[c], cg), None, True, out=yv[..., c * Cp:(c + 1) * Cp], out_ld=2 * Cp)
            parts.append(st)
        rows = N * do * ho * 2 * g
        mi, ss = finalize(node.bn, torch.cat(parts), rows, None)
        pg, res, = bn_apply(node, y, ss, Cp, N, do, ho, 2 * g, xin, key)
        if keep:
            ctx.saved[key] = ("vstem", x_e, x_o, y, mi, ss, cg, pg, vs)

    def convbn(node, key, gated=None):
        xin = slots[node.src]
        if training and node.virtual_w and VirtualStem.applies(node, xin):
            return convbn_virtual(node, key)
        N, D, H, W, Cin = xin.shape
        w = node.conv.weight
        Cout = w.shape[0]
        Cp = node.cout_pad if node.cout_pad > Cout else Cout
        cg = ConvGeom(N, D, H, W, Cin, Cp, node.k, node.s, node.p, Cin_alg=w.shape[1])
        bias = getattr(node.conv, "bias", None)
        bn = node.bn
        bias_d = None if bias is None else bias.data
        if training:
            # (no shipped backbone has a conv bias on channel-padded units; if one does, the conv needs it at the padded length)
            bias_conv = bias_d if (bias_d is None or Cp == Cout) else _pad_vec(bias_d, Cp)
            y, stats = be.conv_fwd(cg, xin, packed.get(node, cg), bias_conv, True)
            # (zero-padded output channels, Cp > Cout: the BatchNorm vectors keep their Cout entries, the kernels treat the rest
            #  as gamma = beta = 0 and leave the running statistics of the real channels alone)
            mi, ss = finalize(bn, stats, cg.rows, bias_d)
        else:
            y, _ = be.conv_fwd(cg, xin, packed.get(node, cg), None, False)     # bias folded into the shift
            ss = _eval_scale_shift(bn, bias)
            mi = None
            if Cp != Cout:
                ss = torch.cat([ss, torch.zeros((2, Cp - Cout), dtype=ss.dtype, device=ss.device)], dim=1).contiguous()
        do, ho, wo = cg.out_dims
        if gated is not None:
            # BatchNorm-apply + self-gating (+ the max-pool behind a front-end unit, when nothing is kept for a backward) in two
            # passes over y: see _gate_fusion
            gi, pi = gated
            gnode = plan.nodes[gi]
            pnode = plan.nodes[pi] if pi is not None else None
            pg = PoolGeom(N, do, ho, wo, cg.Cout)
            pool = PoolGeom(N, do, ho, wo, cg.Cout, pnode.k, pnode.s, pnode.p) if pnode is not None else None
            # (a backward recomputes the activation from y: nothing but the (sample, channel) means and gates is kept)
            keep_act = keep and not GATE_BWD_FUSED
            # a forward a backward follows pools in the same pass only if that pass can also write the pool's arg-max
            with_idx = (keep and pool is not None and not keep_act and getattr(be, "gate_pool_keep", False) and not GATE_POOL_APART
                        and be.bn_act_gate_pool_idx_ok(pool, y, ss))
            if keep and not with_idx:
                pnode = pool = None
            out = None
            if gnode.into is not None:
                out = _view(_slice_of(slots, gnode.into, (N, do, ho, wo), xin.device), gnode.into, cg.Cout)
            if with_idx:
                o, a, mean, gate, idx = be.bn_act_gate_fwd(pg, y, ss, node.relu, gnode.conv.weight.data, gnode.conv.bias.data, False,
                                                           pool=pool, out=out, pool_idx=True)
                ctx.saved[pi] = (pool, idx)
            else:
                o, a, mean, gate = be.bn_act_gate_fwd(pg, y, ss, node.relu, gnode.conv.weight.data, gnode.conv.bias.data, keep_act,
                                                      pool=pool, out=out)
            if pnode is not None:
                slots[pnode.dst] = o
                skipped.add(pi)
            elif gnode.into is None:
                slots[gnode.dst] = o
            skipped.add(gi)
            if keep:
                ctx.saved[key] = _Saved(xin, y, mi, ss, cg, pg, None)
                ctx.saved[gi] = (a, mean, gate) if keep_act else ("fused", key, mean, gate)
            return
        pg, res = bn_apply(node, y, ss, cg.Cout, N, do, ho, wo, xin, key)
        if keep:
            ctx.saved[key] = _Saved(xin, y, mi, ss, cg, pg, res)

    def convbn_group(node, ni):
        """One GEMM over the members' concatenated filters (see ConvBNGroup); falls back to member-by-member execution."""
        ms = node.members
        wcat = _adjacent_cat([m.conv.weight.data for m in ms]) if training else None
        if wcat is None:
            for j, m in enumerate(ms):
                convbn(m, (ni, j))
            return
        xin = slots[ms[0].src]
        N, D, H, W, Cin = xin.shape
        if getattr(node, "_cat", None) is None:
            node._cat = _CatConv()
        node._cat.weight = wcat
        cg = ConvGeom(N, D, H, W, Cin, wcat.shape[0], ms[0].k, ms[0].s, ms[0].p)
        y, stats = be.conv_fwd(cg, xin, packed.get(node._cat_node(), cg), None, True)
        do, ho, wo = cg.out_dims
        off, per = 0, []
        for m in ms:
            C = m.conv.weight.shape[0]
            bn = m.bn
            mi, ss = finalize(bn, stats[:, off:off + C], cg.rows, None)
            pg, _ = bn_apply(m, y[..., off:off + C], ss, C, N, do, ho, wo, xin)
            per.append((off, C, mi, ss, pg))
            off += C
        if keep:
            ctx.saved[ni] = ("group", xin, y, cg, per)

    fusion = _gate_fusion(plan) if training else {}

    def run_node(ni, node):
        if ni in skipped:
            return
        if isinstance(node, ConvBN):
            convbn(node, ni, fusion.get(ni))
        elif isinstance(node, ConvBNGroup):
            convbn_group(node, ni)
        elif isinstance(node, ConvBias):
            xin = slots[node.src]
            N, D, H, W, Cin = xin.shape
            w = node.conv.weight
            cg = ConvGeom(N, D, H, W, Cin, w.shape[0], node.k, node.s, node.p)
            y, _ = be.conv_fwd(cg, xin, packed.get(node, cg), node.conv.bias.data, False)
            if node.relu:
                y = be.eltwise("relu_fwd", y, out=y)
            slots[node.dst] = y
            if keep:
                ctx.saved[ni] = (xin, y, cg)
        elif isinstance(node, Pool):
            xin = slots[node.src]
            N, D, H, W, Cc = xin.shape
            pg = PoolGeom(N, D, H, W, Cc, node.k, node.s, node.p)
            slots[node.dst], idx = be.maxpool_fwd(pg, xin, keep)
            if keep:
                ctx.saved[ni] = (pg, idx)
        elif isinstance(node, Gate):
            xin = slots[node.src]
            out = (_view(_slice_of(slots, node.into, tuple(xin.shape[:4]), xin.device), node.into, xin.shape[4])
                   if node.into is not None else None)
            o, mean, gate = be.gate_fwd(xin, node.conv.weight.data, node.conv.bias.data, out=out)
            if node.into is None:
                slots[node.dst] = o
            if keep:
                ctx.saved[ni] = (xin, mean, gate)
        else:
            raise NotImplementedError(f"plan node {type(node).__name__}")

    branches = BranchStreams(x)
    for ni, node in enumerate(plan.nodes):
        branches.run(node, lambda: run_node(ni, node))
    branches.finish()
    out = slots[plan.output_slot]
    if keep:
        ctx.feat_shape = tuple(out.shape)
    return out, ctx


def run_backward(plan: Plan, ctx: ForwardCtx, dfeat: torch.Tensor, grad_of, after_param_grads=None,
                 want_input_grad: bool = False, packed: Optional[PackedWeights] = None):
    """Backward through the plan in one go: see run_backward_iter."""
    it = run_backward_iter(plan, ctx, dfeat, grad_of, after_param_grads, want_input_grad, packed)
    try:
        while True:
            next(it)
    except StopIteration as done:
        return done.value


def run_backward_iter(plan: Plan, ctx: ForwardCtx, dfeat: torch.Tensor, grad_of, after_param_grads=None,
                      want_input_grad: bool = False, packed: Optional[PackedWeights] = None):
    """Generator form of the backward: yields the index of each plan node once its kernels have been issued, so that a caller can
    cut the backward into pieces (rspnet_amd/graph_step.py captures each piece as a linear HIP graph and runs the weight gradients
    that `BranchStreams.deferred` collected beside the next piece); the generator's return value is run_backward's.

    Backward through the plan.  `grad_of(param)` returns the (pre-allocated, flat-buffer) gradient view to
    fill for a parameter, or None to skip it.  `after_param_grads(node_index, grads_ready)` is called once a node's parameter
    gradients have been ISSUED (used to launch bucketed all-reduces overlapped with the rest of backward); a weight gradient may
    still be running on the side-task stream then — the hook issues whatever reads gradients inside `with grads_ready():`, a
    stream context ordered behind all of them (BranchStreams.grads_ready).
    want_input_grad: also propagate to the plan's input slot and return that gradient (projection-head sub-plans, whose
    input is the backbone feature; the backbone's own input is the clip and needs none)."""
    be = _ops.backend()
    packed = packed if packed is not None else ctx.packed
    dslots: Dict[int, torch.Tensor] = {plan.output_slot: dfeat}
    branches = BranchStreams(dfeat)
    fused_gates: Dict[int, Tuple] = {}      # key of a ConvBN -> (index, means, gates) of the gate fused behind it

    def add_grad(slot, g):
        # gradient accumulation at fan-out points; g is always the fresh output of the op that produced it, so the sum is
        # written over it (no allocation, nothing else aliases it)
        if g is None:
            return
        if slot in dslots:
            dslots[slot] = be.eltwise("add", dslots[slot].contiguous(), g, out=g)
        else:
            dslots[slot] = g

    def convbn_virtual_bwd(node, sv, ni):
        _, x_e, x_o, y, mi, ss, cg, pg, vs = sv
        assert node.src == plan.input_slot and not want_input_grad, "a virtual-pixel stem reads the clip: no input gradient"
        dout = dslots.pop(node.dst)
        bn = node.bn
        dy, _ = be.bn_act_pool_bwd(pg, y, None, dout, bn.weight.data, mi, ss, node.relu, False, grad_of(bn.weight), grad_of(bn.bias))
        gw = grad_of(node.conv.weight)
        if gw is not None:
            N, do, ho, Wo, Cp = dy.shape
            dyv = dy.view(N, do, ho, Wo // 2, 2 * Cp)
            gflat = gw.view(-1)
            gflat.zero_()
            for c, xc in enumerate((x_e, x_o)):
                gv = torch.empty_like(vs.holders[c].conv.weight)
                be.conv_wgrad(cg, xc, dyv[..., c * Cp:(c + 1) * Cp], gv)
                # every real filter element appears once in a class's virtual filter: unique indices, order-independent
                gflat.index_add_(0, vs.dst[c], gv.view(-1).index_select(0, vs.src[c]))
        if after_param_grads is not None:
            after_param_grads(ni, branches.grads_ready)

    def convbn_bwd(node, key, ni):
        sv = ctx.saved.pop(key)
        if isinstance(sv, tuple) and sv[0] == "vstem":
            return convbn_virtual_bwd(node, sv, ni)
        bn = node.bn
        gated = fused_gates.pop(key, None)
        if gated is not None:
            # the self-gating unit behind this BatchNorm, whose forward kept no activation: both backwards as one op
            gi, mean, gate = gated
            gnode = plan.nodes[gi]
            dout = _view(dslots[gnode.into[0]], gnode.into, sv.cg.Cout) if gnode.into is not None else dslots.pop(gnode.dst)
            dy = be.bn_act_gate_bwd(sv.pg, sv.y, dout, bn.weight.data, sv.mi, sv.ss, node.relu, gnode.conv.weight.data, mean, gate,
                                    grad_of(bn.weight), grad_of(bn.bias), grad_of(gnode.conv.weight), grad_of(gnode.conv.bias))
            dres = None
            if after_param_grads is not None:
                after_param_grads(gi, branches.grads_ready)
        else:
            dout = _view(dslots[node.into[0]], node.into, sv.cg.Cout) if node.into is not None else dslots.pop(node.dst)
            dy, dres = be.bn_act_pool_bwd(sv.pg, sv.y, sv.res, dout, bn.weight.data, sv.mi, sv.ss, node.relu,
                                          node.residual is not None, grad_of(bn.weight), grad_of(bn.bias))
        if node.residual is not None:
            add_grad(node.residual, dres)
        bias = getattr(node.conv, "bias", None)
        if bias is not None:
            gb = grad_of(bias)
            if gb is not None:
                # A conv bias in front of train-mode BatchNorm has an identically-zero gradient (BN subtracts the
                # batch mean); the reference's autograd produces ~1e-8 rounding noise there.
                gb.zero_()
        # (a channel-padded geometry writes only the parameter's own channels: the reduce drops the padding's gradients)
        gw = grad_of(node.conv.weight)
        # (side stream: a gradient hook — the bucketed all-reduce of the data-parallel path — joins the task before it lets a
        #  bucket go, see run_backward's docstring)
        if (sv.cg.flops < BranchStreams.SMALL_WGRAD_FLOPS or
                (sv.cg.flops < BranchStreams.MID_WGRAD_FLOPS and sv.cg.bytes < BranchStreams.MID_WGRAD_BYTES)):
            branches.side_task(lambda: be.conv_wgrad(sv.cg, sv.x, dy, gw), (sv.x, dy), cost=float(sv.cg.flops))
        else:
            be.conv_wgrad(sv.cg, sv.x, dy, gw)
        if after_param_grads is not None:
            after_param_grads(ni, branches.grads_ready)
        if node.src != plan.input_slot or want_input_grad:
            add_grad(node.src, be.conv_dgrad_packed(sv.cg, dy, packed.get_dgrad(node, sv.cg)))

    def convbn_group_bwd(node, ni):
        sv = ctx.saved.pop(ni, None)
        ms = node.members
        if sv is None:                                   # forward ran the members one by one
            for j in range(len(ms) - 1, -1, -1):
                convbn_bwd(ms[j], (ni, j), ni)
            return
        _, xin, y, cg, per = sv
        dy_cat = torch.empty_like(y)
        for m, (off, C, mi, ss, pg) in zip(ms, per):
            dout = _view(dslots[m.into[0]], m.into, C) if m.into is not None else dslots.pop(m.dst)
            be.bn_act_pool_bwd(pg, y[..., off:off + C], None, dout, m.bn.weight.data, mi, ss, m.relu, False,
                               grad_of(m.bn.weight), grad_of(m.bn.bias), dy_out=dy_cat[..., off:off + C])
        gws = [grad_of(m.conv.weight) for m in ms]
        gcat = _adjacent_cat(gws) if all(g is not None for g in gws) else None
        if gcat is not None:
            be.conv_wgrad(cg, xin, dy_cat, gcat)
        else:
            tmp = torch.empty_like(node._cat.weight)
            be.conv_wgrad(cg, xin, dy_cat, tmp)
            off = 0
            for m, g in zip(ms, gws):
                if g is not None:
                    g.copy_(tmp[off:off + g.shape[0]])
                off += m.conv.weight.shape[0]
        if after_param_grads is not None:
            after_param_grads(ni, branches.grads_ready)
        if ms[0].src != plan.input_slot or want_input_grad:
            add_grad(ms[0].src, be.conv_dgrad_packed(cg, dy_cat, packed.get_dgrad(node._cat_node(), cg)))

    def run_node(ni, node):
        if isinstance(node, ConvBN):
            convbn_bwd(node, ni, ni)
        elif isinstance(node, ConvBNGroup):
            convbn_group_bwd(node, ni)
        elif isinstance(node, ConvBias):
            xin, y, cg = ctx.saved.pop(ni)
            dout = dslots.pop(node.dst)
            dz = be.eltwise("relu_bwd", y, dout.contiguous()) if node.relu else dout.contiguous()
            be.conv_wgrad(cg, xin, dz, grad_of(node.conv.weight), grad_of(node.conv.bias))
            if after_param_grads is not None:
                after_param_grads(ni, branches.grads_ready)
            if node.src != plan.input_slot or want_input_grad:
                add_grad(node.src, be.conv_dgrad_packed(cg, dz, packed.get_dgrad(node, cg)))
            del dz, dout
        elif isinstance(node, Pool):
            pg, idx = ctx.saved.pop(ni)
            add_grad(node.src, be.maxpool_bwd(pg, dslots.pop(node.dst), idx))
        elif isinstance(node, Gate):
            sv = ctx.saved.pop(ni)
            if sv[0] == "fused":      # its backward runs with the BatchNorm's in front of it (convbn_bwd)
                fused_gates[sv[1]] = (ni, sv[2], sv[3])
                return
            xin, mean, gate = sv
            dout = (_view(dslots[node.into[0]], node.into, xin.shape[4]) if node.into is not None
                    else dslots.pop(node.dst))
            add_grad(node.src, be.gate_bwd(xin, dout, node.conv.weight.data, mean, gate, grad_of(node.conv.weight),
                                           grad_of(node.conv.bias)))
            if after_param_grads is not None:
                after_param_grads(ni, branches.grads_ready)
        else:
            raise NotImplementedError(f"plan node {type(node).__name__}")

    # (side branches of a block — engine.BranchStreams — each own their slots; the fan-out sum at the block's input happens on the
    #  trunk after the join: the block's last backward node, the grouped pointwise convolution, is a trunk node)
    for ni in range(len(plan.nodes) - 1, -1, -1):
        node = plan.nodes[ni]
        branches.run(node, lambda: run_node(ni, node))
        yield ni
    branches.finish()
    return dslots.get(plan.input_slot) if want_input_grad else None
