"""MoCo-style two-branch pretext model with relative-speed + appearance heads (RSPNet), MI355X-native.

Same module surface as /root/reference/moco/builder_diffspeed_diffloss.py:263-547 (`Loss`, `MoCoDiffLossTwoFc`):
constructor arguments, forward signature / return structure, state-dict keys, side effects (momentum update before
the key passes, two train-mode key passes, queue enqueue of k_neg_A).  What differs is the machinery:

* one autograd node wraps the whole query encoder; its backward walks the layer plan with HIP kernels and writes
  straight into a flat gradient buffer (rspnet_amd/flat.py), launching bucketed RCCL all-reduces as it goes;
* shuffle-BN is a clip all-to-all over xGMI (each clip has exactly one destination rank) instead of the reference's
  all-gather + select (:361-387); the permutation travels host-side, so no device sync is needed;
* the feature un-shuffle (:389-406) and the queue's key all-gather (:348) share ONE small all-gather per key pass;
* queue_ptr is mirrored on the host (the reference reads it back with int(), :352, every step).
"""
from __future__ import annotations

import contextlib
import os
import random
from typing import List, Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist
from torch import Tensor, nn

from .. import ops as _ops
from .. import streams as _streams
from ..engine import INPUT_CHANNEL_PAD
from ..flat import FlatEncoderPair

BUCKET_FLOATS = 8 << 20  # 32 MiB gradient buckets: few large xGMI collectives


def _world():
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


class _CommTimer:
    """Books how long the compute stream is held up by a collective: HIP events on the current stream around the blocking
    call / the wait (the collective itself runs on RCCL's stream), host wall time for CPU process groups."""

    def __init__(self, log, name, device):
        self.log, self.name, self.cuda = log, name, device.type == "cuda"

    def __enter__(self):
        if self.log is None:
            return self
        if self.cuda:
            self.e0 = torch.cuda.Event(enable_timing=True)
            self.e0.record()
        else:
            import time
            self.t0 = time.perf_counter()
        return self

    def __exit__(self, *exc):
        if self.log is None:
            return False
        if self.cuda:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            self.log.setdefault(self.name, []).append((self.e0, e1))
        else:
            import time
            self.log.setdefault(self.name, []).append((time.perf_counter() - self.t0) * 1e3)
        return False


class _LossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, l1, l2, lp, ln, margin, A, M):
        losses, d1, d2, dp, dn = _ops.backend().loss_fwd_bwd(l1.contiguous(), l2.contiguous(), lp.contiguous(),
                                                             ln.contiguous(), margin, A, M)
        ctx.save_for_backward(d1, d2, dp, dn)
        return losses[0], losses[1], losses[2]

    @staticmethod
    def backward(ctx, g, gA, gM):
        d1, d2, dp, dn = ctx.saved_tensors
        # loss_A / loss_M are reporting outputs (pretrain.py:160-195 only logs them); gradients flow through `loss`.
        return d1 * g, d2 * g, dp * g, dn * g, None, None, None


class Loss(nn.Module):
    """A*(CE(logits1,0)+CE(logits2,0)) + M*MarginRanking(l_pos_M, l_neg_M; margin)  (reference :263-283).

    The reference passes (B,1),(B,1) inputs with a (B,) target to nn.MarginRankingLoss; under its pinned torch 1.6 that
    evaluates to mean_i max(0, margin - (l_pos_M[i] - l_neg_M[i])) — that value and gradient are
    what the kernel computes.  `target` must be all zeros and `ranking_target` all ones, as the model produces them.
    """

    def __init__(self, margin=1.0, A: float = 1.0, M: float = 1.0):
        super().__init__()
        self.margin, self.A, self.M = float(margin), float(A), float(M)

    def forward(self, output: Tuple[Tensor, Tensor], target: Tensor, ranking_logits: Tuple[Tensor, Tensor],
                ranking_target: Tensor):
        loss, ce, ranking = _LossFn.apply(output[0], output[1], ranking_logits[0], ranking_logits[1], self.margin,
                                          self.A, self.M)
        return loss, ce.detach(), ranking.detach()


class _PretextFn(torch.autograd.Function):
    """Query-encoder forward + logits as one autograd node (parameters are listed as inputs only so that autograd
    routes the backward call here; their gradients are written directly into the flat gradient buffer)."""

    @staticmethod
    def forward(ctx, model: "MoCoDiffLossTwoFc", x_q: Tensor, keys, *params):
        be = _ops.backend()
        with torch.no_grad():
            pre, model._q_pre = model._q_pre, None
            q_A, q_M, ectx = pre if pre is not None else model.encoder_q.forward_ndhwc(x_q, keep=True)
            k_A, k_M, kneg_A, kneg_M = keys
            queue = model.queue
            l1, l2, lp, ln = be.logits_fwd(q_A, q_M, k_A, k_M, kneg_A, kneg_M, queue, 1.0 / model.T)
        ctx.model, ctx.ectx = model, ectx
        # the queue is overwritten by the enqueue right after forward: backward needs this step's copy (the reference
        # clones it too, :525)
        ctx.keys = (k_A, k_M, kneg_A, kneg_M, queue.clone())
        model._last_q = (q_A, q_M)
        return l1, l2, lp, ln

    @staticmethod
    def backward(ctx, dl1, dl2, dlp, dln):
        model = ctx.model
        be = _ops.backend()
        k_A, k_M, kneg_A, kneg_M, queue = ctx.keys
        dev = queue.device

        def z(g, like_shape):
            return torch.zeros(like_shape, dtype=torch.float32, device=dev) if g is None else g.contiguous()

        B, K = k_A.shape[0], queue.shape[1]
        dqA, dqM = be.logits_bwd(z(dl1, (B, K + 1)), z(dl2, (B, K + 1)), z(dlp, (B, 1)), z(dln, (B, 1)), k_A, k_M,
                                 kneg_A, kneg_M, queue, 1.0 / model.T)
        if model._defer_backward:
            # rspnet_amd/graph_step.py runs the encoder's backward itself, in pieces (`_backward_iter`)
            model._pending_bwd = (ctx.ectx, dqA, dqM)
        else:
            model._backward_encoder_q(ctx.ectx, dqA, dqM)
        ctx.ectx = None
        return (None, None, None) + (None,) * len(model._q_params)


class MoCoDiffLossTwoFc(nn.Module):
    def __init__(self, base_encoder, dim=128, K=65536, m=0.999, T=0.07, mlp=False,
                 diff_speed: Optional[List[int]] = None, *, force_collectives: Optional[bool] = None):
        """Reference signature (:286-296) + one keyword: `force_collectives` (default: the RSP_FORCE_COLLECTIVES environment
        variable) makes every collective of the data-parallel step run even in a world of ONE rank — the clip all-to-all with
        its split lists, the fused key all-gather, the bucketed gradient all-reduce launched from inside backward, the host-side
        broadcast of the step's random draws.  With one rank they move nothing, so the results are those of the plain step; the
        switch exists so that the RCCL path executes on a 1-GPU machine (tests/test_rccl_gpu.py, bench.py --force-dp)."""
        super().__init__()
        self.K, self.m, self.T = K, m, T
        self.diff_speed = diff_speed
        self.force_collectives = bool(os.environ.get("RSP_FORCE_COLLECTIVES")) if force_collectives is None else bool(force_collectives)
        self.encoder_q = base_encoder(num_classes=dim)
        self.encoder_k = base_encoder(num_classes=dim)
        if mlp:
            raise NotImplementedError("mlp=True: never set by ModelFactory (moco/__init__.py:39-46)")
        for param_q, param_k in zip(self.encoder_q.parameters(), self.encoder_k.parameters()):
            param_k.data.copy_(param_q.data)
            param_k.requires_grad = False
        self.register_buffer("queue", nn.functional.normalize(torch.randn(dim, K), dim=0))
        self.register_buffer("queue_ptr", torch.zeros(1, dtype=torch.long))
        self.alpha = 0.5
        assert self.diff_speed is not None, "This branch is for diff speed"

        self._flat: Optional[FlatEncoderPair] = None
        self._q_params: List[nn.Parameter] = []
        self._ptr_host: Optional[int] = None
        self._ptr_on_device = False           # True: queue_ptr lives on the device only (graph-captured steps)
        self._ptr_checked = False             # the device-resident pointer has been validated against the enqueue size
        self.overlap_query = not os.environ.get("RSP_NO_QOVERLAP")     # (switches for A/B runs of tools/)
        self.overlap_keys = not os.environ.get("RSP_NO_KOVERLAP")
        # issued eagerly (more than one rank; --no-graph) the same fork pays when the host runs far ahead of the GPU — C3D's 250 long
        # launches — and the unequal tiles of DESIGN 5d leave slots for it to fill
        self.overlap_query_eager = not os.environ.get("RSP_NO_EAGER_OVERLAP")
        self._query_stream = self._key_stream = None
        self._ema_k = self._ema_map = None
        self._q_pre = None
        self._cpu_group = None
        self._last_q = None
        self._last_k = []
        self._last_speed = None
        self._last_draw = None
        self.comm_log = None                  # set to {} to collect per-collective stall times (see _comm)
        self._pending = []
        self._q_version = None
        self._defer_reduce = False            # True: backward leaves the gradient all-reduce to `_reduce_gradients` (segmented replay)
        self._defer_backward = False          # True: loss.backward() stops at the query features; `_backward_iter` runs the encoder's part
        self._pending_bwd = None
        # DistributedDataParallel(broadcast_buffers=True), the reference's default (moco/__init__.py:49-53): rank 0's BatchNorm
        # running statistics reach every rank before each forward (`_broadcast_running_stats`).  DataParallelPretext sets it.
        self.broadcast_buffers = True
        self._bn_flat = None
        self._lanes_chosen = False
        self.register_load_state_dict_post_hook(lambda mod, keys: mod._state_loaded())

    def _dp(self):
        """(rank, world size, collectives on?) — the data-parallel exchanges run with more than one rank, or with any
        initialised process group when `force_collectives` is set."""
        rank, ws = _world()
        return rank, ws, ws > 1 or (self.force_collectives and dist.is_available() and dist.is_initialized())

    # ---- state ----------------------------------------------------------------------------------------------------
    def _state_loaded(self):
        self._ptr_host = None
        self._ptr_checked = False
        self._q_version = None
        self.encoder_q.weights_changed()
        self.encoder_k.weights_changed()

    def _check_q_weights(self):
        """Re-pack encoder_q's forward weights when ANY optimizer touched them since the last step.  The fused SGD kernel
        writes through raw pointers and invalidates the cache itself (optim.py); every torch-side update (torch.optim.SGD,
        nesterov / dampening fallbacks, manual p.add_) bumps the parameters' autograd version counters instead."""
        ver = sum(p._version for p in self._q_params)
        if ver != self._q_version:
            self._q_version = ver
            self.encoder_q.weights_changed()

    def _prepare(self):
        if self._flat is None:
            self._flat = FlatEncoderPair(self.encoder_q, self.encoder_k, self.encoder_q.untrained_prefixes(),
                                         adjacent=self.encoder_q.adjacent_parameters())
            self._q_params = list(self.encoder_q.parameters())
        self._flat.ensure()
        self._tie_num_batches_tracked()
        self._tie_running_stats()
        dev = self.queue.device
        if dev.type == "cuda" and not self._lanes_chosen and not torch.cuda.is_current_stream_capturing():
            # the process's side lanes are MEASURED (rspnet_amd/streams.py: spin kernels and device synchronisations) — here, on a quiet
            # GPU in front of the first step, not in the middle of it between its first collectives
            _streams.lane(dev, "q")
            self._lanes_chosen = True
        self._check_q_weights()

    def _bns(self, enc):
        """The BatchNorm modules of an encoder, in module order (the module tree is fixed after construction: walked once, not
        per step — 8 000 `named_modules` frames per S3D-G step otherwise)."""
        cache = self.__dict__.setdefault("_bn_lists", {})
        got = cache.get(id(enc))
        if got is None:
            got = cache[id(enc)] = [mod for mod in enc.modules() if isinstance(mod, nn.modules.batchnorm._BatchNorm)]
        return got

    def _tie_num_batches_tracked(self):
        """All BN step counters of one encoder share one int64 buffer so a key/query pass bumps them with one add."""
        for enc, attr in ((self.encoder_q, "_nbt_q"), (self.encoder_k, "_nbt_k")):
            bns = self._bns(enc)
            flat = getattr(self, attr, None)
            dev = self.queue.device
            ok = flat is not None and flat.device == dev and all(
                b.num_batches_tracked.data_ptr() == flat.data_ptr() + 8 * i for i, b in enumerate(bns))
            if ok:
                continue
            flat = torch.stack([b.num_batches_tracked.to(dev) for b in bns]) if bns else torch.zeros(0, dtype=torch.long,
                                                                                                      device=dev)
            for i, b in enumerate(bns):
                b._buffers["num_batches_tracked"] = flat[i]
            object.__setattr__(self, attr, flat)

    def _tie_running_stats(self):
        """All BatchNorm running means / variances of both encoders are views into ONE fp32 buffer (each on a 16-byte boundary): DDP's
        per-forward buffer broadcast is then one small collective per step (`_broadcast_running_stats`) instead of one per tensor."""
        bns = self._bns(self.encoder_q) + self._bns(self.encoder_k)
        dev = self.queue.device
        flat, off, ok = self._bn_flat, 0, self._bn_flat is not None and self._bn_flat.device == dev
        offs = []
        for b in bns:
            for t in (b.running_mean, b.running_var):
                offs.append(off)
                ok = ok and t.data_ptr() == flat.data_ptr() + 4 * off
                off += (t.numel() + 3) // 4 * 4
        if ok:
            return
        flat = torch.zeros(off, dtype=torch.float32, device=dev)
        it = iter(offs)
        for b in bns:
            for name in ("running_mean", "running_var"):
                t, o = b._buffers[name], next(it)
                v = flat[o:o + t.numel()]
                v.copy_(t.to(dev))
                b._buffers[name] = v
        self._bn_flat = flat

    @torch.no_grad()
    def _broadcast_running_stats(self):
        """What DistributedDataParallel does before every forward with the reference's default broadcast_buffers=True
        (moco/__init__.py:49-53, torch's `_sync_buffers`): rank 0's buffers overwrite every rank's.  queue / queue_ptr /
        num_batches_tracked are equal on all ranks by construction (every rank enqueues the same all-gathered keys and counts the same
        passes); the BatchNorm running statistics are not — each rank's moving averages see its own half of every batch — so they
        travel: one broadcast of the flat buffer in front of the step (every pass of the step, forked ones included, starts from
        this stream behind it: none reads or moves a running statistic before rank 0's have arrived)."""
        if not (self.broadcast_buffers and self._dp()[2]):
            return
        if self._bn_flat is None:
            self._prepare()
        if self._bn_flat.numel() == 0:
            return
        h = dist.broadcast(self._bn_flat, src=0, async_op=True)
        with self._comm("broadcast_buffers"):
            h.wait()

    # ---- reference-named pieces -------------------------------------------------------------------------------------
    @torch.no_grad()
    def _momentum_update_key_encoder(self):
        """:337-343 on the flat buffers (params only; BN buffers of encoder_k evolve by its own forwards)."""
        _ops.backend().momentum_update(self._flat.k_flat, self._flat.q_flat, self.m)
        self.encoder_k.weights_changed()

    def _cpu_pg(self):
        """Host-side side channel for the per-step (speed, permutations) message (single node: loopback).  The gloo group is
        created collectively by DataParallelPretext.__init__ (setup_side_group) and agreed on by all ranks; None means the
        message travels through the default (device) group instead, at the price of one small device sync per step."""
        if self._cpu_group is None:
            if dist.get_backend() == "gloo":
                self._cpu_group = dist.group.WORLD
            else:
                self._cpu_group = False          # no side group was set up (bare model under an RCCL group)
        return self._cpu_group or None

    def setup_side_group(self):
        """Collective over the default group: create a gloo group next to it and keep it only if EVERY rank succeeded (a
        rank-local fallback would have ranks issuing different collectives for the same step: deadlock)."""
        import logging
        import os
        log = logging.getLogger(__name__)
        backend = dist.get_backend()
        if backend == "gloo":
            self._cpu_group = dist.group.WORLD
            return
        if backend != "nccl":                 # e.g. the in-process "threaded" group of the tests: no sockets to build on
            self._cpu_group = False
            return
        os.environ.setdefault("GLOO_SOCKET_IFNAME", "lo")
        group, err = None, None
        try:
            group = dist.new_group(backend="gloo")
        except Exception as e:      # noqa: BLE001 - rendezvous / interface problems differ per platform
            err = e
        ok = torch.tensor([0 if group is None else 1], dtype=torch.int32, device=self.queue.device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 1:
            self._cpu_group = group
            log.info("rspnet_amd: gloo side group created; shuffle permutations travel host-side (no per-step device sync)")
        else:
            self._cpu_group = False
            log.warning("rspnet_amd: gloo side group unavailable on at least one rank (%s); shuffle permutations are broadcast "
                        "through the device group: one small device sync per step", err)

    def _draw_step_randomness(self, B: int):
        """This step's (speed, idx_shuffle #1, idx_shuffle #2).  Drawn on every rank in the reference's call order
        (random.choice :430, torch.randperm :372 twice); rank 0's values win (:375-378 broadcasts idx_shuffle; the speed must be
        rank-shared as well, or T_real — hence every all-to-all / all-gather shape — would differ across ranks when
        diff_speed has several entries).  One host-side broadcast carries all three."""
        rank, ws, coll = self._dp()
        speed = random.choice(self.diff_speed)
        sh1, sh2 = torch.randperm(B * ws), torch.randperm(B * ws)
        if coll:
            msg = torch.cat([torch.tensor([speed], dtype=torch.int64), sh1.to(torch.int64), sh2.to(torch.int64)])
            group = self._cpu_pg()
            if group is not None:
                dist.broadcast(msg, src=0, group=group)
            else:
                dev_msg = msg.to(self.queue.device)
                dist.broadcast(dev_msg, src=0)
                msg = dev_msg.cpu()
            speed = int(msg[0])
            sh1, sh2 = msg[1:1 + B * ws], msg[1 + B * ws:]
        return speed, sh1.numpy().astype(np.int64), sh2.numpy().astype(np.int64)

    @staticmethod
    def _exchange_plan(idx: np.ndarray, B: int, rank: int, ws: int, splits: bool = False):
        """Host arithmetic of one shuffle-BN exchange under the global permutation `idx` (:361-387): which of my clips go
        where (send order, all-to-all splits), where every global sample ends up (for the un-shuffle) and the position each
        arriving clip has in the reference's shuffled batch.  `splits`: build the all-to-all split lists at one rank too
        (force_collectives)."""
        G = idx.reshape(ws, B)
        owner = G // B
        if ws == 1 and not splits:
            send_src, in_splits, out_splits = idx.astype(np.int32), None, None
        else:
            parts, in_splits = [], []
            for r in range(ws):
                mine = G[r][owner[r] == rank] - rank * B
                parts.append(mine)
                in_splits.append(int(mine.size))
            send_src = np.concatenate(parts).astype(np.int32)
            out_splits = [int((owner[rank] == s).sum()) for s in range(ws)]
        # where does global sample g end up?  rank r = position(g)//B; inside r's batch, clips arrive ordered by
        # source rank, then by their order in G[r]
        loc = np.empty(B * ws, dtype=np.int32)
        for r in range(ws):
            order = np.argsort(owner[r], kind="stable")
            loc[G[r][order]] = r * B + np.arange(B)
        arrival = np.argsort(owner[rank], kind="stable") if ws > 1 else np.arange(B)
        return send_src, loc, in_splits, out_splits, arrival

    @staticmethod
    def _upload_indices(arrays, dev):
        """All of a step's host-computed index vectors in ONE pinned staging buffer and ONE asynchronous copy.  (A copy from
        pageable memory — `torch.from_numpy(a).to(dev)`, or a Python scalar assigned through a tensor index — makes the HIP
        runtime wait for the stream to drain first: the host then never runs ahead of the GPU and every host hiccup becomes
        GPU idle time.  The pinned blocks come from torch's caching host allocator, which recycles a block only after the copy
        that read it has completed.)"""
        total = sum(int(a.size) for a in arrays)
        host = torch.empty(total, dtype=torch.int32, pin_memory=(dev.type == "cuda"))
        hv, off = host.numpy(), 0
        for a in arrays:
            hv[off:off + a.size] = a
            off += a.size
        d = host.to(dev, non_blocking=True)
        out, off = [], 0
        for a in arrays:
            out.append(d[off:off + a.size])
            off += a.size
        return out

    @torch.no_grad()
    def _shuffle_exchange(self, im: Tensor, step: Tensor, T_out: int, plan, src: Tensor):
        """First half of a key pass: shuffle-BN's sample exchange (:361-387).  The clips are sub-sampled first (each has exactly
        one destination rank) and travel in ONE all-to-all, issued asynchronously: both key passes' exchanges are started back
        to back at the top of the step — the permutations are known then — so the second one runs over xGMI under the first
        key pass's convolutions.  Returns the state `_key_pass` consumes."""
        be = _ops.backend()
        _, _, coll = self._dp()
        _, _, in_splits, out_splits, arrival = plan
        handle = None
        x = be.clip_gather(im, src, step[src.long()].contiguous(), T_out, max(im.shape[1], INPUT_CHANNEL_PAD))
        if coll:
            xs, x = x, torch.empty_like(x)
            handle = dist.all_to_all_single(x, xs, out_splits, in_splits, async_op=True)
        return x, handle, arrival

    @torch.no_grad()
    def _deferred_k(self):
        """{id(BatchNorm of encoder_k): its [2][C] batch-moment buffer} + the set that applies the deferred updates.  The second
        key pass of a step reports its batch moments there instead of moving the running statistics itself: the two passes go
        through the same BatchNorm buffers, and with the second one's update applied afterwards (`BnEmaSet.run`, one launch)
        they can run side by side inside a captured graph and still leave the buffers as two consecutive forwards do."""
        bns = self._bns(self.encoder_k)
        ptrs = [(b.running_mean.data_ptr(), b.running_var.data_ptr()) for b in bns]
        if self._ema_k is None or self._ema_k.ptrs != ptrs:
            self._ema_k = _ops.backend().bn_ema_set([(b.running_mean, b.running_var, float(b.momentum)) for b in bns])
            self._ema_map = {id(b): self._ema_k.stats[i] for i, b in enumerate(bns)}
        return self._ema_map

    def _key_pass(self, exchange, tag: str, deferred=None, bump: bool = True):
        """Second half (:408-419): encoder_k on the exchanged clips.  Returns this rank's fused (A | M) features in ARRIVAL
        order and the width of the A part; `_gather_keys` un-shuffles them."""
        x, handle, arrival = exchange
        if handle is not None:
            with self._comm("all_to_all_" + tag):
                handle.wait()
        if bump:
            self._nbt_k += 1
        a, m, _ = self.encoder_k.forward_ndhwc(x, keep=False, deferred=deferred)
        feats = torch.cat([a, m], dim=1)
        # introspection only (tests compare with the reference's encoder_k outputs): this rank's key features in arrival
        # order + the position each arrival has in the reference's shuffled batch G[rank]
        self._last_k[0 if tag == "kneg" else 1] = (feats, arrival)
        return feats, a.shape[1]

    @staticmethod
    def _pair_rows(loc: np.ndarray, B: int, which: int) -> np.ndarray:
        """Row of global sample g of key pass `which` (0: k_negative, 1: k) in the gathered (ws, 2, B, width) feature block,
        given its row `loc[g]` in that pass's own rank-major (ws, B) order."""
        return ((loc // B) * (2 * B) + which * B + loc % B).astype(np.int32)

    def _gather_keys(self, feats_neg: Tensor, feats_k: Tensor, rows_neg: Tensor, rows_k: Tensor):
        """ONE all-gather per step carries the fused (A | M) features of BOTH key passes: it serves the two un-shuffles
        (:389-406, twice) and the queue's key all-gather (:348) — five small collectives in the reference.  Returns
        (k_negative features of all samples in global order, k features of all samples in global order)."""
        be = _ops.backend()
        _, ws, coll = self._dp()
        B = feats_neg.shape[0]
        mine = torch.stack([feats_neg, feats_k])                    # (2, B, width)
        if coll:
            gathered = torch.empty((ws * 2,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device)
            with self._comm("all_gather_keys"):
                dist.all_gather_into_tensor(gathered, mine)
        else:
            gathered = mine
        flat = gathered.view(-1, mine.shape[2])
        return be.rows_gather(flat, rows_neg), be.rows_gather(flat, rows_k)

    def _forward_encoder_k(self, im: Tensor, step: Tensor, T_out: int, idx: np.ndarray):
        """One key pass with shuffle-BN (:408-419) under the global permutation `idx` (exchange + pass + un-shuffle, back to
        back).  Returns (features [A | M] of my samples in my order, features of ALL samples in global order, width of A)."""
        rank, ws, coll = self._dp()
        B = im.shape[0]
        plan = self._exchange_plan(idx, B, rank, ws, splits=coll)
        src, rows, rows2 = self._upload_indices([plan[0], self._pair_rows(plan[1], B, 0), self._pair_rows(plan[1], B, 1)], im.device)
        self._last_k = [None, None]
        feats, dim = self._key_pass(self._shuffle_exchange(im, step, T_out, plan, src), "k")
        all_feats, _ = self._gather_keys(feats, feats, rows, rows2)
        return all_feats[rank * B:(rank + 1) * B], all_feats, dim

    def _comm(self, name: str):
        """Context that books the time the compute stream (CPU backends: the host) is stalled by a collective under `name` in
        `self.comm_log` (a dict the caller installs, e.g. bench.py at N > 1); a no-op otherwise."""
        return _CommTimer(self.comm_log, name, self.queue.device)

    def _check_queue_ptr(self, n: int):
        """The device-side enqueue (rsp_queue_enqueue_dev) writes a slab only if it fits [ptr, ptr + n) inside the queue; a
        pointer that is not a multiple of the global batch — a checkpoint saved with another batch size or world size — would
        silently stop the queue from being refreshed.  The reference's slice assignment (:356) raises there; so does this, once
        per loaded state (one host read of the pointer)."""
        ptr = int(self.queue_ptr)
        if ptr < 0 or ptr >= self.K or ptr % n != 0:
            raise ValueError(f"queue_ptr = {ptr} is not a multiple of the global batch {n} below K = {self.K}: the queue state "
                             "was saved with another batch size / world size (reference :353-356 fails on the slab shape here)")
        self._ptr_checked = True

    @torch.no_grad()
    def _dequeue_and_enqueue(self, keys_all: Tensor):
        """:345-359 — keys_all is already the rank-ordered concat of k_neg_A over all ranks."""
        n = keys_all.shape[0]
        assert self.K % n == 0  # for simplicity (reference :353)
        if self._ptr_on_device:
            # graph-captured steps (rspnet_amd/graph_step.py): the pointer is read and advanced on the device
            if not self._ptr_checked and not (keys_all.is_cuda and torch.cuda.is_current_stream_capturing()):
                self._check_queue_ptr(n)
            _ops.backend().queue_enqueue_dev(self.queue, self.queue_ptr, keys_all.contiguous())
            self._ptr_host = None
            return
        if self._ptr_host is None:
            # same validation as the device-pointer path (stricter than the reference, which fails only at the step whose slab no
            # longer fits, :353-356 — a misaligned pointer always reaches that step): both issue modes refuse the same checkpoints
            self._check_queue_ptr(n)
            self._ptr_host = int(self.queue_ptr)
        ptr = self._ptr_host
        _ops.backend().queue_enqueue(self.queue, ptr, keys_all.contiguous())
        ptr = (ptr + n) % self.K
        self._ptr_host = ptr
        self.queue_ptr.fill_(ptr)

    # ---- backward of the query encoder -------------------------------------------------------------------------------
    def _backward_encoder_q(self, ectx, dqA, dqM):
        """Backward of encoder_q into the flat gradient buffer; with the collectives on, each 32 MiB bucket's all-reduce is
        launched as soon as the layer that completes it has run (RCCL works on its own stream under the rest of the
        backward), then gradients are averaged as DDP does (moco/__init__.py:49-53)."""
        flat = self._flat
        _, ws, coll = self._dp()
        if self._defer_reduce:
            # segmented replay (rspnet_amd/graph_step.py): the backward is one captured graph; `_reduce_gradients` /
            # `_scale_gradients` follow it
            self.encoder_q.backward_ndhwc(ectx, dqA, dqM, flat.grad_of, None)
            flat.attach_grads()
            return
        buckets = flat.buckets(BUCKET_FLOATS) if coll else []
        handed_out, launched, handles = set(), set(), []

        def grad_of(p):
            handed_out.add(id(p))
            return flat.grad_of(p)

        def after(_node_index, grads_ready=None):
            # runs after each plan node: every gradient view handed out so far has been issued — on this stream or as a
            # weight-gradient task on the engine's side stream (engine.BranchStreams.side_task).  A finished bucket's all-reduce is
            # issued from a stream context ordered behind both (RCCL runs it behind the stream it is issued from); the backward
            # itself does not wait.
            for bi, (s, e, ids) in enumerate(buckets):
                if bi not in launched and all(pid in handed_out for pid in ids):
                    launched.add(bi)
                    with (grads_ready() if grads_ready is not None else contextlib.nullcontext()):
                        handles.append(dist.all_reduce(flat.g_flat[s:e], async_op=True))

        self.encoder_q.backward_ndhwc(ectx, dqA, dqM, grad_of, after if buckets else None)
        for bi, (s, e, ids) in enumerate(buckets):
            if bi not in launched:
                handles.append(dist.all_reduce(flat.g_flat[s:e], async_op=True))
        if handles:
            with self._comm("allreduce_wait"):
                for h in handles:
                    h.wait()
        if ws > 1:
            flat.g_flat.mul_(1.0 / ws)
        flat.attach_grads()

    @torch.no_grad()
    def _backward_iter(self):
        """The query encoder's backward as a generator (after a `loss.backward()` under `_defer_backward`): yields, after each plan
        node, the set of parameter ids whose gradient kernels have been ISSUED so far — inline, or handed to
        engine.BranchStreams.deferred.  The caller cuts the backward into graphs there and starts a gradient bucket's all-reduce
        once all of its parameters are in the set (rspnet_amd/graph_step.py)."""
        ectx, dqA, dqM = self._pending_bwd
        self._pending_bwd = None
        flat = self._flat
        handed_out = set()

        def grad_of(p):
            handed_out.add(id(p))
            return flat.grad_of(p)

        for _ in self.encoder_q.backward_ndhwc_iter(ectx, dqA, dqM, grad_of, None):
            yield handed_out
        flat.attach_grads()

    @torch.no_grad()
    def _reduce_gradients(self):
        """The gradient all-reduce of a step whose backward ran with `_defer_reduce`: the same 32 MiB buckets, issued back to back
        after the backward (a replayed graph cannot launch them from inside)."""
        flat = self._flat
        if not self._dp()[2]:
            return
        handles = [dist.all_reduce(flat.g_flat[s:e], async_op=True) for s, e, _ in flat.buckets(BUCKET_FLOATS)]
        with self._comm("allreduce_wait"):
            for h in handles:
                h.wait()

    @torch.no_grad()
    def _scale_gradients(self):
        """... and DDP's average (moco/__init__.py:49-53)."""
        ws = self._dp()[1]
        if ws > 1:
            self._flat.g_flat.mul_(1.0 / ws)

    # ---- forward ------------------------------------------------------------------------------------------------------
    def _host_part(self, B: int, dev, static=None):
        """Everything of a step that is decided on the host: the speed drawn from diff_speed, both shuffle-BN permutations, the
        exchange plans derived from them, and their index vectors on the device.  `static` (rspnet_amd/graph_step.py): a
        (pinned staging tensor, device tensor) pair of 4*B*ws int32 the index vectors are written to IN PLACE, so that a step
        captured in a HIP graph reads this step's permutations at replay."""
        rank, ws, coll = self._dp()
        speed, sh1, sh2 = self._draw_step_randomness(B)
        plan1, plan2 = self._exchange_plan(sh1, B, rank, ws, splits=coll), self._exchange_plan(sh2, B, rank, ws, splits=coll)
        # (src: my clips in send order; rows: where each global sample's features sit in the step's one gathered block)
        arrays = (plan1[0], self._pair_rows(plan1[1], B, 0), plan2[0], self._pair_rows(plan2[1], B, 1))
        if static is None:
            src1, loc1, src2, loc2 = self._upload_indices(arrays, dev)
        else:
            host, devt = static
            hv, off, views = host.numpy(), 0, []
            for a in arrays:
                hv[off:off + a.size] = a
                views.append(devt[off:off + a.size])
                off += a.size
            devt.copy_(host, non_blocking=True)
            src1, loc1, src2, loc2 = views
        return {"speed": speed, "sh": (sh1, sh2), "plans": (plan1, plan2), "idx": (src1, loc1, src2, loc2)}

    def forward(self, im_q: Tensor, im_k: Tensor):
        """im_q, im_k: (B, 3, T=speed*16, H, W) fp32 on this rank's device.  Returns
        ((logits1, logits2), labels_A, (l_pos_M, l_neg_M), labels_M) exactly as the reference (:492-547)."""
        return self._device_part(im_q, im_k, self._host_part(im_q.shape[0], im_q.device))

    def _device_part(self, im_q: Tensor, im_k: Tensor, host):
        """The step's device work under the host decisions of `_host_part`: its phases back to back, the collectives between them
        issued asynchronously (the second clip exchange runs under the first key pass, the key all-gather under the tail of the
        query forward).  rspnet_amd/graph_step.py replays the same phases as HIP graphs with the collectives in between."""
        self._broadcast_running_stats()
        st = self._phase_top(im_q, im_k, host)
        self._phase_exchange(st, host, wait=False)
        self._phase_passes(st, join_query=False)
        self._phase_gather(st)
        return self._phase_logits(st)

    # ---- the step in phases: device work between the collective points -------------------------------------------------------
    def _phase_top(self, im_q: Tensor, im_k: Tensor, host):
        """Momentum update (:337-343), _diff_speed (:421-447) and the three clip gathers: the send buffers of both shuffle-BN
        exchanges (each clip has exactly one destination rank) and the query clips.  Returns the step's state dict."""
        be = _ops.backend()
        self._prepare()
        dev = im_q.device
        B, C, T, H, W = im_q.shape
        im_q, im_k = im_q.contiguous(), im_k.contiguous()
        self._last_k = [None, None]          # (k_negative pass, k pass)
        _, _, coll = self._dp()
        with torch.no_grad():
            self._momentum_update_key_encoder()
            # _diff_speed (:421-447)
            random_indices = torch.randperm(B, device=dev)
            n1 = int(B * self.alpha)
            speed, (sh1, sh2) = host["speed"], host["sh"]
            self._last_speed = speed
            # introspection only (parity checks replay the step on the checker with the same draws): device tensor, not read here
            self._last_draw = (random_indices, speed, sh1, sh2)
            T_real = T // speed
            step_q = torch.full((B,), speed, dtype=torch.int32, device=dev)
            step_q.index_fill_(0, random_indices[:n1], 1)               # s1 rows play q,k at normal speed
            step_kn = (1 + speed) - step_q if speed != 1 else step_q.clone()   # k_negative swaps the speeds
            src1, loc1, src2, loc2 = host["idx"]
            cpad = max(C, INPUT_CHANNEL_PAD)
            # the key passes keep the reference's order (k_negative first, :445, then k, :512)
            src = torch.arange(B, dtype=torch.int32, device=dev)
            xs_neg, xs_k, x_q = be.clip_gather_multi([(im_k, src1, step_kn[src1.long()].contiguous()),
                                                      (im_k, src2, step_q[src2.long()].contiguous()), (im_q, src, step_q)], T_real, cpad)
            self._nbt_q += 1
            self._nbt_k += 2
            self.encoder_k._packed.refresh_now()      # (re-pack of the momentum-updated weights: before the passes fork)
            st = {"B": B, "dev": dev, "x_q": x_q, "send": (xs_neg, xs_k), "loc": (loc1, loc2), "handles": [None, None],
                  "feats": [None, None], "deferred": self._deferred_k(),
                  "arrival": [np.arange(B), np.arange(B)],
                  # with the collectives on, the clips arrive in buffers of their own (every rank receives exactly B clips)
                  "recv": (torch.empty_like(xs_neg), torch.empty_like(xs_k)) if coll else (xs_neg, xs_k)}
        return st

    @torch.no_grad()
    def _phase_exchange(self, st, host, wait):
        """Shuffle-BN's sample exchange (:361-387): ONE all-to-all per key pass over the already sub-sampled clips.  Both are
        started back to back — the permutations are known at the top of the step — so the second one runs over xGMI under the
        first key pass's convolutions.  wait=False: the key passes wait for their clips themselves (`_key_pass`); True: the current
        stream waits for both; "first": for the k_negative clips only (`_wait_exchange(st, 1)` follows on the k pass's stream)."""
        _, _, coll = self._dp()
        for i, plan in enumerate(host["plans"]):
            _, _, in_splits, out_splits, arrival = plan
            st["arrival"][i] = arrival
            if coll:
                st["handles"][i] = dist.all_to_all_single(st["recv"][i], st["send"][i], out_splits, in_splits, async_op=True)
        if wait:
            self._wait_exchange(st, 0)
            if wait != "first":
                self._wait_exchange(st, 1)

    def _wait_exchange(self, st, i: int):
        """The current stream waits for the clips of key pass i (0: k_negative, 1: k)."""
        h, st["handles"][i] = st["handles"][i], None
        if h is not None:
            with self._comm("all_to_all_" + ("k" if i else "kneg")):
                h.wait()

    @torch.no_grad()
    def _phase_passes(self, st, join_query: bool):
        """The three forward passes: query encoder (kept for backward), k_negative and k through encoder_k — forked onto streams of
        their own where that pays (inside a capture, or issued eagerly with the host far ahead)."""
        dev = st["dev"]
        # The query encoder's forward does not depend on the key passes (other weights, other BatchNorm buffers) before the
        # logits: it is forked onto its own stream and runs beside them — the small late layers of either pass leave most of
        # the machine idle on their own.  (Captured into a HIP graph, rspnet_amd/graph_step.py, or issued eagerly; any world size.)
        side = None
        if self.overlap_query and dev.type == "cuda" and (torch.cuda.is_current_stream_capturing() or self.overlap_query_eager):
            main = torch.cuda.current_stream(dev)
            side = self._query_stream = self._query_stream or _streams.lane(dev, "q")
            side.wait_stream(main)
            with torch.cuda.stream(side):
                self._pass_query(st)
        # The two key passes (k_negative first, :445; then k, :512) go through the same encoder_k.  The second one defers its
        # running-statistics update (_deferred_k), so it is forked onto its own stream beside the first (and beside the query
        # forward); the deferred update is applied after both, in the reference's order.  Neither pass holds a collective of
        # its own beyond the wait for its clips: the features of both travel in one all-gather after the join.
        side_k = None
        if side is not None and self.overlap_keys:
            main = torch.cuda.current_stream(dev)
            side_k = self._key_stream = self._key_stream or _streams.lane(dev, "k")
            side_k.wait_stream(main)
            with torch.cuda.stream(side_k):
                self._pass_key(st, 1)
        self._pass_key(st, 0)
        if side_k is None:
            self._pass_key(st, 1)
        else:
            torch.cuda.current_stream(dev).wait_stream(side_k)
        self._passes_join(st)
        st["side"] = side
        if side is None:
            self._pass_query(st)
        elif join_query:
            torch.cuda.current_stream(dev).wait_stream(side)
            st["side"] = None

    # the pieces of `_phase_passes`, each a plain kernel sequence on the current stream (rspnet_amd/graph_step.py captures each as a
    # LINEAR graph and replays the three passes side by side on streams of its own)
    @torch.no_grad()
    def _pass_query(self, st):
        self._q_pre = self.encoder_q.forward_ndhwc(st["x_q"], keep=True)

    @torch.no_grad()
    def _pass_key(self, st, which: int):
        """which = 0: k_negative (first in the reference, :445: moves encoder_k's running statistics itself); 1: k (:512: reports
        its batch moments to the deferred set, applied by `_passes_join` after both)."""
        ex = (st["recv"][which], st["handles"][which], st["arrival"][which])
        st["handles"][which] = None
        feats, dim = self._key_pass(ex, "k" if which else "kneg", deferred=st["deferred"] if which else None, bump=False)
        st["feats"][which] = feats
        st["dim"] = dim

    @torch.no_grad()
    def _passes_join(self, st):
        self._ema_k.run()
        _, ws, coll = self._dp()
        mine = torch.stack(st["feats"])                             # (2, B, width): k_negative, k
        st["feats"] = [None, None]
        st["mine"] = mine
        st["gathered"] = (torch.empty((ws * 2,) + tuple(mine.shape[1:]), dtype=mine.dtype, device=mine.device) if coll else mine)
        st.setdefault("side", None)

    @torch.no_grad()
    def _phase_gather(self, st):
        """ONE all-gather per step carries the fused (A | M) features of BOTH key passes: it serves the two un-shuffles
        (:389-406, twice) and the queue's key all-gather (:348) — five small collectives in the reference."""
        if self._dp()[2]:
            with self._comm("all_gather_keys"):
                dist.all_gather_into_tensor(st["gathered"], st["mine"])

    def _phase_logits(self, st):
        """Un-shuffle (:389-406), the five logits (:521-538) as the autograd node of the query encoder, enqueue (:345-359)."""
        be = _ops.backend()
        B, dev, dim = st["B"], st["dev"], st["dim"]
        with torch.no_grad():
            rank = self._dp()[0]
            flat = st["gathered"].view(-1, st["mine"].shape[2])
            kneg_all, k_all = be.rows_gather(flat, st["loc"][0]), be.rows_gather(flat, st["loc"][1])
            k_mine, kneg_mine = k_all[rank * B:(rank + 1) * B], kneg_all[rank * B:(rank + 1) * B]
            k_A, k_M = k_mine[:, :dim].contiguous(), k_mine[:, dim:].contiguous()
            kneg_A, kneg_M = kneg_mine[:, :dim].contiguous(), kneg_mine[:, dim:].contiguous()
            if st["side"] is not None:
                torch.cuda.current_stream(dev).wait_stream(st["side"])
                st["side"] = None

        l1, l2, lp, ln = _PretextFn.apply(self, st["x_q"], (k_A, k_M, kneg_A, kneg_M), *self._q_params)

        labels_A = torch.zeros(B, dtype=torch.long, device=dev)
        labels_M = torch.ones_like(labels_A)
        self._dequeue_and_enqueue(kneg_all[:, :dim])
        return (l1, l2), labels_A, (lp, ln), labels_M
