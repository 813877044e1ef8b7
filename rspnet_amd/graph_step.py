"""The pretext step as ONE HIP graph.

The reference's loop body (pretrain.py:157-165) —

    output, target, ranking_logits, ranking_target = model(clip_q, clip_k)
    loss, loss_A, loss_M = criterion(output, target, ranking_logits, ranking_target)
    optimizer.zero_grad(); loss.backward(); optimizer.step()

— issues 250 (C3D) to 2 000 (S3D-G) kernel launches from Python.  `GraphedPretextStep` runs exactly these five statements under
HIP stream capture once per (speed, learning rate, input shape) and replays the captured graph afterwards: no Python and no
launch call per kernel, and — since a graph carries dependencies instead of one stream's order — S3D-G's independent inception
branches (models/s3dg.py:80-99) are forked onto side streams inside the capture and run next to each other
(engine.run_forward / run_backward; measured on one sepInc block, tools/stream_overlap_probe.py: x1.14-1.31 as a graph, x1.0
issued eagerly).

What stays on the host per step is what is DECIDED there: the speed drawn from diff_speed and the two shuffle-BN permutations
(builder_diffspeed_diffloss.py:372,430 — `MoCoDiffLossTwoFc._host_part`); their index vectors are written into a static
pinned buffer and copied to a static device buffer before the replay.  Everything else of the reference step is inside the
graph: momentum update, the device-side randperm of _diff_speed (graph-safe Philox offsets), both key passes, query forward,
logits, losses, backward, SGD, enqueue (pointer read and advanced on the device: rsp_queue_enqueue_dev).

With the data-parallel collectives on (more than one rank, or `force_collectives`) the step is replayed in SEGMENTS: RCCL calls
cannot be captured into the step graph on this stack (hipStreamEndCapture segfaults, profiles/r04/experiments_r4.txt), and the step
has exactly four collective points — the two clip all-to-alls at the top, the fused key all-gather after the three forward passes,
the gradient all-reduce after the backward.  The device work BETWEEN them is captured as four graphs

    top      momentum update, _diff_speed, the three clip gathers                   -> all-to-all x2   (eager, RCCL)
    passes   query forward | k_negative pass | k pass on three forked streams       -> all-gather      (eager, RCCL)
    tail     un-shuffle, logits, losses, enqueue, the whole backward                -> all-reduce      (eager, RCCL, 32 MiB buckets)
    update   DDP's 1/ws average, SGD

and the host issues four replays and four-plus collective calls per step instead of 250-2 000 launches (S3D-G eagerly: 38.9 ms of
host time per 39.5 ms step with ONE interpreter on an idle host; eight ranks on one node share that host).  What the segments give
up against the eager data-parallel step: the second clip exchange no longer hides under the first key pass and the bucket
all-reduces no longer start inside the backward — a fraction of a millisecond each over xGMI, the price of not being
Python-bound.  The same `issue="auto"` policy picks between the two.  Any failure to capture falls back to the eager loop with a
logged warning — the result is the same either way, kernel for kernel.

WHEN the graph is used (`issue="auto"`): only for a step the host cannot issue fast enough.  A replayed graph removes Python and
the launch calls, but its nodes reach the GPU with more dependency bookkeeping than a stream's in-order launches: replayed, the
same step is 0.5-2.4 % SLOWER than issued eagerly with its side streams (C3D 347.7 vs 352.2 clips/s, R3D-18 1251 vs 1281, R(2+1)D 428
vs 431, S3D-G 404 vs 406; profiles/r04/experiments_r4.txt) — as long as the host keeps ahead.  The last eager warm-up step of a
configuration is therefore measured: host time to issue it against GPU time to run it.  Below HOST_BOUND (half) the configuration
stays eager; above it (S3D-G: 38 of 39 ms — any slower host and the GPU would wait for Python) it is captured.
"""
from __future__ import annotations

import logging
from typing import Dict, Tuple

import torch
import torch.distributed as dist

log = logging.getLogger(__name__)


class GraphedPretextStep:
    RING = 8
    MAX_GRAPHS = 4      # configurations kept (diff_speed has at most three entries; a new learning rate retires the old graphs)
    HOST_BOUND = 0.5    # issue="auto": capture a configuration whose eager issue takes more than this share of its GPU time
    MAX_KEYS = 64       # bookkeeping entries kept per dictionary (a per-iteration LR schedule would otherwise grow them without bound)

    def __init__(self, model, criterion, optimizer, warmup: int = 2, issue: str = "auto"):
        if issue not in ("auto", "graph"):
            raise ValueError("GraphedPretextStep: issue is 'auto' (graph only when the host is the limiter) or 'graph' (always)")
        self.issue = issue
        # configurations = (speed, clip shape): the learning rate is a kernel argument of the captured SGD launch, so a GRAPH is
        # per (configuration, learning rates) — but the warm-up steps and the eager-or-graph decision are per configuration
        self.eager_keys: Dict[Tuple, str] = {}    # configurations that stay eager under issue="auto", with the measurement
        self.wrapped = model
        self.model = getattr(model, "module", model)
        self.criterion, self.optimizer = criterion, optimizer
        # at least two eager steps: the packed-weight sets an encoder builds during its first step are merged into one batched
        # set (a host-to-device copy of the job table) when its second step re-packs them
        self.warmup = max(2, int(warmup))
        if issue == "auto":
            # ... and the steps that are MEASURED must be ordinary ones: from the third on (the second still merges the packed-weight
            # sets; on a cold box its host time once tipped a 74 ms R(2+1)D step over the threshold)
            self.warmup = max(5, self.warmup)
        self.graphs: Dict[Tuple, Tuple] = {}      # insertion order = least recently used first
        self.eager_steps: Dict[Tuple, int] = {}
        self.samples: Dict[Tuple, list] = {}      # issue="auto": (host ms, GPU ms) of the measured warm-up steps of a configuration
        self.static = None
        self.last_wait_s = 0.0
        self.pool = None                          # one memory pool for all graphs of this stepper: only one replays at a time
        # With the data-parallel collectives on, the step is replayed as SEGMENTS between the collective points (module docstring).
        # RSP_GRAPH_COLLECTIVES=1 captures the collectives into one whole-step graph instead (exercised at one rank with
        # force_collectives; never the default), RSP_NO_SEGMENTS=1 issues the data-parallel step eagerly as rounds 1-4 did (A/B).
        import os
        coll = bool(self.model._dp()[2])
        self.mode = "segments" if (coll and not os.environ.get("RSP_GRAPH_COLLECTIVES")) else "whole"
        self.disabled = coll and bool(os.environ.get("RSP_NO_SEGMENTS")) and not os.environ.get("RSP_GRAPH_COLLECTIVES")
        self.fallback_reason = "data-parallel collectives on, RSP_NO_SEGMENTS (issued eagerly)" if self.disabled else None

    # ---- the five statements ------------------------------------------------------------------------------------------------
    def _eager(self, im_q, im_k, host):
        out, tgt, rl, rt = self.model._device_part(im_q, im_k, host)
        loss, loss_A, loss_M = self.criterion(out, tgt, rl, rt)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss, loss_A, loss_M, out, rl, tgt, rt

    def _segments(self, im_q, im_k):
        """The same five statements cut at the collective points: [(kind, fn(host))] with kind 'graph' (device work: captured once,
        replayed) or 'eager' (a collective: issued every step with that step's split lists).  `box` carries the step's state
        between them — static buffers of the graphs' memory pool once captured."""
        m, box = self.model, {}

        def top(host):
            box["st"] = m._phase_top(im_q, im_k, host)

        def exchange(host):
            m._phase_exchange(box["st"], host, wait=True)
            m._last_draw = (m._last_draw[0], host["speed"]) + tuple(host["sh"])

        def passes(host):
            m._phase_passes(box["st"], join_query=True)

        def gather(host):
            m._phase_gather(box["st"])

        def tail(host):
            out, tgt, rl, rt = m._phase_logits(box["st"])
            loss, loss_A, loss_M = self.criterion(out, tgt, rl, rt)
            self.optimizer.zero_grad()
            loss.backward()
            box["outs"] = (loss, loss_A, loss_M, out, rl)

        def reduce(host):
            m._reduce_gradients()

        def update(host):
            m._scale_gradients()
            self.optimizer.step()

        return [("graph", top), ("eager", exchange), ("graph", passes), ("eager", gather), ("graph", tail), ("eager", reduce),
                ("graph", update)], box

    def _config(self, im_q, host):
        return (host["speed"], tuple(im_q.shape))

    def _key(self, im_q, host):
        lrs = tuple(float(g["lr"]) for g in self.optimizer.param_groups)
        return self._config(im_q, host) + (lrs,)

    @staticmethod
    def _cap(d, n):
        while len(d) > n:
            d.pop(next(iter(d)))

    def __call__(self, im_q, im_k):
        """One training step.  Returns (loss, loss_A, loss_M, output, ranking_logits) — the tensors of a replayed step are the
        graph's own output buffers: read them before the next call."""
        m = self.model
        B, dev = im_q.shape[0], im_q.device
        if self.disabled or dev.type != "cuda":
            return self._eager(im_q, im_k, m._host_part(B, dev))[:5]
        if self.static is None:
            n = 4 * B * (dist.get_world_size() if dist.is_initialized() else 1)
            # the host runs several steps ahead of the GPU: a ring of pinned staging buffers, each guarded by the event of the
            # copy that last read it (re-filling a slot waits for that copy — back-pressure only if the GPU is RING steps behind)
            self.static = {"ring": [[torch.empty(n, dtype=torch.int32, pin_memory=True), None] for _ in range(self.RING)],
                           "turn": 0, "dev": torch.empty(n, dtype=torch.int32, device=dev), "im_q": None, "im_k": None,
                           "shape": tuple(im_q.shape)}
            m._ptr_on_device = True
            m._ptr_host = None
        st = self.static
        if not m._ptr_checked:
            m._check_queue_ptr(B * (dist.get_world_size() if dist.is_initialized() else 1))
        if tuple(im_q.shape) != st["shape"]:
            raise ValueError("GraphedPretextStep: the clip shape changed; build a new GraphedPretextStep for it")
        slot = st["ring"][st["turn"] % self.RING]
        st["turn"] += 1
        self.last_wait_s = 0.0
        if slot[1] is not None:
            # back-pressure: the host may run at most RING steps ahead of the GPU.  The time spent here is the GPU's, not the
            # host's: `last_wait_s` lets a caller separate it from the submission cost proper (bench.py: steps_ms.host_submit_*)
            import time
            t0 = time.perf_counter()
            slot[1].synchronize()
            self.last_wait_s = time.perf_counter() - t0
        host = m._host_part(B, dev, static=(slot[0], st["dev"]))
        slot[1] = torch.cuda.Event()
        slot[1].record()
        cfg, key = self._config(im_q, host), self._key(im_q, host)
        entry = self.graphs.get(key)
        if entry is None:
            # the first steps of a configuration run eagerly (lazy initialisation inside the library, allocator pools, the
            # optimizer's momentum buffers, packed-weight sets) on the caller's own clip tensors; then the configuration is
            # captured — or kept eager by policy, and then never pays for the copy into the graphs' static clip buffers
            if cfg in self.eager_keys:
                return self._eager(im_q, im_k, host)[:5]
            done = self.eager_steps.get(cfg, 0)
            if done < self.warmup:
                self.eager_steps[cfg] = done + 1
                self._cap(self.eager_steps, self.MAX_KEYS)
                if done >= 2 and self.issue == "auto":
                    return self._measured_eager(cfg, host, im_q, im_k, last=(done == self.warmup - 1))[:5]
                return self._eager(im_q, im_k, host)[:5]
            self._static_clips(im_q, im_k)
            entry = self._capture(key, host)
            if entry is None:
                return self._eager(im_q, im_k, host)[:5]
        else:
            self._static_clips(im_q, im_k)
        self.graphs[key] = self.graphs.pop(key)         # most recently used last
        if entry[3] is None:
            entry[0].replay()
        else:
            for g, fn in entry[3]:
                if g is not None:
                    g.replay()
                else:
                    fn(host)
        return entry[1]

    def _static_clips(self, im_q, im_k):
        st = self.static
        if st["im_q"] is None:
            st["im_q"], st["im_k"] = torch.empty_like(im_q), torch.empty_like(im_k)
        if im_q.data_ptr() != st["im_q"].data_ptr():
            st["im_q"].copy_(im_q, non_blocking=True)
        if im_k.data_ptr() != st["im_k"].data_ptr():
            st["im_k"].copy_(im_k, non_blocking=True)

    def _measured_eager(self, cfg, host, im_q, im_k, last: bool):
        """An eager warm-up step of a configuration (from its third on), timed on both sides with an empty queue at its start: does
        the host keep ahead?  Decided on the MEDIAN host share of the measured steps (three by default): one host hiccup (a
        collection, a cold page) on a step near the threshold does not flip the choice between runs."""
        import time
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        out = self._eager(im_q, im_k, host)
        host_ms = (time.perf_counter() - t0) * 1e3
        e1.record()
        e1.synchronize()
        gpu_ms = e0.elapsed_time(e1)
        got = self.samples.setdefault(cfg, [])
        got.append((host_ms, gpu_ms))
        self._cap(self.samples, self.MAX_KEYS)
        if last:
            shares = sorted(h / max(g, 1e-6) for h, g in got)
            share = shares[len(shares) // 2]
            h_med, g_med = sorted(h for h, _ in got)[len(got) // 2], sorted(g for _, g in got)[len(got) // 2]
            if share <= self.HOST_BOUND:
                self.eager_keys[cfg] = (f"issued eagerly by policy: the host issues this step in {h_med:.1f} ms of the {g_med:.1f} ms it runs "
                                        f"(median of {len(got)} measured warm-up steps; a replayed graph pays off above {self.HOST_BOUND:.0%})")
                self._cap(self.eager_keys, self.MAX_KEYS)
                self.fallback_reason = self.eager_keys[cfg]
                log.info("rspnet_amd: pretext step (speed %s) %s", cfg[0], self.eager_keys[cfg])
        return out

    def _capture(self, key, host):
        from . import ops as _ops
        st = self.static
        be = _ops.backend()
        if getattr(be, "event_log", None) is not None:
            return None                                  # per-launch timing events cannot be recorded inside a capture
        m = self.model
        try:
            torch.cuda.synchronize()
            from .engine import BranchStreams
            while len(self.graphs) >= self.MAX_GRAPHS:      # (the scheduler changes the learning rate every epoch: old graphs go)
                self.graphs.pop(next(iter(self.graphs)))
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            dev = st["dev"].device
            if self.mode == "whole":
                g = torch.cuda.CUDAGraph()
                try:
                    with torch.cuda.graph(g, pool=self.pool):
                        BranchStreams.origin = torch.cuda.current_stream(dev).cuda_stream
                        loss, loss_A, loss_M, out, rl, _, _ = self._eager(st["im_q"], st["im_k"], host)
                finally:
                    BranchStreams.origin = None
                outs, seq = (loss, loss_A, loss_M, out, rl), None
            else:
                # one graph per device segment, captured in step order into ONE pool (they replay in that order, never side by
                # side); the collectives between them are not executed now — nothing is, a capture only records — the replay
                # below runs the step this capture stands for
                segs, box = self._segments(st["im_q"], st["im_k"])
                seq, g = [], None
                m._defer_reduce = True
                try:
                    for kind, fn in segs:
                        if kind == "eager":
                            seq.append((None, fn))
                            continue
                        sg = torch.cuda.CUDAGraph()
                        try:
                            with torch.cuda.graph(sg, pool=self.pool):
                                BranchStreams.origin = torch.cuda.current_stream(dev).cuda_stream
                                fn(host)
                        finally:
                            BranchStreams.origin = None
                        seq.append((sg, None))
                finally:
                    m._defer_reduce = False
                outs = box["outs"]
                st.setdefault("boxes", []).append(box)       # the segments' shared state: static buffers of the pool
            # Everything the graph's kernels address that was allocated OUTSIDE the capture must outlive the graph: a later
            # configuration may rebuild a packed-weight set, and the superseded buffers — still baked into this graph's kernel
            # arguments — would be freed.  (The library's scratch buffers come from the graph's own pool: ops.HipOps._workspace.)
            # (also baked into captured kernel arguments: the deferred BatchNorm-update set of encoder_k — its moment buffer and job
            #  table — and the virtual-pixel stems' derived filters and index vectors; both are rebuilt when pointers / devices change)
            keep = [list(m.encoder_q._packed._sets), list(m.encoder_k._packed._sets), m._flat,
                    getattr(m._flat, "m_flat", None), getattr(m, "_nbt_q", None), getattr(m, "_nbt_k", None), st,
                    m._ema_k, m._ema_map, list(m.encoder_q._packed._virtual.values()), list(m.encoder_k._packed._virtual.values())]
            self.graphs[key] = (g, outs, keep, seq)
            log.info("rspnet_amd: pretext step captured as %s (speed %s, clips %s)",
                     "a HIP graph" if seq is None else f"{sum(1 for s in seq if s[0] is not None)} HIP-graph segments between its collectives",
                     key[0], key[1])
            # the capture itself executed nothing: run the step it stands for
            return self.graphs[key]
        except Exception as e:      # noqa: BLE001 - whatever refuses the capture, the eager loop still works
            self.disabled, self.fallback_reason = True, f"{type(e).__name__}: {e}"
            log.warning("rspnet_amd: HIP-graph capture of the pretext step failed (%s); running eagerly", self.fallback_reason)
            torch.cuda.synchronize()
            return None
