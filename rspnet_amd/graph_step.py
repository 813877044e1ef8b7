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

Single rank by default (the collectives of the data-parallel path are issued eagerly, with the same side streams); any failure to
capture falls back to the eager loop with a logged warning — the result is the same either way, kernel for kernel.

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

    def __init__(self, model, criterion, optimizer, warmup: int = 2, issue: str = "auto"):
        if issue not in ("auto", "graph"):
            raise ValueError("GraphedPretextStep: issue is 'auto' (graph only when the host is the limiter) or 'graph' (always)")
        self.issue = issue
        self.eager_keys: Dict[Tuple, str] = {}    # configurations that stay eager under issue="auto", with the measurement
        self.wrapped = model
        self.model = getattr(model, "module", model)
        self.criterion, self.optimizer = criterion, optimizer
        # at least two eager steps: the packed-weight sets an encoder builds during its first step are merged into one batched
        # set (a host-to-device copy of the job table) when its second step re-packs them
        self.warmup = max(2, int(warmup))
        if issue == "auto":
            # ... and the step that is MEASURED must be an ordinary one: the third (the second still merges the packed-weight sets;
            # on a cold box its host time once tipped a 74 ms R(2+1)D step over the threshold)
            self.warmup = max(3, self.warmup)
        self.graphs: Dict[Tuple, Tuple] = {}      # insertion order = least recently used first
        self.eager_steps: Dict[Tuple, int] = {}
        self.static = None
        self.last_wait_s = 0.0
        self.pool = None                          # one memory pool for all graphs of this stepper: only one replays at a time
        # with the data-parallel collectives on, the step is issued eagerly — same kernels, same side streams (query forward, second
        # key pass, small weight gradients), RCCL on its own stream.  RSP_GRAPH_COLLECTIVES=1 captures the collectives too
        # (exercised at one rank with force_collectives; not the default at N > 1, where it has never run).
        import os
        self.disabled = bool(self.model._dp()[2]) and not os.environ.get("RSP_GRAPH_COLLECTIVES")
        self.fallback_reason = "data-parallel collectives on (issued eagerly)" if self.disabled else None

    # ---- the five statements ------------------------------------------------------------------------------------------------
    def _eager(self, im_q, im_k, host):
        out, tgt, rl, rt = self.model._device_part(im_q, im_k, host)
        loss, loss_A, loss_M = self.criterion(out, tgt, rl, rt)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss, loss_A, loss_M, out, rl, tgt, rt

    def _key(self, im_q, host):
        lrs = tuple(float(g["lr"]) for g in self.optimizer.param_groups)
        return (host["speed"], tuple(im_q.shape), lrs)

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
                           "turn": 0, "dev": torch.empty(n, dtype=torch.int32, device=dev),
                           "im_q": torch.empty_like(im_q), "im_k": torch.empty_like(im_k)}
            m._ptr_on_device = True
            m._ptr_host = None
        st = self.static
        if not m._ptr_checked:
            m._check_queue_ptr(B * (dist.get_world_size() if dist.is_initialized() else 1))
        if im_q.shape != st["im_q"].shape:
            raise ValueError("GraphedPretextStep: the clip shape changed; build a new GraphedPretextStep for it")
        if im_q.data_ptr() != st["im_q"].data_ptr():
            st["im_q"].copy_(im_q, non_blocking=True)
        if im_k.data_ptr() != st["im_k"].data_ptr():
            st["im_k"].copy_(im_k, non_blocking=True)
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
        key = self._key(im_q, host)
        entry = self.graphs.get(key)
        if entry is None:
            # the first steps of a configuration run eagerly (lazy initialisation inside the library, allocator pools, the
            # optimizer's momentum buffers, packed-weight sets), then the configuration is captured
            done = self.eager_steps.get(key, 0)
            if key in self.eager_keys:
                return self._eager(st["im_q"], st["im_k"], host)[:5]
            if done < self.warmup:
                self.eager_steps[key] = done + 1
                if done == self.warmup - 1 and self.issue == "auto":
                    return self._measured_eager(key, host)[:5]
                return self._eager(st["im_q"], st["im_k"], host)[:5]
            entry = self._capture(key, host)
            if entry is None:
                return self._eager(st["im_q"], st["im_k"], host)[:5]
        self.graphs[key] = self.graphs.pop(key)         # most recently used last
        entry[0].replay()
        return entry[1]

    def _measured_eager(self, key, host):
        """The last eager warm-up step of a configuration, timed on both sides (empty queue at its start): does the host keep ahead?"""
        import time
        st = self.static
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        t0 = time.perf_counter()
        out = self._eager(st["im_q"], st["im_k"], host)
        host_ms = (time.perf_counter() - t0) * 1e3
        e1.record()
        e1.synchronize()
        gpu_ms = e0.elapsed_time(e1)
        if host_ms <= self.HOST_BOUND * gpu_ms:
            self.eager_keys[key] = (f"issued eagerly by policy: the host issues this step in {host_ms:.1f} ms of the {gpu_ms:.1f} ms it runs "
                                    f"(a replayed graph pays off above {self.HOST_BOUND:.0%})")
            self.fallback_reason = self.eager_keys[key]
            log.info("rspnet_amd: pretext step (speed %s) %s", key[0], self.eager_keys[key])
        return out

    def _capture(self, key, host):
        from . import ops as _ops
        st = self.static
        be = _ops.backend()
        if getattr(be, "event_log", None) is not None:
            return None                                  # per-launch timing events cannot be recorded inside a capture
        try:
            torch.cuda.synchronize()
            from .engine import BranchStreams
            while len(self.graphs) >= self.MAX_GRAPHS:      # (the scheduler changes the learning rate every epoch: old graphs go)
                self.graphs.pop(next(iter(self.graphs)))
            if self.pool is None:
                self.pool = torch.cuda.graph_pool_handle()
            g = torch.cuda.CUDAGraph()
            try:
                with torch.cuda.graph(g, pool=self.pool):
                    BranchStreams.origin = torch.cuda.current_stream(st["dev"].device).cuda_stream
                    loss, loss_A, loss_M, out, rl, _, _ = self._eager(st["im_q"], st["im_k"], host)
            finally:
                BranchStreams.origin = None
            outs = (loss, loss_A, loss_M, out, rl)
            # Everything the graph's kernels address that was allocated OUTSIDE the capture must outlive the graph: a later
            # configuration may rebuild a packed-weight set, and the superseded buffers — still baked into this graph's kernel
            # arguments — would be freed.  (The library's scratch buffers come from the graph's own pool: ops.HipOps._workspace.)
            m = self.model
            # (also baked into captured kernel arguments: the deferred BatchNorm-update set of encoder_k — its moment buffer and job
            #  table — and the virtual-pixel stems' derived filters and index vectors; both are rebuilt when pointers / devices change)
            keep = [list(m.encoder_q._packed._sets), list(m.encoder_k._packed._sets), m._flat,
                    getattr(m._flat, "m_flat", None), getattr(m, "_nbt_q", None), getattr(m, "_nbt_k", None), st,
                    m._ema_k, m._ema_map, list(m.encoder_q._packed._virtual.values()), list(m.encoder_k._packed._virtual.values())]
            self.graphs[key] = (g, outs, keep)
            log.info("rspnet_amd: pretext step captured as a HIP graph (speed %s, clips %s)", key[0], key[1])
            # the capture itself executed nothing: run the step it stands for
            return self.graphs[key]
        except Exception as e:      # noqa: BLE001 - whatever refuses the capture, the eager loop still works
            self.disabled, self.fallback_reason = True, f"{type(e).__name__}: {e}"
            log.warning("rspnet_amd: HIP-graph capture of the pretext step failed (%s); running eagerly", self.fallback_reason)
            torch.cuda.synchronize()
            return None
