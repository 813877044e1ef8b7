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

HOW the step is captured (`mode`, default "lanes").  Measured on this stack (tools/chain_gap_probe.py, profiles/r05): a LINEAR graph
— every node depends on the previous one — is handed to the GPU in one piece: 0.3 us of host per node, 1.5 us from one kernel of
a dependent chain to the next.  A graph with parallel branches (streams forked inside the capture) is issued node by node by the
runtime: 3.2 us of host per node and 6-9 us per chain link — S3D-G's 1 800-node step cost 5-6 ms of host per replay and every one of
its BatchNorm chains (reduce -> finalize -> apply -> ...) paid the longer link.  So the step is cut into graphs that are each a
plain chain, and the concurrency lives BETWEEN graphs, on streams the stepper orders with events:

                                                                                        [DDP's buffer broadcast: BN running statistics]
    main lane   top (momentum update, _diff_speed, the three clip gathers)              [clip all-to-all x2: eager RCCL calls]
    main | q | k   key_kneg | query | key_k   three linear graphs replayed side by side on three streams
    main lane   keys_join (deferred BatchNorm statistics, feature stack)                [key all-gather]
    main lane   tail (un-shuffle, logits, losses, enqueue, the heads' backward)
    main | w    the encoder's backward in pieces (`_backward_piece`); each piece's small weight gradients replay as a graph of their
                own on the "w" lane beside the NEXT piece                               [gradient all-reduce, 32 MiB buckets, from "w"]
    main lane   update (DDP's 1/ws average, SGD)

The side lanes q / k / w are streams MEASURED to run beside the main stream and one another (rspnet_amd/streams.py): HIP maps a
process's streams onto four hardware queues in creation order, and two lanes that share a queue run one after the other whatever the
events say (round 6: the "w" lane sat on the main lane's queue and bought nothing until it was moved; profiles/r06/experiments_r6.txt).

The collective points of the data-parallel step fall between graphs, so the SAME schedule serves one rank and N > 1 (RCCL calls
cannot be captured on this stack: hipStreamEndCapture segfaults, profiles/r04/experiments_r4.txt); with one rank and no process
group the collective slots are empty.  Each lane's graphs share a memory pool (they replay in capture order), different lanes have
different pools.  What the lanes give up against the forked capture: S3D-G's sibling branches run one after the other inside a
lane (the three passes side by side fill the machine instead).  What the data-parallel step gives up against eager issue: the
second clip exchange no longer hides under the first key pass — a fraction of a millisecond over xGMI, the price
of not being Python-bound; the bucket all-reduces do start inside the backward, at piece boundaries.  mode "segments": four graphs between
the collective points with the forks inside (round 5's first version); "whole": one graph, forks inside (rounds 2-4; N = 1 only).
Any failure to capture falls back to the eager loop with a logged warning — the result is the same either way, kernel for kernel.

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
    HOST_BOUND_DP = 0.25  # ... with the data-parallel collectives on: N interpreters (and their RCCL proxy threads) share one host
    DEFAULT_MODE = "lanes"
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
        self.segment_host_ms = None               # {} to collect host time per segment of the segmented replay
        self.segment_gpu_events = None            # [] to collect (graph name, start event, end event) of every replayed graph
        # How the captured step is laid out (module docstring).  RSP_GRAPH_MODE = whole | segments | lanes overrides the default for
        # A/B runs; "whole" with the collectives on captures the RCCL calls too (RSP_GRAPH_COLLECTIVES=1 of round 4: exercised at one
        # rank, never the default); RSP_NO_SEGMENTS=1 issues the data-parallel step eagerly as rounds 1-4 did.
        import os
        coll = bool(self.model._dp()[2])
        self.mode = os.environ.get("RSP_GRAPH_MODE") or ("whole" if (os.environ.get("RSP_GRAPH_COLLECTIVES") and coll) else self.DEFAULT_MODE)
        if self.mode not in ("whole", "segments", "lanes"):
            raise ValueError(f"RSP_GRAPH_MODE={self.mode}: whole, segments or lanes")
        if coll and self.mode == "whole" and not os.environ.get("RSP_GRAPH_COLLECTIVES"):
            self.mode = "segments"                # (RCCL calls are not captured: hipStreamEndCapture segfaults on this stack)
        self.disabled = coll and bool(os.environ.get("RSP_NO_SEGMENTS")) and not os.environ.get("RSP_GRAPH_COLLECTIVES")
        self.fallback_reason = "data-parallel collectives on, RSP_NO_SEGMENTS (issued eagerly)" if self.disabled else None
        self.lane_streams = {}                    # lane name -> side stream ("main" = the caller's current stream)
        self.pools = {}                           # lane name -> memory pool of that lane's graphs

    # ---- the five statements ------------------------------------------------------------------------------------------------
    def _eager(self, im_q, im_k, host):
        out, tgt, rl, rt = self.model._device_part(im_q, im_k, host)
        loss, loss_A, loss_M = self.criterion(out, tgt, rl, rt)
        self.optimizer.zero_grad()
        loss.backward()
        self.optimizer.step()
        return loss, loss_A, loss_M, out, rl, tgt, rt

    def _segments(self, im_q, im_k):
        """The five statements cut at the collective points: [(kind, fn(host))] with kind 'graph' (device work: captured once,
        replayed) or 'eager' (a collective: issued every step with that step's split lists).  `box` carries the step's state
        between them — static buffers of the graphs' memory pool once captured.  (The "segments" schedule without its lane
        annotations: tests issue these eagerly, one after the other, on any device.)"""
        ops, box = self._schedule(im_q, im_k, "segments")
        return [("graph" if op[0] == "g" else "eager", op[-1]) for op in list(ops) if op[0] in ("g", "e")], box

    def _schedule(self, im_q, im_k, mode):
        """The step as a sequence of operations over `box`, the state they share (an iterable: the "lanes" schedule is a generator
        whose later operations depend on what capturing the earlier ones found — how many pieces the backward is cut into):
            ("g", lane, name, fn)   device work, captured once as ONE graph and replayed on the lane's stream
            ("e", lane, name, fn)   host-issued work on the lane's stream every step: a collective with that step's split lists
            ("fork", lane)          the lane's stream waits for the main stream
            ("join", lane)          the main stream waits for the lane's stream
        mode "whole": one graph, the three passes and the weight-gradient tasks forked INSIDE the capture; "segments": the same
        cut at the collective points; "lanes": every graph a LINEAR chain — the three forward passes as three graphs replayed side
        by side on streams of their own, the backward as a chain of pieces on the main lane with the weight gradients of each
        piece as a graph on the "w" lane beside the next piece."""
        m, box = self.model, {}

        def whole(host):
            loss, loss_A, loss_M, out, rl, _, _ = self._eager(im_q, im_k, host)
            box["outs"] = (loss, loss_A, loss_M, out, rl)

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

        if mode == "whole":
            return [("g", "main", "step", whole)], box
        def sync_buffers(host):
            m._broadcast_running_stats()         # DDP's per-forward buffer broadcast (a no-op without a process group)

        if mode == "segments":
            pre = [("e", "main", "broadcast_buffers", sync_buffers)] if (m._dp()[2] and m.broadcast_buffers) else []
            return pre + [("g", "main", "top", top), ("e", "main", "all_to_all", exchange), ("g", "main", "passes", passes),
                    ("e", "main", "all_gather", gather), ("g", "main", "tail", tail), ("e", "main", "all_reduce", reduce),
                    ("g", "main", "update", update)], box
        return self._lanes(box, top, gather, tail, update, sync_buffers), box

    # Plan nodes per piece of the backward (lanes mode) when weight gradients are set aside for the "w" lane.  History: round 5 measured
    # 8-node pieces everywhere at one rank — every extra graph costs ~50 us on the GPU side, R3D-18 1 259 -> 1 235, S3D-G 402 -> 398 — and
    # kept the backward in one graph; round 6 found why (the "w" lane shared the main lane's hardware queue), see `_backward_piece`.
    import os as _os
    BACKWARD_PIECE = int(_os.environ.get("RSP_BWD_PIECE", "-1"))      # -1: the default policy of `_backward_piece`
    # ... and in the last TAIL_NODES plan nodes of the chain (-1: one piece's worth) a cut as soon as the weight gradients set aside
    # since the last one add up to TAIL_CUT_FLOPS (0: off): the last piece's weight gradients run ALONE behind the end of the chain
    # (S3D-G: 1.4 ms of them, sep_conv2's 89 GFLOP among them), a piece of their own lets the big ones start beside the memory-bound
    # front-end layers instead.  profiles/r06/experiments_r6.txt r6s-w: S3D-G 421.8-424.4 -> 428.6-430.3 on one box, 425.4-426.8 ->
    # 426.6-428.8 on another (0 / 40 / 60 / 80 / 120 GFLOP: 60 best or equal); R3D-18 through the collectives 1 334-1 336 -> 1 341-1 342.
    TAIL_CUT_FLOPS = float(_os.environ.get("RSP_BWD_TAIL_CUT_GFLOP", "60")) * 1e9
    TAIL_NODES = int(_os.environ.get("RSP_BWD_TAIL_NODES", "-1"))
    # A bucket's all-reduce is issued on the "w" lane behind the weight gradients that complete it — but AFTER the next piece has been
    # handed to the main lane: issued at the boundary itself it held that piece up until the weight gradients were done (R3D-18, gap
    # wgrad0 -> backward1 +0.03 ms instead of -0.5 ms: the collective's wait sits in front of the piece in a queue they share)
    REDUCE_LATE = bool(int(_os.environ.get("RSP_REDUCE_LATE", "1")))

    def _backward_piece(self, coll: bool) -> int:
        """Plan nodes per backward piece of the "lanes" schedule; 0: the backward is not cut (beyond the gradient-bucket boundaries).
        Pieces of a ninth of the plan, at least 12 nodes (S3D-G: 15 of ~130, R3D-18: 12 of 28), each piece's small weight gradients
        replayed as a graph of their own on the "w" lane beside the next piece (and, with a process group, the gradient buckets leaving from that lane as they complete) — WHEN the "w" lane
        really runs beside the main lane.  Round 6 (profiles/r06/experiments_r6.txt r6b-j): HIP multiplexes streams onto four hardware
        queues, and a "w" stream that shares the main lane's queue replays its graphs BETWEEN the pieces: three extra graphs for
        nothing (S3D-G 420.2 uncut vs 419.0-420.9 cut, one rank).  With the lanes on queues of their own (rspnet_amd/streams.py
        measures it) the cut gives S3D-G 427.1 at 15 nodes, 421.7 at 44, 417.4 uncut on one box (+2.3 %; another box, eight hardware
        queues: 426.9 at 15, 425.5 at 25, 426.0 at 40, 422.2 at 60, 420.2 uncut); with the collectives on, where the
        process group's streams had happened to shift the "w" lane onto a free queue, S3D-G 412.4 -> 422.5 and R3D-18 1282.9 -> 1319.0.
        RSP_BWD_PIECE overrides (sweeps)."""
        if self.BACKWARD_PIECE >= 0:
            return self.BACKWARD_PIECE
        try:
            n = len(self.model.encoder_q.plan().nodes)
            dev = self.model.queue.device
            if dev.type != "cuda":
                return max(12, -(-n // 9)) if coll else 0         # (host-logic tests on the checker backend: the data-parallel cut)
            from . import streams as _streams
            if not _streams.lanes_overlap(dev).get("w", False):
                return 0                                         # the weight-gradient lane would only wait its turn
        except Exception:      # noqa: BLE001 - an encoder without a layer plan: no cut
            return 0
        return max(12, -(-n // 9))

    def _lanes(self, box, top, gather, tail, update, sync_buffers):
        """The "lanes" schedule (see `_schedule`), a generator consumed by `_capture`."""
        from .engine import BranchStreams
        from .moco.builder_diffspeed_diffloss import BUCKET_FLOATS
        m = self.model

        def exchange(host):
            # both clip all-to-alls start here; the main stream waits for the k_negative clips only, the k clips are awaited on the
            # k lane: the second exchange runs over xGMI under the first key pass, as in the eager data-parallel step
            m._phase_exchange(box["st"], host, wait="first")
            m._last_draw = (m._last_draw[0], host["speed"]) + tuple(host["sh"])

        coll = bool(m._dp()[2])
        piece_nodes = self._piece_agreed if getattr(self, "_piece_agreed", None) is not None else self._backward_piece(coll)
        pieces = piece_nodes > 0
        if not coll and not pieces:
            # One rank, no process group: no collective points, so the main lane needs only three graphs (every graph boundary
            # costs ~50 us on the GPU side, profiles/r05/experiments_r5.txt): top | key_kneg | everything behind the joins
            def rest(host):
                m._passes_join(box["st"])
                m._defer_backward = False
                tail(host)
                update(host)

            yield ("g", "main", "top", top)
            yield ("fork", "q")
            yield ("g", "q", "query", lambda host: m._pass_query(box["st"]))
            yield ("fork", "k")
            yield ("g", "k", "key_k", lambda host: m._pass_key(box["st"], 1))
            yield ("g", "main", "key_kneg", lambda host: m._pass_key(box["st"], 0))
            yield ("join", "k")
            yield ("join", "q")
            yield ("g", "main", "tail+update", rest)
            return
        if coll and m.broadcast_buffers:
            yield ("e", "main", "broadcast_buffers", sync_buffers)      # (in front of the query pass's fork: see `top` below)
        yield ("g", "main", "top", top)
        yield ("fork", "q")
        yield ("g", "q", "query", lambda host: m._pass_query(box["st"]))
        if coll:
            yield ("e", "main", "all_to_all", exchange)
        yield ("fork", "k")
        if coll:
            yield ("e", "k", "all_to_all_k", lambda host: m._wait_exchange(box["st"], 1))
        yield ("g", "k", "key_k", lambda host: m._pass_key(box["st"], 1))
        yield ("g", "main", "key_kneg", lambda host: m._pass_key(box["st"], 0))
        yield ("join", "k")
        if coll:
            yield ("g", "main", "keys_join", lambda host: m._passes_join(box["st"]))
            yield ("e", "main", "all_gather", gather)
            yield ("join", "q")
            yield ("g", "main", "tail", tail)      # ... up to the gradient of the query features (`_defer_backward`)
        else:
            # one rank with the backward in pieces (BACKWARD_PIECE > 0): the joins and the tail share a graph
            def joined_tail(host):
                m._passes_join(box["st"])
                tail(host)

            yield ("join", "q")
            yield ("g", "main", "tail", joined_tail)
        # the encoder's backward: pieces of the node chain on the main lane; the weight gradients a piece set aside
        # (engine.BranchStreams.deferred) as one graph on the "w" lane, beside the next piece; a gradient bucket whose last
        # parameter has been issued is all-reduced from the "w" lane's stream — behind everything that writes into it
        # (buckets of 64 MiB here: each one ends a graph, and an extra graph costs about what 10 MB of all-reduce do)
        buckets = m._flat.buckets(2 * BUCKET_FLOATS) if coll else []
        bw = {"it": None, "done": False, "tasks": [], "keep": [], "handed": set(), "launched": set(), "handles": [], "seen": 0}
        box["backward"] = bw
        n_nodes = len(m.encoder_q.plan().nodes) if pieces else 0
        tail_nodes = self.TAIL_NODES if self.TAIL_NODES >= 0 else piece_nodes

        def piece(host):
            if bw["it"] is None:
                bw["it"] = m._backward_iter()
            BranchStreams.deferred = bw["tasks"] if pieces else None
            try:
                n = 0
                for handed in bw["it"]:
                    bw["handed"] = handed
                    n += 1
                    done_now = [all(pid in handed for pid in ids) for _, _, ids in buckets]
                    # (a cut where a bucket has just completed — unless every bucket has: what is left then is the end of the
                    #  chain, and the flush behind the loop takes the rest)
                    ready = not all(done_now) and any(d and bi not in bw["launched"] for bi, d in enumerate(done_now))
                    bw["seen"] += 1
                    # The weight gradients of the LAST piece have nothing left to run beside: a big one set aside in the final stretch
                    # of the chain (S3D-G's sep_conv2: 89 GFLOP, 0.9 ms) starts a piece of its own at once, beside the memory-bound
                    # BatchNorm / pool / gate passes of the front-end layers that follow, instead of alone behind the end of the chain
                    tail_cut = (pieces and self.TAIL_CUT_FLOPS > 0 and bw["tasks"] and n_nodes - bw["seen"] <= tail_nodes
                                and n_nodes - bw["seen"] > 0 and sum(t[2] for t in bw["tasks"]) >= self.TAIL_CUT_FLOPS)
                    if ready or tail_cut or (pieces and n >= piece_nodes and bw["tasks"]):
                        return
                bw["done"] = True
            finally:
                BranchStreams.deferred = None

        def wgrads(host):
            for fn, keep, _ in bw["tasks"]:
                fn()
                # the operands stay alive until the whole backward has been captured: the main lane's later pieces must not be
                # handed their memory while this graph may still be reading it
                bw["keep"].append(keep)
            del bw["tasks"][:]

        def reducer(s, e):
            def fn(host):
                bw["handles"].append(dist.all_reduce(m._flat.g_flat[s:e], async_op=True))
            return fn

        def reduce_wait(host):
            hs, bw["handles"] = bw["handles"], []
            if hs:
                with m._comm("allreduce_wait"):
                    for h in hs:
                        h.wait()

        j = 0
        used_w = False
        held = []          # all-reduces whose issue is held back until the NEXT piece is in the main lane's queue (REDUCE_LATE)
        while not bw["done"]:
            yield ("g", "main", f"backward{j}", piece)
            for op in held:
                yield op
            del held[:]
            forked = False
            if bw["tasks"]:
                yield ("fork", "w")
                forked = used_w = True
                yield ("g", "w", f"wgrad{j}", wgrads)
            for bi, (s, e, ids) in enumerate(buckets):
                if bi not in bw["launched"] and all(pid in bw["handed"] for pid in ids):
                    bw["launched"].add(bi)
                    if not forked:
                        yield ("fork", "w")
                        forked = used_w = True
                    op = ("e", "w", f"all_reduce{bi}", reducer(s, e))
                    if self.REDUCE_LATE:
                        held.append(op)
                    else:
                        yield op
            j += 1
        for op in held:
            yield op
        for bi, (s, e, ids) in enumerate(buckets):
            if bi not in bw["launched"]:
                bw["launched"].add(bi)
                yield ("fork", "w")
                used_w = True
                yield ("e", "w", f"all_reduce{bi}", reducer(s, e))
        if used_w:
            yield ("join", "w")
        if coll:
            yield ("e", "main", "all_reduce_wait", reduce_wait)
        yield ("g", "main", "update", update)
        del bw["keep"][:]
        bw["it"] = None

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
            from . import ops as _ops
            if getattr(_ops.backend(), "event_log", None) is not None:
                # per-launch timing events are being recorded (bench.py's roofline pass): nothing can be captured now, and nothing
                # is decided — no copy into the static clip buffers, no cross-rank agreement; the step is simply issued eagerly
                return self._eager(im_q, im_k, host)[:5]
            self._static_clips(im_q, im_k)
            entry = self._capture(key, host)
            if entry is None:
                return self._eager(im_q, im_k, host)[:5]
        else:
            self._static_clips(im_q, im_k)
        self.graphs[key] = self.graphs.pop(key)         # most recently used last
        self._replay(entry[3], host)
        return entry[1]

    def _replay(self, seq, host):
        """Run a captured schedule (see `_schedule`): graphs replayed on their lanes' streams, collectives issued in between."""
        dev = self.static["dev"].device
        m = self.model
        if m._last_draw is not None:
            # introspection state follows THIS step's host draws in every mode (the device-side permutation is the graph's own
            # buffer, rewritten by the replay)
            m._last_speed = host["speed"]
            m._last_draw = (m._last_draw[0], host["speed"]) + tuple(host["sh"])
        prof = self.segment_host_ms
        gev = self.segment_gpu_events
        if prof is not None:
            import time

        def replay(op):
            # (measurement: where the GPU's time of a replayed step goes — an event pair on the graph's own stream around it)
            if gev is not None:
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                op[3].replay()
                e1.record()
                gev.append((op[1] + ":" + op[2], e0, e1))
            else:
                op[3].replay()

        for op in seq:
            t0 = time.perf_counter() if prof is not None else 0.0
            kind = op[0]
            if kind == "g":
                if op[1] == "main":
                    replay(op)
                else:
                    with torch.cuda.stream(self.lane_streams[op[1]]):
                        replay(op)
            elif kind == "e":
                if op[1] == "main":
                    op[3](host)
                else:
                    with torch.cuda.stream(self.lane_streams[op[1]]):
                        op[3](host)
            elif kind == "fork":
                self.lane_streams[op[1]].wait_stream(torch.cuda.current_stream(dev))
            else:
                torch.cuda.current_stream(dev).wait_stream(self.lane_streams[op[1]])
            if prof is not None and kind in ("g", "e"):
                # (measurement: where the host's time of a replayed step goes — set `segment_host_ms = {}` to collect)
                prof.setdefault(("graph:" if kind == "g" else "rccl:") + op[2], []).append((time.perf_counter() - t0) * 1e3)

    def clip_buffers(self, like_q, like_k):
        """The static clip tensors the captured graphs read (allocated on first use, shaped like the arguments).  A data path that
        writes its batches straight into them (a GPU augmentation with `out=`, synthetic clips) and passes them to `__call__` skips
        the per-step copy a replayed graph otherwise needs (2 x 154 MB for C3D's 32 clips: 0.5 % of an R3D-18 step)."""
        if getattr(self, "_clip_bufs", None) is None:
            self._clip_bufs = (torch.empty_like(like_q), torch.empty_like(like_k))
        return self._clip_bufs

    def _static_clips(self, im_q, im_k):
        st = self.static
        if st["im_q"] is None:
            st["im_q"], st["im_k"] = self.clip_buffers(im_q, im_k)
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
            bound = self.HOST_BOUND_DP if self.model._dp()[2] else self.HOST_BOUND
            share = self._agree(share, max)      # every rank must reach the same decision: the two issue modes bucket the
            #   gradient all-reduce differently, and ranks that disagree would wait for each other's collectives forever
            if share <= bound:
                self.eager_keys[cfg] = (f"issued eagerly by policy: the host issues this step in {h_med:.1f} ms of the {g_med:.1f} ms it runs "
                                        f"(median of {len(got)} measured warm-up steps; a replayed graph pays off above {bound:.0%})")
                self._cap(self.eager_keys, self.MAX_KEYS)
                self.fallback_reason = self.eager_keys[cfg]
                log.info("rspnet_amd: pretext step (speed %s) %s", cfg[0], self.eager_keys[cfg])
        return out

    def _agree(self, value: float, op):
        """The same value on every rank of the data-parallel job (op = max or min over the ranks); the value itself otherwise."""
        if not self.model._dp()[2]:
            return value
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.model.queue.device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX if op is max else dist.ReduceOp.MIN)
        return float(t.item())

    def _capture(self, key, host):
        # Where the backward is cut decides WHEN each gradient bucket's all-reduce is issued, and buckets that complete inside one piece
        # leave in index order: ranks that cut differently (the cut follows a local MEASUREMENT of the "w" lane, `_backward_piece`)
        # would issue the same buckets in different orders and pair up the wrong collectives.  The smallest answer wins everywhere.
        self._piece_agreed = None
        if self.model._dp()[2] and self.mode == "lanes":
            self._piece_agreed = int(self._agree(float(self._backward_piece(True)), min))
        entry = self._capture_local(key, host)
        # a capture that failed on ONE rank (that rank would issue the eager step, with other collectives) is a failure everywhere
        if self.model._dp()[2] and self._agree(0.0 if entry is None else 1.0, min) < 0.5 and entry is not None:
            self.graphs.pop(key, None)
            self.disabled, self.fallback_reason = True, "HIP-graph capture failed on another rank; all ranks issue eagerly"
            log.warning("rspnet_amd: %s", self.fallback_reason)
            return None
        return entry

    def _capture_local(self, key, host):
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
            dev = st["dev"].device
            # With a process group alive, its watchdog thread polls the events of finished collectives at any time; under the default
            # "global" capture mode such a query from ANOTHER thread while this one captures is an error that takes the process down
            # (hipErrorStreamCaptureUnsupported raised inside the watchdog: seen on the first --graph on run of round 5).  "thread_local"
            # restricts only the capturing thread.
            cmode = "thread_local" if (dist.is_available() and dist.is_initialized()) else "global"
            # Graphs are captured in step order, each into the memory pool of its LANE: graphs of one lane replay in capture order
            # and never side by side, so they share a pool (a later graph re-uses what an earlier one freed); graphs of different
            # lanes run concurrently and must not.  A tensor that crosses lanes (the query clips, the kept activations) lives in its
            # producer's pool until its last consumer's capture drops it.  The collectives between the graphs are not executed now —
            # nothing is, a capture only records — the replay below runs the step this capture stands for.
            ops, box = self._schedule(st["im_q"], st["im_k"], self.mode)
            seq = []
            m._defer_reduce = self.mode != "whole"
            m._defer_backward = self.mode == "lanes"
            try:
                for op in ops:
                    lane = op[1]
                    if lane != "main" and lane not in self.lane_streams:
                        # (the process's three measured-to-overlap side streams: rspnet_amd/streams.py; assigned before the first
                        #  capture of this call begins — `lane` never measures inside one)
                        from . import streams as _streams
                        self.lane_streams[lane] = _streams.lane(dev, lane) if lane in ("q", "k", "w") else torch.cuda.Stream(device=dev)
                    if op[0] != "g":
                        seq.append(op)
                        continue
                    _, lane, name, fn = op
                    if lane not in self.pools:
                        self.pools[lane] = torch.cuda.graph_pool_handle()
                    sg = torch.cuda.CUDAGraph()
                    try:
                        with torch.cuda.graph(sg, pool=self.pools[lane], capture_error_mode=cmode):
                            # forks INSIDE a capture (query / key passes, weight-gradient tasks: engine.BranchStreams) only where the
                            # schedule wants them: a graph with parallel branches is issued node by node by the runtime (3.2 us of
                            # host per node, 6-9 us per chain link; a linear graph: 0.3 us and 1.5 us, tools/chain_gap_probe.py)
                            # (origin None: engine.BranchStreams keeps every node on the capturing stream)
                            BranchStreams.origin = torch.cuda.current_stream(dev).cuda_stream if self.mode != "lanes" else None
                            fn(host)
                    finally:
                        BranchStreams.origin = None
                    seq.append(("g", lane, name, sg))
            finally:
                m._defer_reduce = m._defer_backward = False
                m._pending_bwd = None
            outs = box["outs"]
            g = None
            # Everything the graph's kernels address that was allocated OUTSIDE the capture must outlive the graph: a later
            # configuration may rebuild a packed-weight set, and the superseded buffers — still baked into this graph's kernel
            # arguments — would be freed.  (The library's scratch buffers come from the graph's own pool: ops.HipOps._workspace.)
            # (also baked into captured kernel arguments: the deferred BatchNorm-update set of encoder_k — its moment buffer and job
            #  table — and the virtual-pixel stems' derived filters and index vectors; both are rebuilt when pointers / devices change)
            keep = [list(m.encoder_q._packed._sets), list(m.encoder_k._packed._sets), m._flat,
                    getattr(m._flat, "m_flat", None), getattr(m, "_nbt_q", None), getattr(m, "_nbt_k", None), st,
                    m._ema_k, m._ema_map, list(m.encoder_q._packed._virtual.values()), list(m.encoder_k._packed._virtual.values())]
            keep.append(box)                             # the graphs' shared state: static buffers of the pools
            self.graphs[key] = (g, outs, keep, seq)
            log.info("rspnet_amd: pretext step captured as %d HIP graph(s), mode %s (speed %s, clips %s)",
                     sum(1 for o in seq if o[0] == "g"), self.mode, key[0], key[1])
            # the capture itself executed nothing: run the step it stands for
            return self.graphs[key]
        except Exception as e:      # noqa: BLE001 - whatever refuses the capture, the eager loop still works
            self.disabled, self.fallback_reason = True, f"{type(e).__name__}: {e}"
            log.warning("rspnet_amd: HIP-graph capture of the pretext step failed (%s); running eagerly", self.fallback_reason)
            torch.cuda.synchronize()
            return None
