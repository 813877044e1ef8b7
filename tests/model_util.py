"""Run rspnet_amd's pretext model on a golden case (any device / op backend) and return comparable results."""
import random

import numpy as np
import torch

from rspnet_amd.moco import Loss, ModelFactory


def make_cfg(arch, K, fc_type="linear", dim=128, m=0.999, T=0.07, speeds=(2,)):
    return {"model": {"arch": arch}, "moco": {"dim": dim, "k": K, "m": m, "t": T, "fc_type": fc_type,
                                               "diff_speed": list(speeds)}}


class ReplayRNG:
    """Replays torch.randperm / random.choice (same trick as oracle/ref_harness.py).  perms = [the _diff_speed permutation of
    range(B) — the model draws it on the DEVICE generator (builder_diffspeed_diffloss.py:423) —, shuffle #1, shuffle #2 — drawn on
    the host generator (:372), in call order].  The product draws its host-side decisions before the device part of the step
    (MoCoDiffLossTwoFc._host_part), so the two kinds are told apart by the `device=` argument, not by position."""

    def __init__(self, perms, speed):
        self.perms = [torch.from_numpy(np.asarray(p, dtype=np.int64)) for p in perms]
        self.speed = speed
        self.n = 1

    def __enter__(self):
        self._rp, self._ch = torch.randperm, random.choice

        def randperm(n, *a, **k):
            dev = k.get("device")
            if dev is not None:
                p = self.perms[0]
            else:
                p = self.perms[self.n]
                self.n += 1
            assert p.numel() == n, (p.numel(), n)
            out = p.clone()
            return out.to(dev) if dev is not None else out

        torch.randperm = randperm
        random.choice = lambda seq: self.speed
        return self

    def __exit__(self, *exc):
        torch.randperm, random.choice = self._rp, self._ch


def run_model_step(arch, meta, inputs, rank, device, optimizer="fused", issue="eager"):
    """One teacher-forced step of rspnet_amd on `device`.  Returns (out dict, post state dict, momentum_post dict).
    issue="segments" / "lanes": the step as the operation list rspnet_amd/graph_step.py captures and replays (GraphedPretextStep.
    _schedule: graphs and collectives alternately; "lanes": the three forward passes as separate graphs, the backward in pieces
    with the weight gradients set aside and the gradient buckets all-reduced between pieces) — here every operation is issued
    eagerly, in order, so the cut itself is what is tested (any device, any op backend)."""
    from rspnet_amd.optim import SGD
    state, mom, clips, perms_B, sh = inputs
    wrapped = ModelFactory(make_cfg(meta.get("arch", arch), meta["K"], fc_type=meta.get("fc_type", "linear"), m=meta["m"],
                                    T=meta["T"])).build_moco_diffloss(device=device)
    model = wrapped.module
    model.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in state.items()})
    model.train()
    params = list(wrapped.parameters())
    params = [p for p in params if p.requires_grad]
    cls = SGD if optimizer == "fused" else torch.optim.SGD
    opt = cls(params, lr=meta["lr"], momentum=meta["sgd_momentum"], dampening=0.0,
              weight_decay=meta["weight_decay"], nesterov=False)
    names = {id(p): n for n, p in model.named_parameters()}
    for p in params:
        n = names[id(p)]
        if n in mom:
            opt.state[p]["momentum_buffer"] = torch.from_numpy(mom[n].copy()).to(device)
    crit = Loss(margin=meta["margin"], A=meta["A"], M=meta["M"])
    im_q = torch.from_numpy(clips[rank][0]).to(device)
    im_k = torch.from_numpy(clips[rank][1]).to(device)
    if issue in ("segments", "lanes"):
        from rspnet_amd.graph_step import GraphedPretextStep
        stepper = GraphedPretextStep(wrapped, crit, opt)
        ops, box = stepper._schedule(im_q, im_k, issue)
        model._defer_reduce, model._defer_backward = True, issue == "lanes"
        names_run = []
        try:
            with ReplayRNG([perms_B[rank], sh[0], sh[1]], meta["speed"]):
                host = model._host_part(im_q.shape[0], device)
                for op in ops:                         # (a generator in "lanes" mode: later operations depend on the earlier ones)
                    if op[0] not in ("g", "e"):
                        continue                       # fork / join: stream ordering only
                    names_run.append(op[2])
                    if op[2] == "update":
                        # the last graph = DDP's average + SGD; the gradients are read between the two
                        model._scale_gradients()
                        grads = {names[id(p)]: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for p in params}
                        opt.step()
                    else:
                        op[3](host)
        finally:
            model._defer_reduce = model._defer_backward = False
        # (with a process group DDP's per-forward buffer broadcast comes first: an eager collective in front of the first graph)
        graphs_run = [n for n in names_run if n != "broadcast_buffers"]
        assert graphs_run[0] == "top" and names_run[-1] == "update" and "tail" in names_run, names_run
        assert (names_run[0] == "broadcast_buffers") == bool(model._dp()[2] and model.broadcast_buffers), names_run
        if issue == "lanes":
            assert {"query", "key_k", "key_kneg", "keys_join", "backward0"} <= set(names_run), names_run
        run_model_step.last_ops = names_run
        loss, loss_A, loss_M, out, rl = box["outs"]
        tgt, rt = torch.zeros(im_q.shape[0], dtype=torch.long), torch.ones(im_q.shape[0], dtype=torch.long)
    else:
        with ReplayRNG([perms_B[rank], sh[0], sh[1]], meta["speed"]):
            out, tgt, rl, rt = wrapped(im_q, im_k)
        loss, loss_A, loss_M = crit(out, tgt, rl, rt)
        opt.zero_grad()
        loss.backward()
        grads = {names[id(p)]: (None if p.grad is None else p.grad.detach().cpu().numpy().copy()) for p in params}
        opt.step()
    if device.type == "cuda":
        torch.cuda.synchronize()
    q_A, q_M = model._last_q
    res = {"loss": loss, "loss_A": loss_A, "loss_M": loss_M, "logits1": out[0], "logits2": out[1], "l_pos_M": rl[0],
           "l_neg_M": rl[1], "q_A": q_A, "q_M": q_M}
    # encoder_k outputs of this rank in the reference's shuffled order (pass #1 = k_negative clips, pass #2 = k clips)
    dim = q_A.shape[1]                       # width of the A head (the M head is 1-d for fc_type 'speednet')
    for tag, (feats, order) in zip(("kneg", "k"), model._last_k):
        shuf = torch.empty_like(feats)
        shuf[torch.from_numpy(order).to(feats.device)] = feats
        res[f"{tag}_A_shuf"], res[f"{tag}_M_shuf"] = shuf[:, :dim], shuf[:, dim:]
    res = {k: v.detach().cpu().numpy() for k, v in res.items()}
    assert tgt.dtype == torch.long and int(tgt.abs().sum()) == 0
    assert rt.dtype == torch.long and int((rt - 1).abs().sum()) == 0
    post = {k: v.detach().cpu().numpy() for k, v in model.state_dict().items()}
    mom_post = {names[id(p)]: opt.state[p]["momentum_buffer"].detach().cpu().numpy()
                for p in params if "momentum_buffer" in opt.state[p]}
    return res, post, mom_post, grads
