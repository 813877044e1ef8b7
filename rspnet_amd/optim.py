"""Fused SGD for the pretext step: torch.optim.SGD semantics (pretrain.py:65-72,165) in one streaming HIP kernel.

Subclasses torch.optim.SGD so param_groups / state_dict / load_state_dict (the checkpoint's 'optimizer' entry,
pretrain.py:250-257) keep torch's exact format: momentum buffers stay addressable as
``state[p]['momentum_buffer']``, they just alias one flat buffer.  Parameters that are not part of a flat encoder
buffer, or option combinations the kernel does not cover (nesterov, dampening), take torch's own step.
"""
from __future__ import annotations

import torch

from . import ops as _ops


class SGD(torch.optim.SGD):
    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        leftovers = False
        for group in self.param_groups:
            owners = {}
            for p in group["params"]:
                info = getattr(p, "_rsp_flat", None)
                if info is not None:
                    owners.setdefault(id(info[0]), (info[0], []))[1].append(p)
                elif p.grad is not None:
                    leftovers = True
            fusable = (not group["nesterov"] and group["dampening"] == 0 and not group.get("maximize", False))
            for flat, plist in owners.values():
                if not fusable or not self._fused_group(group, flat, plist):
                    leftovers = True
                    for p in plist:
                        p._rsp_skip = False
                else:
                    for p in plist:
                        p._rsp_skip = True
        if leftovers:
            self._torch_step_for_unfused()
        return loss

    def _fused_group(self, group, flat, plist) -> bool:
        trained = [flat._pq[n] for n in flat.names[:flat.n_trained_params]]
        if {id(p) for p in trained} != {id(p) for p in plist if p.grad is not None}:
            return False
        for p in trained:
            if p.grad is None or p.grad.data_ptr() != flat.grad_views[id(p)].data_ptr():
                return False
        n = flat.train_end
        mom = getattr(flat, "m_flat", None)
        have = ["momentum_buffer" in self.state[p] and self.state[p]["momentum_buffer"] is not None for p in trained]
        mu = group["momentum"]
        if mu != 0 and any(have) and not all(have):
            return False
        first = not any(have)
        if mom is None or mom.device != flat.q_flat.device:
            mom = torch.zeros(n, dtype=torch.float32, device=flat.q_flat.device)
            flat.m_flat = mom
            relink = True
        else:
            relink = False
        for p in trained:
            off, numel = p._rsp_flat[1:]
            view = mom[off:off + numel].view(p.shape)
            st = self.state[p]
            buf = st.get("momentum_buffer")
            if buf is not None and buf.data_ptr() != view.data_ptr():
                view.copy_(buf)          # e.g. after optimizer.load_state_dict (resume)
                st["momentum_buffer"] = view
            elif buf is None or relink:
                st["momentum_buffer"] = view
        _ops.backend().sgd_step(flat.q_flat[:n], flat.g_flat[:n], mom, float(group["lr"]), float(mu),
                                float(group["weight_decay"]), 1.0, first or mu == 0)
        # the weights changed: packed copies are stale
        flat.enc_q.weights_changed()
        return True

    def _torch_step_for_unfused(self):
        saved = []
        for group in self.param_groups:
            for p in group["params"]:
                if getattr(p, "_rsp_skip", False) and p.grad is not None:
                    saved.append((p, p.grad))
                    p.grad = None
        try:
            super().step()
        finally:
            for p, g in saved:
                p.grad = g
        # torch updated (some of) the flat-buffer parameters in place: their packed forward copies are stale
        seen = set()
        for group in self.param_groups:
            for p in group["params"]:
                info = getattr(p, "_rsp_flat", None)
                if info is not None and id(info[0]) not in seen:
                    seen.add(id(info[0]))
                    info[0].enc_q.weights_changed()
