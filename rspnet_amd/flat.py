"""Flat parameter / gradient storage for the (query, key) encoder pair.

Every parameter of ``encoder_q`` is a view into one contiguous fp32 buffer, ``encoder_k``'s into a second one
with the identical layout, so that the momentum update (builder_diffspeed_diffloss.py:337-343), the gradient
all-reduce (DDP, moco/__init__.py:49-53) and SGD (pretrain.py:65-72) are single streaming kernels / few large
collectives instead of one launch per tensor.  Parameters that never receive a gradient in the pretext step —
the backbone's own classifier (``encoder.linear`` / ``encoder.fc``, built with num_classes=1,
split_wrapper.py:99) — sit after the trainable range: the momentum update covers them (as the reference does),
gradients / SGD do not (torch.optim.SGD skips tensors whose .grad is None).
"""
from __future__ import annotations

from typing import Dict, List, Tuple

import torch
from torch import nn

ALIGN = 4  # floats: every view starts on a 16-byte boundary


class FlatEncoderPair:
    def __init__(self, enc_q: nn.Module, enc_k: nn.Module, untrained_prefixes: Tuple[str, ...],
                 adjacent: Tuple[Tuple[str, ...], ...] = ()):
        """adjacent: groups of parameter names that must sit back to back (in the given order, no padding between them) so
        that the group is ONE contiguous tensor — the filters of convolutions that run as a single GEMM (S3D-G's inception
        siblings).  Every member's numel must be a multiple of ALIGN."""
        self.enc_q, self.enc_k = enc_q, enc_k
        names = [n for n, _ in enc_q.named_parameters()]
        follower = {}
        for grp in adjacent:
            for a, b in zip(grp[:-1], grp[1:]):
                follower[a] = b
        placed_later = set(follower.values())
        ordered = []
        for n in names:
            if n in placed_later:
                continue
            ordered.append(n)
            while ordered[-1] in follower:
                ordered.append(follower[ordered[-1]])
        assert sorted(ordered) == sorted(names)
        names = ordered
        self.adjacent = tuple(tuple(g) for g in adjacent)
        trained = [n for n in names if not n.startswith(untrained_prefixes)]
        untrained = [n for n in names if n.startswith(untrained_prefixes)]
        self.names: List[str] = trained + untrained
        self.n_trained_params = len(trained)
        self.offsets: Dict[str, Tuple[int, int]] = {}
        off = 0
        pq = dict(enc_q.named_parameters())
        for i, n in enumerate(self.names):
            if i == self.n_trained_params:
                self.train_end = off
            numel = pq[n].numel()
            self.offsets[n] = (off, numel)
            off += (numel + ALIGN - 1) // ALIGN * ALIGN
        if self.n_trained_params == len(self.names):
            self.train_end = off
        self.total = off
        self.q_flat = self.k_flat = self.g_flat = None
        self._pq = pq
        self._pk = dict(enc_k.named_parameters())
        self.grad_views: Dict[int, torch.Tensor] = {}

    def _views_ok(self) -> bool:
        if self.q_flat is None:
            return False
        base_q, base_k = self.q_flat.data_ptr(), self.k_flat.data_ptr()
        for n, (off, _) in self.offsets.items():
            if self._pq[n].data_ptr() != base_q + 4 * off or self._pk[n].data_ptr() != base_k + 4 * off:
                return False
        return True

    def ensure(self):
        """(Re)build the flat buffers if the parameters were moved (.cuda()/.to()) since last time."""
        if self._views_ok():
            return
        dev = next(iter(self._pq.values())).device
        q = torch.zeros(self.total, dtype=torch.float32, device=dev)
        k = torch.zeros(self.total, dtype=torch.float32, device=dev)
        for n, (off, numel) in self.offsets.items():
            for flat, p in ((q, self._pq[n]), (k, self._pk[n])):
                v = flat[off:off + numel].view(p.shape)
                v.copy_(p.data)
                p.data = v
        self.q_flat, self.k_flat = q, k
        self.g_flat = torch.zeros(self.train_end, dtype=torch.float32, device=dev)
        self.grad_views = {}
        for n in self.names[:self.n_trained_params]:
            off, numel = self.offsets[n]
            p = self._pq[n]
            self.grad_views[id(p)] = self.g_flat[off:off + numel].view(p.shape)
            p._rsp_flat = (self, off, numel)

    def grad_of(self, p: nn.Parameter):
        return self.grad_views.get(id(p))

    def group_view(self, which: str, names, shape):
        """One tensor over the adjacent parameters `names`: which = 'q' / 'k' weights or 'g' gradients."""
        off0, _ = self.offsets[names[0]]
        total, off = 0, off0
        for n in names:
            o, numel = self.offsets[n]
            assert o == off and numel % ALIGN == 0, f"{n} is not adjacent to its group (flat layout)"
            off += numel
            total += numel
        flat = {"q": self.q_flat, "k": self.k_flat, "g": self.g_flat}[which]
        return flat[off0:off0 + total].view(shape)

    def attach_grads(self):
        """Make .grad of every trained parameter the view into the flat gradient buffer."""
        for n in self.names[:self.n_trained_params]:
            p = self._pq[n]
            v = self.grad_views[id(p)]
            if p.grad is None or p.grad.data_ptr() != v.data_ptr():
                p.grad = v

    def buckets(self, bucket_floats: int) -> List[Tuple[int, int, List[int]]]:
        """Contiguous [start, end) ranges of g_flat on parameter boundaries with the ids of their parameters."""
        out, start, ids = [], 0, []
        for n in self.names[:self.n_trained_params]:
            off, numel = self.offsets[n]
            end = off + (numel + ALIGN - 1) // ALIGN * ALIGN
            ids.append(id(self._pq[n]))
            if end - start >= bucket_floats:
                out.append((start, end, ids))
                start, ids = end, []
        if ids:
            out.append((start, self.train_end, ids))
        return out
