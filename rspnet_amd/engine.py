"""Plan executor: runs a backbone's layer plan forward/backward through the HIP ops.

A backbone (rspnet_amd/models/*) is an ``nn.Module`` tree that only *holds* parameters under the reference's
state-dict names; its ``plan()`` lists fused units over numbered tensor slots:

  ConvBN   conv3d → BatchNorm3d(train) [→ + residual] [→ ReLU] [→ disjoint MaxPool3d]     (all four backbones)
  Pool     stand-alone MaxPool3d with overlapping windows                                  (ResNet stem, S3D-G)
  Gate     S3D-G self-gating  x * sigmoid(W·mean(x) + b)
  Concat   channel concat of branch outputs (S3D-G inception)

Forward keeps, per ConvBN, only the conv input, the raw conv output y and the per-channel (mean, invstd, scale,
shift); the activation / ReLU mask / pool arg-max are recomputed from y in backward.
Activations are (N, D, H, W, C) tensors.
"""
from __future__ import annotations

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


class PackedWeights:
    """Forward-packed conv weights of one encoder, rebuilt when the owner says the weights changed."""

    def __init__(self):
        self._cache: Dict[int, torch.Tensor] = {}

    def invalidate(self):
        self._cache.clear()

    def get(self, node: ConvBN, cg: ConvGeom):
        key = id(node.conv)
        w = self._cache.get(key)
        if w is None:
            w = _ops.backend().conv_pack_fwd(cg, node.conv.weight.data)
            self._cache[key] = w
        return w


def run_forward(plan: Plan, x: torch.Tensor, packed: PackedWeights, keep: bool) -> Tuple[torch.Tensor, Optional[ForwardCtx]]:
    """Execute `plan` on x (N,D,H,W,C).  keep=True records what backward needs.  BN is always in train mode
    (the pretext step never runs eval-mode BN: pretrain.py:225)."""
    be = _ops.backend()
    slots: Dict[int, torch.Tensor] = {plan.input_slot: x}
    ctx = ForwardCtx() if keep else None
    for ni, node in enumerate(plan.nodes):
        if isinstance(node, ConvBN):
            xin = slots[node.src]
            N, D, H, W, Cin = xin.shape
            w = node.conv.weight
            cg = ConvGeom(N, D, H, W, Cin, w.shape[0], node.k, node.s, node.p)
            bias = getattr(node.conv, "bias", None)
            y, stats = be.conv_fwd(cg, xin, packed.get(node, cg), None if bias is None else bias.data, True)
            bn = node.bn
            mi, ss = be.bn_finalize(stats, cg.rows, None if bias is None else bias.data, bn.weight.data, bn.bias.data,
                                    float(bn.eps), float(bn.momentum), bn.running_mean, bn.running_var)
            do, ho, wo = cg.out_dims
            pk, ps = node.pool if node.pool else ((1, 1, 1), (1, 1, 1))
            pg = PoolGeom(N, do, ho, wo, cg.Cout, pk, ps, (0, 0, 0))
            res = slots[node.residual] if node.residual is not None else None
            slots[node.dst] = be.bn_act_pool_fwd(pg, y, ss, res, node.relu)
            if keep:
                ctx.saved[ni] = _Saved(xin, y, mi, ss, cg, pg, res)
        else:
            raise NotImplementedError(f"plan node {type(node).__name__}")
    out = slots[plan.output_slot]
    if keep:
        ctx.feat_shape = tuple(out.shape)
    return out, ctx


def run_backward(plan: Plan, ctx: ForwardCtx, dfeat: torch.Tensor, grad_of, after_param_grads=None):
    """Backward through the plan.  `grad_of(param)` returns the (pre-allocated, flat-buffer) gradient view to
    fill for a parameter, or None to skip it.  `after_param_grads(node_index)` is called once a node's parameter
    gradients are complete (used to launch bucketed all-reduces overlapped with the rest of backward)."""
    be = _ops.backend()
    dslots: Dict[int, torch.Tensor] = {plan.output_slot: dfeat}

    def add_grad(slot, g):
        if g is None:
            return
        if slot in dslots:
            dslots[slot] = dslots[slot] + g
        else:
            dslots[slot] = g

    for ni in range(len(plan.nodes) - 1, -1, -1):
        node = plan.nodes[ni]
        if isinstance(node, ConvBN):
            sv = ctx.saved.pop(ni)
            dout = dslots.pop(node.dst)
            bn = node.bn
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
            be.conv_wgrad(sv.cg, sv.x, dy, grad_of(node.conv.weight))
            if after_param_grads is not None:
                after_param_grads(ni)
            if node.src != plan.input_slot:
                add_grad(node.src, be.conv_dgrad(sv.cg, dy, node.conv.weight.data))
            del dy, dout, sv
        else:
            raise NotImplementedError(f"plan node {type(node).__name__}")
