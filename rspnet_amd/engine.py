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
    into: Optional[Tuple[int, int, int]] = None    # (concat slot, channel offset, total channels): write a slice
    cout_pad: int = 0                              # run with Cout zero-padded to this many channels (0: as is).  R(2+1)D's
    #   mid-channel counts (83, 230, 921 ...) are not multiples of 4; padded, this conv's output and the next conv's input are
    #   16-byte rows and both take the LDS-DMA kernels instead of the scalar gather.  Pad channels carry zero weights and
    #   gamma = beta = 0, so they are exactly 0 after BN+ReLU and contribute nothing downstream; their gradients are dropped.


@dataclass
class Pool:
    """Stand-alone MaxPool3d (overlapping / padded windows allowed)."""
    src: int
    dst: int
    k: Triple
    s: Triple
    p: Triple = (0, 0, 0)


@dataclass
class Gate:
    """S3D-G self-gating: x * sigmoid(conv1x1x1(mean(x)))   (models/s3dg.py:63-72)."""
    conv: nn.Module                                # excitation: weight (C,C,1,1,1), bias (C)
    src: int
    dst: int
    into: Optional[Tuple[int, int, int]] = None


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
            w = _ops.backend().conv_pack_fwd(cg, pad_weight(node.conv.weight.data, cg.Cout, cg.Cin))
            self._cache[key] = w
        return w


def pad_in_channels(w: torch.Tensor, cin: int) -> torch.Tensor:
    """(Cout,Cin,k..) weight zero-padded to `cin` input channels — the 3-channel stems run on clips padded to 4
    channels so their im2col gathers are 16-byte loads (INPUT_CHANNEL_PAD)."""
    if w.shape[1] == cin:
        return w
    out = torch.zeros((w.shape[0], cin) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
    out[:, :w.shape[1]] = w
    return out


def pad_weight(w: torch.Tensor, cout: int, cin: int) -> torch.Tensor:
    """(Cout,Cin,k..) weight zero-padded to (cout, cin, k..)."""
    if w.shape[0] == cout:
        return pad_in_channels(w, cin)
    out = torch.zeros((cout, cin) + tuple(w.shape[2:]), dtype=w.dtype, device=w.device)
    out[:w.shape[0], :w.shape[1]] = w
    return out


def _pad_vec(v: torch.Tensor, n: int, fill: float = 0.0) -> torch.Tensor:
    out = torch.full((n,), fill, dtype=v.dtype, device=v.device)
    out[:v.shape[0]] = v
    return out


INPUT_CHANNEL_PAD = 4


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


def run_forward(plan: Plan, x: torch.Tensor, packed: PackedWeights, keep: bool, training: bool = True
                ) -> Tuple[torch.Tensor, Optional[ForwardCtx]]:
    """Execute `plan` on x (N,D,H,W,C).  keep=True records what backward needs.  training=True: batch-statistics BN (the
    pretext step never runs anything else: pretrain.py:225); training=False: running-statistics BN for the fine-tune /
    validation forward (finetune.py:333-345), no backward."""
    be = _ops.backend()
    assert training or not keep, "eval-mode forward keeps nothing for backward"
    slots: Dict[int, torch.Tensor] = {plan.input_slot: x}
    ctx = ForwardCtx() if keep else None
    for ni, node in enumerate(plan.nodes):
        if isinstance(node, ConvBN):
            xin = slots[node.src]
            N, D, H, W, Cin = xin.shape
            w = node.conv.weight
            Cout = w.shape[0]
            Cp = node.cout_pad if node.cout_pad > Cout else Cout
            cg = ConvGeom(N, D, H, W, Cin, Cp, node.k, node.s, node.p, Cin_alg=w.shape[1])
            bias = getattr(node.conv, "bias", None)
            bn = node.bn
            bias_d = None if bias is None else (bias.data if Cp == Cout else _pad_vec(bias.data, Cp))
            if training:
                y, stats = be.conv_fwd(cg, xin, packed.get(node, cg), bias_d, True)
                if Cp == Cout:
                    mi, ss = be.bn_finalize(stats, cg.rows, bias_d, bn.weight.data, bn.bias.data, float(bn.eps),
                                            float(bn.momentum), bn.running_mean, bn.running_var)
                else:
                    rm, rv = _pad_vec(bn.running_mean, Cp), _pad_vec(bn.running_var, Cp, 1.0)
                    mi, ss = be.bn_finalize(stats, cg.rows, bias_d, _pad_vec(bn.weight.data, Cp), _pad_vec(bn.bias.data, Cp),
                                            float(bn.eps), float(bn.momentum), rm, rv)
                    bn.running_mean.copy_(rm[:Cout])
                    bn.running_var.copy_(rv[:Cout])
            else:
                y, _ = be.conv_fwd(cg, xin, packed.get(node, cg), None, False)     # bias folded into the shift
                ss = _eval_scale_shift(bn, bias)
                mi = None
                if Cp != Cout:
                    ss = torch.cat([ss, torch.zeros((2, Cp - Cout), dtype=ss.dtype, device=ss.device)], dim=1).contiguous()
            do, ho, wo = cg.out_dims
            pk, ps = node.pool if node.pool else ((1, 1, 1), (1, 1, 1))
            pg = PoolGeom(N, do, ho, wo, cg.Cout, pk, ps, (0, 0, 0))
            res = slots[node.residual] if node.residual is not None else None
            if node.into is not None:
                pdo, pho, pwo = pg.out_dims
                out = _view(_slice_of(slots, node.into, (N, pdo, pho, pwo), xin.device), node.into, cg.Cout)
                be.bn_act_pool_fwd(pg, y, ss, res, node.relu, out=out)
            else:
                slots[node.dst] = be.bn_act_pool_fwd(pg, y, ss, res, node.relu)
            if keep:
                ctx.saved[ni] = _Saved(xin, y, mi, ss, cg, pg, res)
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
            if node.into is not None:
                dout = _view(dslots[node.into[0]], node.into, sv.cg.Cout)
            else:
                dout = dslots.pop(node.dst)
            bn = node.bn
            Cout, Cp = node.conv.weight.shape[0], sv.cg.Cout
            if Cp == Cout:
                dy, dres = be.bn_act_pool_bwd(sv.pg, sv.y, sv.res, dout, bn.weight.data, sv.mi, sv.ss, node.relu,
                                              node.residual is not None, grad_of(bn.weight), grad_of(bn.bias))
            else:
                dgam = torch.empty(Cp, dtype=torch.float32, device=dout.device)
                dbet = torch.empty(Cp, dtype=torch.float32, device=dout.device)
                dy, dres = be.bn_act_pool_bwd(sv.pg, sv.y, sv.res, dout, _pad_vec(bn.weight.data, Cp), sv.mi, sv.ss, node.relu,
                                              node.residual is not None, dgam, dbet)
                for g, src in ((grad_of(bn.weight), dgam), (grad_of(bn.bias), dbet)):
                    if g is not None:
                        g.copy_(src[:Cout])
            if node.residual is not None:
                add_grad(node.residual, dres)
            bias = getattr(node.conv, "bias", None)
            if bias is not None:
                gb = grad_of(bias)
                if gb is not None:
                    # A conv bias in front of train-mode BatchNorm has an identically-zero gradient (BN subtracts the
                    # batch mean); the reference's autograd produces ~1e-8 rounding noise there.
                    gb.zero_()
            gw = grad_of(node.conv.weight)
            if gw is not None and (gw.shape[1] != sv.cg.Cin or gw.shape[0] != Cp):   # channel-padded: drop the pad gradients
                gpad = torch.empty((Cp, sv.cg.Cin) + tuple(gw.shape[2:]), dtype=gw.dtype, device=gw.device)
                be.conv_wgrad(sv.cg, sv.x, dy, gpad)
                gw.copy_(gpad[:gw.shape[0], :gw.shape[1]])
            else:
                be.conv_wgrad(sv.cg, sv.x, dy, gw)
            if after_param_grads is not None:
                after_param_grads(ni)
            if node.src != plan.input_slot:
                add_grad(node.src, be.conv_dgrad(sv.cg, dy, pad_weight(node.conv.weight.data, Cp, sv.cg.Cin)))
            del dy, dout, sv
        elif isinstance(node, Pool):
            pg, idx = ctx.saved.pop(ni)
            add_grad(node.src, be.maxpool_bwd(pg, dslots.pop(node.dst), idx))
        elif isinstance(node, Gate):
            xin, mean, gate = ctx.saved.pop(ni)
            dout = (_view(dslots[node.into[0]], node.into, xin.shape[4]) if node.into is not None
                    else dslots.pop(node.dst))
            add_grad(node.src, be.gate_bwd(xin, dout, node.conv.weight.data, mean, gate, grad_of(node.conv.weight),
                                           grad_of(node.conv.bias)))
            if after_param_grads is not None:
                after_param_grads(ni)
        else:
            raise NotImplementedError(f"plan node {type(node).__name__}")
