"""MultiTaskWrapper: backbone + two projection heads (A-VID, RSP), L2-normalised outputs.

Mirror of /root/reference/moco/split_wrapper.py:66-190 for the pretext configuration (finetune=False, groups=1) with every
`fc_type` the reference accepts: 'linear' (the shipped configs), 'mlp', 'conv' (ConvFc :18-39), 'convbn' (ConvBnFc :42-63) and
'speednet' (second head = Linear(feat,1) + sigmoid, :125-126,146-147).  Heads keep the reference's containers so the
state-dict keys are ``fc1.2.weight`` / ``fc1.conv1.weight`` / ``fc1.bn.running_mean`` ....  For 'linear' the
pool+linear+normalize arithmetic is one HIP kernel (rsp_head_fwd); the other types are composed from the generic pieces
(engine sub-plan for the head's convolutions, spatial mean, linear, l2-norm / sigmoid).

finetune=True (split_wrapper.py:104-106,131-135, built by models/__init__.py:125-143) is the downstream
classifier: backbone -> AdaptiveAvgPool3d(1) -> Linear(feat, num_classes).  Its forward is a regular autograd node
(`_FinetuneFn`) over the module's parameters, so `nn.CrossEntropyLoss`, any torch optimizer and DistributedDataParallel
work on it exactly as finetune.py uses them; `model.eval()` switches BatchNorm to its running statistics.
"""
from typing import Callable

import torch
from torch import Tensor, nn

from .. import ops as _ops
from ..engine import INPUT_CHANNEL_PAD, ConvBN, ConvBias, PackedWeights, Plan, run_backward, run_backward_iter, run_forward

FC_TYPES = ("linear", "mlp", "conv", "convbn", "speednet")


class Flatten(nn.Module):
    def forward(self, x: Tensor):
        return x.flatten(1)


class ConvFc(nn.Module):
    """conv -> relu -> conv -> global average -> linear (parameter holder; split_wrapper.py:18-39)."""

    def __init__(self, feat_dim: int, moco_dim: int, kernel_size, padding):
        super().__init__()
        self.conv1 = nn.Conv3d(feat_dim, feat_dim, kernel_size, padding=padding)
        self.relu = nn.ReLU(inplace=True)
        self.conv2 = nn.Conv3d(feat_dim, feat_dim, kernel_size, padding=padding)
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.linear = nn.Linear(feat_dim, moco_dim)
        self.kernel_size, self.padding = tuple(kernel_size), tuple(padding)

    def plan(self) -> Plan:
        return Plan([ConvBias(self.conv1, 0, 1, self.kernel_size, (1, 1, 1), self.padding, relu=True),
                     ConvBias(self.conv2, 1, 2, self.kernel_size, (1, 1, 1), self.padding, relu=False)], 0, 2)


class ConvBnFc(nn.Module):
    """conv -> bn -> relu -> global average -> linear (parameter holder; split_wrapper.py:42-63)."""

    def __init__(self, feat_dim: int, moco_dim: int, kernel_size, padding):
        super().__init__()
        self.conv1 = nn.Conv3d(feat_dim, feat_dim, kernel_size, padding=padding)
        self.bn = nn.BatchNorm3d(feat_dim)
        self.relu = nn.ReLU(inplace=True)
        self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
        self.linear = nn.Linear(feat_dim, moco_dim)
        self.kernel_size, self.padding = tuple(kernel_size), tuple(padding)

    def plan(self) -> Plan:
        return Plan([ConvBN(self.conv1, self.bn, 0, 1, self.kernel_size, (1, 1, 1), self.padding, relu=True)], 0, 1)


class MultiTaskWrapper(nn.Module):
    def __init__(self, base_encoder: Callable[[int], nn.Module], num_classes: int = 128, finetune: bool = False,
                 fc_type: str = "linear", groups: int = 1):
        super().__init__()
        if groups != 1:
            raise NotImplementedError("groups != 1 is not used by any shipped pretext config")
        if not finetune and fc_type not in FC_TYPES:
            # the reference silently builds no heads for an unknown type and fails in forward (split_wrapper.py:107-126,138)
            raise ValueError(f"unknown fc_type '{fc_type}' (expected one of {FC_TYPES})")
        self.finetune = finetune
        self.moco_dim = num_classes
        self.num_classes = num_classes
        self.groups = groups
        self.fc_type = fc_type
        self.feat = None

        self.encoder = base_encoder(num_classes=1)
        feat_dim = self._get_feat_dim(self.encoder)
        if finetune:
            self.avg_pool = nn.AdaptiveAvgPool3d((1, 1, 1))
            self.fc = nn.Linear(feat_dim, num_classes)
        else:
            if fc_type in ("linear", "mlp"):
                make = self._get_linear_fc if fc_type == "linear" else self._get_mlp_fc
                self.fc1 = make(feat_dim, self.moco_dim)
                self.fc2 = make(feat_dim, self.moco_dim)
            elif fc_type in ("conv", "convbn"):
                cls = ConvFc if fc_type == "conv" else ConvBnFc
                self.fc1 = cls(feat_dim, self.moco_dim, (3, 3, 3), (1, 1, 1))
                self.fc2 = cls(feat_dim, self.moco_dim, (3, 3, 3), (1, 1, 1))
            else:   # speednet: the second head is a 1-d speed-up probability (split_wrapper.py:124-126)
                self.fc1 = self._get_linear_fc(feat_dim, self.moco_dim)
                self.fc2 = self._get_linear_fc(feat_dim, 1)

        self._plan = None
        self._packed = PackedWeights()

    # ---- reference surface -------------------------------------------------------------------------------------
    @staticmethod
    def _get_linear_fc(feat_dim: int, moco_dim: int):
        return nn.Sequential(nn.AdaptiveAvgPool3d((1, 1, 1)), Flatten(), nn.Linear(feat_dim, moco_dim))

    @staticmethod
    def _get_mlp_fc(feat_dim: int, moco_dim: int):
        # split_wrapper.py:171-179: indices 2 and 4 carry the parameters (state-dict keys fcN.2.*, fcN.4.*)
        return nn.Sequential(nn.AdaptiveAvgPool3d((1, 1, 1)), Flatten(), nn.Linear(feat_dim, feat_dim), nn.ReLU(inplace=True),
                             nn.Linear(feat_dim, moco_dim))

    @staticmethod
    def _get_feat_dim(encoder):
        # split_wrapper.py:181-190: looks for fc / new_fc / classifier, else 512
        for fc_name in ("fc", "new_fc", "classifier"):
            if hasattr(encoder, fc_name):
                return getattr(encoder, fc_name).in_features
        return 512

    def _get_last_feature(self):
        return self.feat

    def _get_fc_weight(self):
        return self.fc1[2].weight.data, self.fc2[2].weight.data

    def untrained_prefixes(self):
        return tuple("encoder." + n + "." for n in getattr(self.encoder, "classifier_names", ()))

    def adjacent_parameters(self):
        """Parameter groups the flat buffers should keep contiguous (engine.ConvBNGroup)."""
        fn = getattr(self.encoder, "adjacent_parameters", None)
        return tuple(tuple("encoder." + n for n in g) for g in fn()) if fn is not None else ()

    # ---- execution ---------------------------------------------------------------------------------------------
    def plan(self):
        if self._plan is None:
            self._plan = self.encoder.plan()
        return self._plan

    def weights_changed(self):
        self._packed.invalidate()

    def forward_ndhwc(self, x: Tensor, keep: bool, deferred=None):
        """x: (N,T,H,W,C).  Returns (x1, x2, ctx) — ctx is what backward_ndhwc needs (None when keep=False).  deferred: see
        engine.run_forward (BatchNorm layers that report their batch moments instead of moving their running statistics)."""
        be = _ops.backend()
        feat, ctx = run_forward(self.plan(), x, self._packed, keep, deferred=deferred)
        self.feat = feat
        if self.fc_type == "linear":
            l1, l2 = self.fc1[2], self.fc2[2]
            x1, x2, pooled, raw = be.head_fwd(feat, l1.weight.data, l1.bias.data, l2.weight.data, l2.bias.data)
            if keep:
                ctx.head = (pooled, raw)
            return x1, x2, ctx
        # generic composition: [head convolutions] -> spatial mean -> linear [-> relu -> linear] -> l2-norm | sigmoid
        pooled_feat = be.spatial_mean_fwd(feat) if self.fc_type in ("mlp", "speednet") else None
        outs, saved = [], []
        for hi, fc in enumerate((self.fc1, self.fc2)):
            hctx = hid = None
            if self.fc_type in ("conv", "convbn"):
                h, hctx = run_forward(fc.plan(), feat, self._packed, keep, deferred=deferred)
                pooled = be.spatial_mean_fwd(h)
                raw = be.linear_fwd(pooled, fc.linear.weight.data, fc.linear.bias.data, False)
                hshape = tuple(h.shape)
            elif self.fc_type == "mlp":
                pooled, hshape = pooled_feat, None
                hid = be.linear_fwd(pooled, fc[2].weight.data, fc[2].bias.data, True)
                raw = be.linear_fwd(hid, fc[4].weight.data, fc[4].bias.data, False)
            else:
                pooled, hshape = pooled_feat, None
                raw = be.linear_fwd(pooled, fc[2].weight.data, fc[2].bias.data, False)
            sig = self.fc_type == "speednet" and hi == 1
            out = be.eltwise("sigmoid_fwd", raw) if sig else be.l2norm_fwd(raw)
            outs.append(out)
            saved.append((hctx, hshape, pooled, hid, raw, out))
        if keep:
            ctx.head = saved
        return outs[0], outs[1], ctx

    def backward_ndhwc(self, ctx, d1: Tensor, d2: Tensor, grad_of, after_param_grads=None):
        for _ in self.backward_ndhwc_iter(ctx, d1, d2, grad_of, after_param_grads):
            pass

    def backward_ndhwc_iter(self, ctx, d1: Tensor, d2: Tensor, grad_of, after_param_grads=None):
        """Generator: the head's backward, then the backbone's node by node (engine.run_backward_iter); yields plan node indices
        (-1 after the head)."""
        be = _ops.backend()
        if self.fc_type == "linear":
            l1, l2 = self.fc1[2], self.fc2[2]
            pooled, raw = ctx.head
            dfeat = be.head_bwd(d1.contiguous(), d2.contiguous(), pooled, raw, l1.weight.data, l2.weight.data, ctx.feat_shape,
                                grad_of(l1.weight), grad_of(l1.bias), grad_of(l2.weight), grad_of(l2.bias))
        else:
            dfeat = dpooled_feat = None
            for hi, (fc, (hctx, hshape, pooled, hid, raw, out), d) in enumerate(zip((self.fc1, self.fc2), ctx.head, (d1, d2))):
                sig = self.fc_type == "speednet" and hi == 1
                draw = be.eltwise("sigmoid_bwd", out, d.contiguous()) if sig else be.l2norm_bwd(raw, d.contiguous())
                if self.fc_type in ("conv", "convbn"):
                    lin = fc.linear
                    dp = be.linear_bwd(pooled, raw, draw, lin.weight.data, False, grad_of(lin.weight), grad_of(lin.bias))
                    dh = be.spatial_mean_bwd(dp, hshape)
                    g = run_backward(fc.plan(), hctx, dh, grad_of, None, want_input_grad=True)
                    dfeat = g if dfeat is None else be.eltwise("add", dfeat, g, out=g)
                    continue
                if self.fc_type == "mlp":
                    dhid = be.linear_bwd(hid, raw, draw, fc[4].weight.data, False, grad_of(fc[4].weight), grad_of(fc[4].bias))
                    dp = be.linear_bwd(pooled, hid, dhid, fc[2].weight.data, True, grad_of(fc[2].weight), grad_of(fc[2].bias))
                else:
                    dp = be.linear_bwd(pooled, raw, draw, fc[2].weight.data, False, grad_of(fc[2].weight), grad_of(fc[2].bias))
                dpooled_feat = dp if dpooled_feat is None else be.eltwise("add", dpooled_feat, dp, out=dp)
            if dfeat is None:
                dfeat = be.spatial_mean_bwd(dpooled_feat, ctx.feat_shape)
        if after_param_grads is not None:
            after_param_grads(-1, None)
        yield -1
        yield from run_backward_iter(self.plan(), ctx, dfeat, grad_of, after_param_grads)

    def _to_ndhwc(self, x: Tensor) -> Tensor:
        be = _ops.backend()
        B = x.shape[0]
        src = torch.arange(B, dtype=torch.int32, device=x.device)
        step = torch.ones(B, dtype=torch.int32, device=x.device)
        return be.clip_gather(x.contiguous(), src, step, x.shape[2], max(x.shape[1], INPUT_CHANNEL_PAD))

    def _bump_bn_counters(self):
        for mod in self.modules():          # train-mode BN bookkeeping (inside the pretext model this is one fused add)
            if isinstance(mod, nn.modules.batchnorm._BatchNorm):
                mod.num_batches_tracked += 1

    def forward(self, x: Tensor):
        """Reference signature: x is NCDHW (B,3,T,H,W).  Pretext wrapper: returns the two unit-norm embeddings (no autograd
        here — inside the pretext model gradients flow through MoCoDiffLossTwoFc's own autograd node).  finetune=True:
        returns the (B, num_classes) logits as an autograd node over this module's parameters."""
        if self.finetune:
            params = [p for p in self.parameters()]
            return _FinetuneFn.apply(self, torch.is_grad_enabled(), x, *params)
        xn = self._to_ndhwc(x)
        with torch.no_grad():
            x1, x2, _ = self.forward_ndhwc(xn, keep=False)
            self._bump_bn_counters()
        return x1, x2


class _FinetuneFn(torch.autograd.Function):
    """logits = Linear(mean_{T,H,W}(backbone(x)))  (split_wrapper.py:131-135) on the HIP kernels."""

    @staticmethod
    def forward(ctx, module: MultiTaskWrapper, grad_on: bool, x: Tensor, *params):
        be = _ops.backend()
        training = module.encoder.training                 # BatchNorm mode (only_train_fc keeps the backbone in eval mode)
        names = [n for n, _ in module.named_parameters()]
        backbone_grad = grad_on and any(need for n, need in zip(names, ctx.needs_input_grad[3:]) if n.startswith("encoder."))
        keep = training and backbone_grad
        ctx.backbone_grad = backbone_grad
        # weights may have been stepped by any optimizer since the last call: re-pack when their version counters moved
        ver = sum(p._version for p in params)
        if ver != getattr(module, "_packed_version", None):
            module._packed.invalidate()
            module._packed_version = ver
        xn = module._to_ndhwc(x)
        feat, ectx = run_forward(module.plan(), xn, module._packed, keep, training=training)
        module.feat = feat
        if training:
            module._bump_bn_counters()
        pooled = be.spatial_mean_fwd(feat)
        logits = be.linear_fwd(pooled, module.fc.weight.data, module.fc.bias.data, False)
        ctx.module, ctx.ectx, ctx.params = module, ectx, params
        ctx.head = (pooled, logits)
        ctx.feat_shape = tuple(feat.shape)
        return logits

    @staticmethod
    def backward(ctx, dlogits: Tensor):
        be = _ops.backend()
        module, params = ctx.module, ctx.params
        grads = {}

        def grad_of(p):
            g = torch.empty_like(p)
            grads[id(p)] = g
            return g

        pooled, logits = ctx.head
        dpooled = be.linear_bwd(pooled, logits, dlogits.contiguous(), module.fc.weight.data, False, grad_of(module.fc.weight),
                                grad_of(module.fc.bias))
        if ctx.ectx is None and ctx.backbone_grad:
            raise RuntimeError("backward through an eval-mode backbone (BatchNorm on running statistics) is not implemented: "
                               "freeze the backbone (only_train_fc) or call model.train()")
        if ctx.ectx is not None:                          # `only_train_fc` (models/__init__.py:82-104) stops at the classifier
            dfeat = be.spatial_mean_bwd(dpooled, ctx.feat_shape)
            run_backward(module.plan(), ctx.ectx, dfeat, grad_of)
        ctx.ectx = None
        return (None, None, None) + tuple(grads.get(id(p)) if p.requires_grad else None for p in params)
