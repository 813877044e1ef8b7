"""TEST-ONLY op backend: every method of rspnet_amd.ops.HipOps restated with plain torch CPU ops.

Two uses: (1) `-m "not gpu"` tests run the *host logic* of rspnet_amd (layer-plan executor, flat parameters,
shuffle-BN exchange, optimizer, checkpoint surface) against the golden fixtures without a GPU; (2) `-m gpu` tests
use the same functions as the per-kernel contract when comparing HIP kernels with torch fp32 references.
Never imported by the product.
"""
import torch
import torch.nn.functional as F

from rspnet_amd.ops import ConvGeom, PoolGeom


def _ncdhw(x):
    return x.permute(0, 4, 1, 2, 3)


def _ndhwc(x):
    return x.permute(0, 2, 3, 4, 1).contiguous()


class CpuOps:
    name = "cpu-checker"

    def conv_pack_fwd(self, g: ConvGeom, w_ref):
        return w_ref.detach().clone()

    def conv_fwd(self, g: ConvGeom, x, w_packed, bias, want_stats, out=None, out_ld=None, in_ld=None):
        y0 = F.conv3d(_ncdhw(x), w_packed, None, stride=g.s, padding=g.p)
        stats = None
        if want_stats:
            s = y0.double().sum(dim=(0, 2, 3, 4))
            ss = (y0.double() ** 2).sum(dim=(0, 2, 3, 4))
            stats = torch.stack([s, ss], dim=1).float().unsqueeze(0).contiguous()   # one "tile"
            stats = stats.double()  # keep precision of the single tile
        if bias is not None:
            y0 = y0 + bias.view(1, -1, 1, 1, 1)
        if out is not None:                      # a channel-slice view of a wider tensor (pitch out_ld)
            out.copy_(_ndhwc(y0))
            return out, stats
        return _ndhwc(y0), stats

    def conv_dgrad(self, g: ConvGeom, dy, w_ref):
        x_shape = (g.N, g.Cin, g.Di, g.Hi, g.Wi)
        dx = torch.nn.grad.conv3d_input(x_shape, w_ref, _ncdhw(dy).contiguous(), stride=g.s, padding=g.p)
        return _ndhwc(dx)

    def conv_wgrad(self, g: ConvGeom, x, dy, dw_out, dbias_out=None):
        # dw_out may hold fewer channels than the (zero-padded) geometry: the padding channels' gradients are dropped
        full = (g.Cout, g.Cin) + tuple(dw_out.shape[2:])
        dw = torch.nn.grad.conv3d_weight(_ncdhw(x).contiguous(), full, _ncdhw(dy).contiguous(), stride=g.s, padding=g.p)
        dw_out.copy_(dw[:dw_out.shape[0], :dw_out.shape[1]])
        if dbias_out is not None:
            dbias_out.copy_(dy.sum(dim=(0, 1, 2, 3)))

    def bn_ema_set(self, entries):
        return _CpuEmaSet(entries)

    def bn_finalize(self, stats, count, conv_bias, gamma, beta, eps, momentum, running_mean, running_var, batch_stats_out=None):
        C, Cv = stats.shape[1], gamma.shape[0]
        if batch_stats_out is not None:
            # deferred running-statistics update: report the batch moments (mean incl. conv bias, unbiased variance) instead
            s = stats.double().sum(dim=0)[:Cv]
            mean0 = s[:, 0] / count
            var = (s[:, 1] / count - mean0 * mean0).clamp_min(0)
            batch_stats_out[0].copy_((mean0 + (conv_bias.double() if conv_bias is not None else 0)).float())
            batch_stats_out[1].copy_((var * count / (count - 1) if count > 1 else var).float())
            running_mean = running_var = None
        if Cv < C:        # zero-padded output channels: parameters hold Cv entries, the padding gets gamma = beta = 0
            pad = lambda v, fill=0.0: None if v is None else torch.cat([v, torch.full((C - Cv,), fill, dtype=v.dtype)])
            rm = None if running_mean is None else pad(running_mean)
            rv = None if running_var is None else pad(running_var, 1.0)
            mi, ss = self.bn_finalize(stats, count, pad(conv_bias), pad(gamma), pad(beta), eps, momentum, rm, rv)
            if running_mean is not None:
                running_mean.copy_(rm[:Cv])
            if running_var is not None:
                running_var.copy_(rv[:Cv])
            return mi, ss
        s = stats.double().sum(dim=0)
        mean0 = s[:, 0] / count
        var = (s[:, 1] / count - mean0 * mean0).clamp_min(0)
        mean = mean0 + (conv_bias.double() if conv_bias is not None else 0)
        invstd = (1.0 / torch.sqrt(var + eps)).float()
        mi = torch.stack([mean.float(), invstd])
        sc = gamma * invstd
        ss = torch.stack([sc, beta - mean.float() * sc])
        if running_mean is not None:
            running_mean.mul_(1 - momentum).add_(momentum * mean.float())
        if running_var is not None:
            unb = var * count / (count - 1) if count > 1 else var
            running_var.mul_(1 - momentum).add_(momentum * unb.float())
        return mi, ss

    def _act(self, pg, y, ss, residual, relu):
        z = y * ss[0] + ss[1]
        if residual is not None:
            z = z + residual
        return z

    def bn_act_pool_fwd(self, pg: PoolGeom, y, scale_shift, residual, relu, out=None):
        z = self._act(pg, y, scale_shift, residual, relu)
        if relu:
            z = F.relu(z)
        if pg.k != (1, 1, 1) or pg.s != (1, 1, 1):
            z = _ndhwc(F.max_pool3d(_ncdhw(z), pg.k, pg.s, pg.p))
        if out is not None:
            out.copy_(z)
            return out
        return z.contiguous()

    def maxpool_fwd(self, pg: PoolGeom, x, keep):
        o, idx = F.max_pool3d(_ncdhw(x), pg.k, pg.s, pg.p, return_indices=True)
        return _ndhwc(o), (_ndhwc(idx).to(torch.int32) if keep else None)

    def maxpool_bwd(self, pg: PoolGeom, dout, idx):
        N, C = pg.N, pg.C
        dx = torch.zeros(N, C, pg.Di * pg.Hi * pg.Wi)
        dx.scatter_add_(2, _ncdhw(idx).reshape(N, C, -1).long(), _ncdhw(dout).reshape(N, C, -1))
        return _ndhwc(dx.view(N, C, pg.Di, pg.Hi, pg.Wi))

    def gate_fwd(self, x, w, b, out=None):
        mean = x.mean(dim=(1, 2, 3))
        gate = torch.sigmoid(F.linear(mean, w.view(w.shape[0], -1), b))
        o = x * gate.view(x.shape[0], 1, 1, 1, -1)
        if out is not None:
            out.copy_(o)
            return out, mean, gate
        return o, mean, gate

    def bn_act_gate_fwd(self, pg: PoolGeom, y, scale_shift, relu, w, b, keep_act, pool=None, out=None):
        a = self.bn_act_pool_fwd(pg, y, scale_shift, None, relu)
        o, mean, gate = self.gate_fwd(a, w, b)
        if pool is not None:
            assert not keep_act
            o = _ndhwc(F.max_pool3d(_ncdhw(o), pool.k, pool.s, pool.p)).contiguous()
        if out is not None:
            out.copy_(o)
            o = out
        return o, (a if keep_act else None), mean, gate

    def bn_act_gate_bwd(self, pg: PoolGeom, y, dout, gamma, mean_invstd, scale_shift, relu, w, mean, gate, dgamma_out, dbeta_out,
                        dw_out, db_out):
        a = self.bn_act_pool_fwd(pg, y, scale_shift, None, relu)
        dx = self.gate_bwd(a, dout, w, mean, gate, dw_out, db_out)
        dy, _ = self.bn_act_pool_bwd(pg, y, None, dx, gamma, mean_invstd, scale_shift, relu, False, dgamma_out, dbeta_out)
        return dy

    @torch.enable_grad()
    def gate_bwd(self, x, dout, w, mean, gate, dw_out, db_out):
        xx = x.detach().requires_grad_(True)
        W = w.detach().view(w.shape[0], -1).requires_grad_(True)
        B = torch.zeros(w.shape[0], requires_grad=True)
        pre = F.linear(xx.mean(dim=(1, 2, 3)), W) + B + (torch.logit(gate) - F.linear(mean, W)).detach()
        o = xx * torch.sigmoid(pre).view(x.shape[0], 1, 1, 1, -1)
        gx, gw, gb = torch.autograd.grad(o, [xx, W, B], dout)
        dw_out.copy_(gw.view_as(dw_out))
        db_out.copy_(gb)
        return gx.contiguous()

    @torch.enable_grad()
    def bn_act_pool_bwd(self, pg: PoolGeom, y, residual, dout, gamma, mean_invstd, scale_shift, relu, want_dres,
                        dgamma_out, dbeta_out, dy_out=None):
        C = y.shape[-1]
        if gamma is not None and gamma.shape[0] < C:      # zero-padded channels (see bn_finalize)
            Cv = gamma.shape[0]
            dg, db = torch.empty(C), torch.empty(C)
            res = self.bn_act_pool_bwd(pg, y, residual, dout, torch.cat([gamma, torch.zeros(C - Cv)]), mean_invstd, scale_shift, relu,
                                       want_dres, dg, db, dy_out=dy_out)
            if dgamma_out is not None:
                dgamma_out.copy_(dg[:Cv])
            if dbeta_out is not None:
                dbeta_out.copy_(db[:Cv])
            return res
        z = self._act(pg, y, scale_shift, residual, relu).detach().requires_grad_(True)
        a = F.relu(z) if relu else z
        if pg.k != (1, 1, 1) or pg.s != (1, 1, 1):
            a = _ndhwc(F.max_pool3d(_ncdhw(a), pg.k, pg.s, pg.p))
        (dz,) = torch.autograd.grad(a, z, dout)
        mean, invstd = mean_invstd[0], mean_invstd[1]
        xhat = (y - mean) * invstd
        n = y.numel() // y.shape[-1]
        s1 = dz.double().sum(dim=(0, 1, 2, 3))
        s2 = (dz.double() * xhat.double()).sum(dim=(0, 1, 2, 3))
        dy = gamma * invstd * (dz - (s1 / n).float() - xhat * (s2 / n).float())
        if dgamma_out is not None:
            dgamma_out.copy_(s2.float())
        if dbeta_out is not None:
            dbeta_out.copy_(s1.float())
        if dy_out is not None:
            dy_out.copy_(dy)
            return dy_out, (dz.contiguous() if want_dres else None)
        return dy.contiguous(), (dz.contiguous() if want_dres else None)

    def head_fwd(self, feat, w1, b1, w2, b2):
        pooled = feat.mean(dim=(1, 2, 3))
        r1 = F.linear(pooled, w1, b1)
        r2 = F.linear(pooled, w2, b2)
        return F.normalize(r1, dim=1), F.normalize(r2, dim=1), pooled, torch.stack([r1, r2])

    @torch.enable_grad()
    def head_bwd(self, d1, d2, pooled, raw, w1, w2, feat_shape, dw1, db1, dw2, db2):
        P = feat_shape[1] * feat_shape[2] * feat_shape[3]
        p = pooled.detach().requires_grad_(True)
        W1 = w1.detach().requires_grad_(True)
        W2 = w2.detach().requires_grad_(True)
        B1 = torch.zeros(w1.shape[0], requires_grad=True)
        B2 = torch.zeros(w2.shape[0], requires_grad=True)
        r1 = F.linear(p, W1) + B1 + (raw[0] - F.linear(pooled, w1)).detach()
        r2 = F.linear(p, W2) + B2 + (raw[1] - F.linear(pooled, w2)).detach()
        o = (F.normalize(r1, dim=1) * d1).sum() + (F.normalize(r2, dim=1) * d2).sum()
        gp, g1, g2, gb1, gb2 = torch.autograd.grad(o, [p, W1, W2, B1, B2])
        dw1.copy_(g1)
        dw2.copy_(g2)
        db1.copy_(gb1)
        db2.copy_(gb2)
        return (gp / P).view(feat_shape[0], 1, 1, 1, -1).expand(feat_shape).contiguous()

    def spatial_mean_fwd(self, x):
        return x.mean(dim=(1, 2, 3))

    def spatial_mean_bwd(self, dmean, shape):
        P = shape[1] * shape[2] * shape[3]
        return (dmean / P).view(shape[0], 1, 1, 1, -1).expand(shape).contiguous()

    def linear_fwd(self, x, w, b, relu):
        y = F.linear(x, w, b)
        return F.relu(y) if relu else y

    def linear_bwd(self, x, y, dy, w, relu, dw_out, db_out, want_dx=True):
        dz = dy * (y > 0) if relu else dy
        dw_out.copy_(dz.t() @ x)
        if db_out is not None:
            db_out.copy_(dz.sum(0))
        return dz @ w if want_dx else None

    def l2norm_fwd(self, x):
        return F.normalize(x, dim=1)

    @torch.enable_grad()
    def l2norm_bwd(self, x, dy):
        xx = x.detach().requires_grad_(True)
        (g,) = torch.autograd.grad(F.normalize(xx, dim=1), xx, dy)
        return g

    def logits_fwd(self, qA, qM, kA, kM, knegA, knegM, queue, inv_T):
        lneg = (qA @ queue) * inv_T
        l1 = torch.cat([(qA * kA).sum(1, keepdim=True) * inv_T, lneg], 1)
        l2 = torch.cat([(qA * knegA).sum(1, keepdim=True) * inv_T, lneg], 1)
        return l1, l2, (qM * kM).sum(1, keepdim=True) * inv_T, (qM * knegM).sum(1, keepdim=True) * inv_T

    def logits_bwd(self, dl1, dl2, dlp, dln, kA, kM, knegA, knegM, queue, inv_T):
        dqA = inv_T * (dl1[:, :1] * kA + dl2[:, :1] * knegA + (dl1[:, 1:] + dl2[:, 1:]) @ queue.t())
        dqM = inv_T * (dlp * kM + dln * knegM)
        return dqA, dqM

    @torch.enable_grad()
    def loss_fwd_bwd(self, l1, l2, lp, ln, margin, A, M):
        a = l1.detach().requires_grad_(True)
        b = l2.detach().requires_grad_(True)
        p = lp.detach().requires_grad_(True)
        n = ln.detach().requires_grad_(True)
        tgt = torch.zeros(a.shape[0], dtype=torch.long)
        ce = F.cross_entropy(a, tgt) + F.cross_entropy(b, tgt)
        rk = torch.clamp(margin - (p - n), min=0).mean()
        loss = A * ce + M * rk
        d1, d2, dp, dn = torch.autograd.grad(loss, [a, b, p, n])
        return torch.stack([loss.detach(), ce.detach(), rk.detach()]), d1, d2, dp, dn

    def queue_enqueue(self, queue, ptr, keys):
        queue[:, ptr:ptr + keys.shape[0]] = keys.t()

    def queue_enqueue_dev(self, queue, queue_ptr, keys):
        ptr, n = int(queue_ptr), keys.shape[0]
        queue[:, ptr:ptr + n] = keys.t()
        queue_ptr.fill_((ptr + n) % queue.shape[1])

    def clip_gather(self, im, src, step, T_out, c_out=None):
        out = torch.zeros((src.shape[0], T_out, im.shape[3], im.shape[4], c_out or im.shape[1]), dtype=im.dtype)
        for j in range(src.shape[0]):
            s = int(step[j])
            frames = im[int(src[j])][:, 0:T_out * s:s][:, :T_out]
            out[j][..., :im.shape[1]] = frames.permute(1, 2, 3, 0)
        return out

    def clip_gather_multi(self, jobs, T_out, c_out=None):
        return [self.clip_gather(im, src, step, T_out, c_out) for im, src, step in jobs]

    def momentum_update(self, k_flat, q_flat, m):
        k_flat.copy_(k_flat * m + q_flat * (1.0 - m))

    def sgd_step(self, p, g, buf, lr, mu, wd, gscale, first):
        d = g * gscale + wd * p
        if first:
            buf.copy_(d)
        else:
            buf.mul_(mu).add_(d)
        p.sub_(lr * buf)

    def rows_gather(self, x, idx):
        return x[idx.long()].contiguous()

    def eltwise(self, op, a, b=None, out=None):
        if op == "relu_fwd":
            r = F.relu(a)
        elif op == "relu_bwd":
            r = torch.where(a > 0, b, torch.zeros_like(b))
        elif op == "sigmoid_fwd":
            r = torch.sigmoid(a)
        elif op == "sigmoid_bwd":
            r = b * a * (1 - a)
        else:
            r = a + b
        if out is not None:
            out.copy_(r)
            return out
        return r

    def conv_dgrad_packed(self, g, dy, w_packed):
        return self.conv_dgrad(g, dy, w_packed)

    def pack_set(self, entries):
        return _CpuPackSet(entries)

    def pack_signature(self, g, which, w_ref):
        # checker "layout" = zero-padded reference layout: depends on the channel counts and the kernel only
        return (which, g.Cin, g.Cout, tuple(g.k), tuple(w_ref.shape))


class _CpuEmaSet:
    """Checker twin of rspnet_amd.ops.BnEmaSet."""

    def __init__(self, entries):
        self.entries = entries
        self.stats = [torch.zeros(2, rm.shape[0]) for rm, _, _ in entries]
        self.ptrs = [(rm.data_ptr(), rv.data_ptr()) for rm, rv, _ in entries]

    def run(self):
        for (rm, rv, mom), st in zip(self.entries, self.stats):
            rm.mul_(1 - mom).add_(mom * st[0])
            rv.mul_(1 - mom).add_(mom * st[1])


class _CpuPackSet:
    """Checker twin of rspnet_amd.ops.PackSet: the "packed" copy is the zero-padded reference-layout weight."""

    def __init__(self, entries):
        self.entries = entries
        self.packed = [torch.zeros((g.Cout, g.Cin) + tuple(w.shape[2:]), dtype=w.dtype) for g, which, w in entries]
        self.run()

    def run(self):
        for (g, which, w), out in zip(self.entries, self.packed):
            out.zero_()
            out[:w.shape[0], :w.shape[1]] = w.detach()
