"""Full-size check of "the replayed step is the eager step, bit for bit" (tests/test_graph_step_gpu.py holds it at fixture size):
two identically seeded models, one driven by the five statements, one by GraphedPretextStep(issue="graph"); per step the losses and
logits, at the end every state tensor.  On the first mismatch: which tensors differ, and by how much.

    python3 tools/graph_vs_eager_fullsize.py resnet18 32 112 12"""
import os
import random
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from golden_util import load_spec  # noqa: E402
from model_util import make_cfg  # noqa: E402
from oracle import portable as P  # noqa: E402  (test infrastructure: seeded states and clips only)


def main():
    arch, B, HW, steps = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    K = 16384
    dev = torch.device("cuda", 0)
    from rspnet_amd.graph_step import GraphedPretextStep
    from rspnet_amd.moco import Loss, ModelFactory
    from rspnet_amd.optim import SGD
    clips = [tuple(torch.from_numpy(c).to(dev) for c in P.clips(10 + i, 0, (B, 3, 32, HW, HW))) for i in range(min(steps, 4))]
    results = []
    light = steps > 16                      # long runs: checksums instead of copies of every tensor
    hows = ("eager", "graph", "graph") if light else ("eager", "graph")
    for how in hows:
        torch.manual_seed(7)
        torch.cuda.manual_seed(7)
        random.seed(7)
        wrapped = ModelFactory(make_cfg(arch, K)).build_moco_diffloss(device=dev)
        spec = dict(load_spec(arch))
        spec["queue"] = ((128, K), "float32")
        wrapped.module.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in P.fill_state(spec, 3).items()})
        wrapped.train()
        crit = Loss(margin=2.0, A=1.0, M=1.0)
        opt = SGD(wrapped.parameters(), lr=0.05, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
        stepper = GraphedPretextStep(wrapped, crit, opt, warmup=2, issue="graph") if how == "graph" else None
        trace = []
        for i in range(steps):
            im_q, im_k = clips[i % len(clips)]
            if stepper is None:
                out, tgt, rl, rt = wrapped(im_q, im_k)
                loss, la, lm = crit(out, tgt, rl, rt)
                opt.zero_grad()
                loss.backward()
                opt.step()
            else:
                loss, la, lm, out, rl = stepper(im_q, im_k)
            sd = wrapped.module.state_dict()
            if light:
                fl = wrapped.module._flat
                per = torch.stack([fl.g_flat[o:o + n].double().abs().sum() for o, n in (fl.offsets[nm] for nm in fl.names[:fl.n_trained_params])])
                bns = torch.stack([b.double().sum() for k, b in sd.items() if k.endswith(("running_mean", "running_var"))])
                trace.append((loss.detach().clone(), out[0].detach().clone(), fl.g_flat.double().sum().reshape(1),
                              {"q_flat": fl.q_flat.double().sum().reshape(1), "k_flat": fl.k_flat.double().sum().reshape(1),
                               "bn": wrapped.module._bn_flat.double().sum().reshape(1), "queue": wrapped.module.queue.double().sum().reshape(1),
                               "per_param_grad": per, "per_bn_buffer": bns}))
                pnames = list(fl.names[:fl.n_trained_params])
                bnames = [k for k in sd if k.endswith(("running_mean", "running_var"))]
            else:
                trace.append((loss.detach().clone(), out[0].detach().clone(), wrapped.module._flat.g_flat.clone(),
                              {k: v.detach().clone() for k, v in sd.items() if k.startswith("encoder_q")}))
        torch.cuda.synchronize()
        if stepper is not None:
            print("graph mode", stepper.mode, "disabled", stepper.disabled, stepper.fallback_reason,
                  [op[:3] for op in next(iter(stepper.graphs.values()))[3]] if stepper.graphs else None)
        results.append(trace)
    if light:
        te, tg, tg2 = results
        first = {}
        for tag, a, b in (("eager vs graph", te, tg), ("graph vs graph (second run)", tg, tg2)):
            for i, ((l0, o0, g0, s0), (l1, o1, g1, s1)) in enumerate(zip(a, b)):
                bad = [n for n, x, y in (("loss", l0, l1), ("logits", o0, o1), ("grad checksum", g0, g1)) if not torch.equal(x, y)]
                bad += [k for k in s0 if not torch.equal(s0[k], s1[k])]
                if bad:
                    first[tag] = (i, bad, float(l0), float(l1))
                    dp = (s0["per_param_grad"] != s1["per_param_grad"]).nonzero().flatten().tolist()
                    db = (s0["per_bn_buffer"] != s1["per_bn_buffer"]).nonzero().flatten().tolist()
                    print(tag, "step", i, ":", len(dp), "of", len(pnames), "parameter gradients differ:", [pnames[j] for j in dp][:12],
                          "...", [pnames[j] for j in dp][-4:], ";", len(db), "BN buffers differ:", [bnames[j] for j in db][:8])
                    break
            print(tag, "-> first difference:", first.get(tag, "none in %d steps" % len(a)))
        print("final losses", float(te[-1][0]), float(tg[-1][0]), float(tg2[-1][0]))
        return
    te, tg = results
    names = None
    for i, ((l0, o0, g0, s0), (l1, o1, g1, s1)) in enumerate(zip(te, tg)):
        same = torch.equal(l0, l1) and torch.equal(o0, o1) and torch.equal(g0, g1) and all(torch.equal(s0[k], s1[k]) for k in s0)
        print(f"step {i}: loss {float(l0):.7f} / {float(l1):.7f}  logits equal {torch.equal(o0, o1)}  gradients equal {torch.equal(g0, g1)}  "
              f"state equal {all(torch.equal(s0[k], s1[k]) for k in s0)}")
        if not same and names is None:
            names = [k for k in s0 if not torch.equal(s0[k], s1[k])]
            print("  first mismatch at step", i, ":", len(names), "state tensors differ; first ten:")
            for k in names[:10]:
                d = (s0[k].double() - s1[k].double()).abs().max().item()
                print(f"    {k}: max abs diff {d:.3e} of max {s0[k].abs().max().item():.3e}")
            gd = (g0.double() - g1.double()).abs()
            print(f"  flat gradient: {int((gd > 0).sum())} of {gd.numel()} elements differ, max abs diff {gd.max().item():.3e}; first differing offset {int((gd > 0).nonzero()[0]) if (gd > 0).any() else None}")
    print("ALL EQUAL" if names is None else "MISMATCH")


if __name__ == "__main__":
    main()
