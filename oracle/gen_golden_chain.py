"""Generate tests/golden/chain_c3d_ws2.npz from the REAL reference: THREE chained pretext steps at two ranks under DDP.
TEST INFRASTRUCTURE ONLY — build container only:  ``python -m oracle.gen_golden_chain``  (needs /root/reference).

What it pins.  torch's DistributedDataParallel broadcasts rank 0's BUFFERS — BatchNorm running
statistics, num_batches_tracked, the queue and its pointer — to every rank at the start of each forward
(/root/reference/moco/__init__.py:49-53: the default broadcast_buffers=True); the product keeps the running statistics per rank
between `sync_buffers()` calls.  Train-mode arithmetic never reads the running statistics, so rank 0's trajectory is the same
either way and rank 0 is the rank that writes checkpoints (/root/reference/pretrain.py:244-260).  The fixture holds, after each
of three chained steps (free-running: step s+1 starts from step s's post-step state, momentum buffers included), rank 0's and
rank 1's buffers and losses; tests/test_distributed_cpu.py holds the product's rank 0 to the fixture's rank 0 at every step
WITHOUT sync_buffers(), and its rank 1 to rank 0's buffers after sync_buffers().

Inputs: the clips and the pre-step state of the 2-rank C3D case of oracle/gen_golden.py (seed CHAIN_SEED, no guard band: only
forward quantities are compared); step s uses the permutations P.permutation(f"perm:{r}:s{s}") / ("shuffle1:s{s}", "shuffle2:s{s}").
"""
from __future__ import annotations

import json
import os
import sys
import tempfile
from typing import Dict

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import gen_golden as G  # noqa: E402
from oracle import portable as P  # noqa: E402

ARCH, B, HW, K, WS, CHAIN_SEED, STEPS = "c3d", 4, 32, 64, 2, 21, 3
BUFFER_SUFFIXES = ("running_mean", "running_var", "num_batches_tracked")


def chain_perms(step: int, ws: int = WS, b: int = B):
    """([per-rank _diff_speed permutation], (shuffle #1, shuffle #2)) of chained step `step`."""
    return ([P.permutation(f"perm:{r}:s{step}", CHAIN_SEED, b) for r in range(ws)],
            (P.permutation(f"shuffle1:s{step}", CHAIN_SEED, b * ws), P.permutation(f"shuffle2:s{step}", CHAIN_SEED, b * ws)))


def _worker(rank, ws, port, tmpdir):
    import torch
    from oracle import ref_harness as R
    torch.set_num_threads(max(1, 8 // ws))
    R.ensure_process_group(rank, ws, port)
    R._install_shims()
    from moco.builder_diffspeed_diffloss import Loss
    model = R.build_reference_model(ARCH, K=K)
    spec = R.state_spec(model)
    state, mom, clips, _, _ = G.case_inputs(spec, ARCH, B, HW, K, ws, CHAIN_SEED)
    model.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in state.items()})
    model.train()
    net = torch.nn.parallel.DistributedDataParallel(model, find_unused_parameters=True)      # broadcast_buffers=True (default)
    params = [p for p in model.parameters() if p.requires_grad]
    names = {id(p): n for n, p in model.named_parameters()}
    opt = torch.optim.SGD(params, lr=G.LR, momentum=0.9, dampening=0.0, weight_decay=1e-4, nesterov=False)
    for p in params:
        if names[id(p)] in mom:
            opt.state[p]["momentum_buffer"] = torch.from_numpy(np.array(mom[names[id(p)]]))
    crit = Loss(margin=2.0, A=1.0, M=1.0)
    out: Dict[str, np.ndarray] = {}
    im_q, im_k = torch.from_numpy(clips[rank][0]), torch.from_numpy(clips[rank][1])
    for s in range(STEPS):
        perms_B, sh = chain_perms(s, ws)
        with R._ReplayRNG([perms_B[rank], sh[0], sh[1]], 2):
            o, tgt, rl, rt = net(im_q, im_k)
        loss, loss_A, loss_M = crit(o, tgt, rl, rt.view(-1, 1))
        opt.zero_grad()
        loss.backward()
        opt.step()
        pre = f"r{rank}.s{s}."
        out[pre + "loss"] = loss.detach().numpy().copy()
        out[pre + "logits1"] = o[0].detach().numpy().copy()
        for k, v in model.state_dict().items():
            if k in ("queue", "queue_ptr") or k.endswith("num_batches_tracked"):
                out[pre + "post." + k] = v.detach().numpy().copy()
            elif k.endswith(BUFFER_SUFFIXES[:2]):
                out[pre + "postsum." + k] = P.summarise(k, v.detach().numpy())
        # two trained tensors as well: the chain is free-running, the test follows it through the optimizer
        for k in ("encoder_q.encoder.conv1.weight", "encoder_q.fc1.2.weight"):
            out[pre + "postsum." + k] = P.summarise(k, dict(model.named_parameters())[k].detach().numpy())
    np.savez(os.path.join(tmpdir, f"r{rank}.npz"), **out)


def main():
    import torch.multiprocessing as mp
    from oracle.ref_harness import _free_port
    with tempfile.TemporaryDirectory() as tmp:
        mp.spawn(_worker, args=(WS, _free_port(), tmp), nprocs=WS, join=True)
        out: Dict[str, np.ndarray] = {}
        for r in range(WS):
            with np.load(os.path.join(tmp, f"r{r}.npz")) as z:
                out.update({k: z[k] for k in z.files})
    out["meta"] = np.array(json.dumps(dict(arch=ARCH, fc_type="linear", B=B, HW=HW, K=K, ws=WS, seed=CHAIN_SEED, lr=G.LR, speed=2, steps=STEPS,
                                           T_in=G.T_IN, m=0.999, T=0.07, sgd_momentum=0.9, weight_decay=1e-4, margin=2.0, A=1.0, M=1.0)))
    path = os.path.join(G.GOLDEN, "chain_c3d_ws2.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, "losses", [[float(out[f"r{r}.s{s}.loss"]) for s in range(STEPS)] for r in range(WS)])


if __name__ == "__main__":
    main()
