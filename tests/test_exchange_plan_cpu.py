"""CPU: the shuffle-BN exchange plan at 1 / 2 / 4 / 8 ranks (the driver's 8-GPU run is the first execution of the RCCL path: the host
arithmetic that sizes its all-to-all must be right at every world size, not only at the 2 ranks the gloo tests run).

The reference (builder_diffspeed_diffloss.py:361-406) all-gathers the clips, rank r keeps x_gather[idx_shuffle.view(ws, -1)[r]],
runs encoder_k, all-gathers the features and restores the order with argsort(idx_shuffle).  The product sends every clip to its
ONE destination (`MoCoDiffLossTwoFc._exchange_plan`: send order, all-to-all splits) and un-shuffles the gathered features with
`loc`.  Simulated here with numpy: a clip's "pixels" and its "feature" are its global sample id."""
import numpy as np
import pytest

from rspnet_amd.moco.builder_diffspeed_diffloss import MoCoDiffLossTwoFc

plan = MoCoDiffLossTwoFc._exchange_plan


@pytest.mark.parametrize("ws", [1, 2, 4, 8])
@pytest.mark.parametrize("B", [4, 32])
def test_exchange_plan_reproduces_the_reference_shuffle(ws, B):
    rng = np.random.default_rng(100 * ws + B)
    for _ in range(5):
        idx = rng.permutation(B * ws).astype(np.int64)
        G = idx.reshape(ws, B)                                   # the reference's batch of rank r: global samples G[r]
        plans = [plan(idx, B, r, ws) for r in range(ws)]
        # what every rank sends: its own clips (global id = rank * B + local index) in send order, cut by in_splits
        received = []
        for r in range(ws):
            got = []
            for s in range(ws):
                send_src, _, in_splits, out_splits, _ = plans[s]
                if ws == 1:
                    seg = send_src
                else:
                    start = int(np.sum(in_splits[:r]))
                    seg = send_src[start:start + in_splits[r]]
                    assert plans[r][3][s] == in_splits[r]          # receiver's out_split from s == sender's in_split for r
                got.append(seg.astype(np.int64) + s * B)
            received.append(np.concatenate(got))
        for r in range(ws):
            send_src, loc, in_splits, out_splits, arrival = plans[r]
            assert len(received[r]) == B and sorted(received[r]) == sorted(G[r])      # exactly the reference's batch, each clip once
            # arrival[j] = position the j-th arriving clip has in the reference's shuffled batch G[r]
            assert np.array_equal(received[r], G[r][arrival])
            if ws > 1:
                assert sum(in_splits) == B and sum(out_splits) == B
        # features: rank r computes feat[j] = id of its j-th arriving clip; all-gather concatenates ranks; `loc` restores global order
        gathered = np.concatenate(received)
        for r in range(ws):
            loc = plans[r][1]
            assert np.array_equal(gathered[loc], np.arange(B * ws))                 # all_feats[g] is sample g's feature
        # the reference's un-shuffle gives rank r the features of its own samples r*B .. r*B+B-1: the slice the product takes
        ref_unshuffle = np.argsort(idx)
        ref_gathered = idx                                                          # reference: feature j of the gathered batch = idx[j]
        for r in range(ws):
            assert np.array_equal(ref_gathered[ref_unshuffle.reshape(ws, B)[r]], np.arange(r * B, (r + 1) * B))
