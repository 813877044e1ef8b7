"""CPU, build container only (skipped where /root/reference is absent, i.e. on the GPU box): run the REAL reference next to
the oracle restatement on fresh seeds -- not the committed fixtures -- so the pin does not rest on files alone.

  * pretext step (C3D, B=4, 32 px, ws=1): reference MoCoDiffLossTwoFc + Loss + torch SGD vs oracle.restatement.moco_step
  * fine-tune forward/backward (C3D): reference MultiTaskWrapper(finetune=True) vs restatement.finetune_step
  * augmentation chains (default and aug_plus): the reference's own transform classes vs oracle.augment
"""
import random

import numpy as np
import pytest
import torch

from oracle import ref_harness as R

pytestmark = pytest.mark.skipif(not R.reference_available(), reason="/root/reference not present (GPU box)")


@pytest.mark.parametrize("B,K,seed", [(4, 64, 41), (5, 60, 42)], ids=["B4", "odd_batch_B5"])
def test_pretext_step_restatement_equals_reference_live(B, K, seed):
    """(B = 5: int(B * alpha) = 2 clips keep their speed and 3 are sub-sampled — the unequal halves of
    builder_diffspeed_diffloss.py:421-431 — and the queue holds 12 batches of 5)"""
    from golden_util import rel_err, run_restatement
    from oracle import gen_golden as G
    arch, HW = "c3d", 32                                              # seeds no fixture uses
    R.ensure_process_group()
    model = R.build_reference_model(arch, K=K)
    spec = R.state_spec(model)
    state, mom, clips, perms_B, sh = G.case_inputs(spec, arch, B, HW, K, 1, seed)
    ref = R.run_reference_step(model, state, clips[0][0], clips[0][1], [perms_B[0], sh[0], sh[1]], 2, lr=G.LR,
                               momentum_buffers=mom)
    meta = dict(arch=arch, fc_type="linear", B=B, HW=HW, K=K, ws=1, seed=seed, lr=G.LR, speed=2, m=0.999, T=0.07,
                sgd_momentum=0.9, weight_decay=1e-4, margin=2.0, A=1.0, M=1.0)
    outs, states, moms = run_restatement(arch, meta, (state, mom, clips, perms_B, sh))
    o = outs[0]
    for k in ("loss", "loss_A", "loss_M", "logits1", "logits2", "l_pos_M", "l_neg_M", "q_A", "q_M"):
        assert rel_err(np.asarray(o[k].detach() if torch.is_tensor(o[k]) else o[k]), ref[k]) <= 1e-5, k
    post = ref["post_state"]
    for k in ("queue", "encoder_q.encoder.conv3a.weight", "encoder_k.encoder.bn5b.running_var", "encoder_q.fc1.2.weight"):
        assert rel_err(states[0][k].numpy(), post[k]) <= 1e-5, k
    assert int(states[0]["queue_ptr"]) == int(post["queue_ptr"])


def test_finetune_restatement_equals_reference_live():
    from oracle import portable as P
    from oracle import restatement as S
    arch, B, HW, ncls, seed = "c3d", 3, 32, 7, 43
    model = R.build_reference_finetune(arch, ncls)
    state = P.fill_state(R.state_spec(model), seed)
    x = P.clips(seed, 0, (B, 3, 16, HW, HW))[0]
    target = np.array([1, 5, 2], dtype=np.int64)
    le, lt, loss, grads, post = R.run_reference_finetune(model, state, x, target)
    sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    assert float((S.finetune_forward(arch, sd, torch.from_numpy(x), training=False) - torch.from_numpy(le)).abs().max()) <= 1e-5
    mlt, mloss, mg = S.finetune_step(arch, sd, torch.from_numpy(x), torch.from_numpy(target))
    assert float((mlt - torch.from_numpy(lt)).abs().max()) <= 1e-5 and abs(float(mloss) - loss) <= 1e-6
    for k, g in grads.items():
        assert (g is None) == (mg[k] is None), k
        if g is not None:
            assert float((mg[k] - torch.from_numpy(g)).abs().max()) <= 1e-5 * max(1.0, float(np.abs(g).max())), k


@pytest.mark.parametrize("plus", [False, True], ids=["default", "aug_plus"])
def test_augmentation_restatement_equals_reference_live(plus):
    from oracle import augment as A
    from oracle import gen_golden_augment as G
    for seed in (101, 102, 103, 104, 105, 106):
        clip = A.synthetic_clip(seed, 3, 27, 35)
        tf = G.reference_gpu_transform_plus(16) if plus else G.reference_gpu_transform(16)
        random.seed(seed)
        ref = tf(clip.clone())
        random.seed(seed)
        prm = A.draw_params_plus() if plus else A.draw_params()
        mine = A.augment_clip(clip, 16, prm, G.MEAN, G.STD)
        assert float((ref - mine).abs().max()) <= 2e-6, (seed, prm)
