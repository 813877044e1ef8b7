"""CPU: the fine-tune restatement (oracle/restatement.py finetune_forward / finetune_step) against the fixtures generated
from the reference's MultiTaskWrapper(finetune=True) (oracle/gen_golden_finetune.py)."""
import numpy as np
import pytest
import torch

from finetune_util import ARCHS, load
from golden_util import rel_err, summary_err
from oracle import restatement as S


@pytest.mark.parametrize("arch", ARCHS)
def test_restatement_matches_reference_fixture(arch):
    z, meta, spec, state, x = load(arch)
    sd = {k: torch.from_numpy(v.copy()) for k, v in state.items()}
    xt = torch.from_numpy(x)
    le = S.finetune_forward(arch, sd, xt, training=False)
    assert rel_err(le.numpy(), z["logits_eval"]) <= 1e-5
    lt, loss, grads = S.finetune_step(arch, sd, xt, torch.from_numpy(z["target"]))
    assert rel_err(lt.numpy(), z["logits"]) <= 1e-5 and abs(float(loss) - float(z["loss"])) <= 1e-5
    for k, g in grads.items():
        ref = z["gradsum." + k]
        assert (g is None) == (ref.size == 0), k
        if g is not None:
            assert summary_err(k, g.numpy(), ref) <= 1e-4, k
