"""Fine-tune / validation step helpers around ``MultiTaskWrapper(finetune=True)`` — the parts of the reference's finetune.py
that touch the model: multi-crop reshape + logit averaging (finetune.py:44-61), the pretext-checkpoint loader
with its prefix / blacklist rule (:273-310), and the train / validate steps (:95-116, :326-345).  The data pipeline, meters and
TensorBoard of finetune.py are not rebuilt."""
from __future__ import annotations

import logging
from typing import Dict

import torch
from torch import Tensor, nn

logger = logging.getLogger(__name__)
BLACKLIST = ("fc.", "linear", "head", "new_fc", "fc8", "encoder_fuse")


def reshape_clip(clip: Tensor, n_crop: int) -> Tensor:
    """(B, C, n_crop*T, H, W) -> (B*n_crop, C, T, H, W), crops of one sample adjacent (finetune.py:44-52)."""
    if n_crop == 1:
        return clip
    B, C, TT, H, W = clip.shape
    T = TT // n_crop
    return clip.view(B, C, n_crop, T, H, W).permute(0, 2, 1, 3, 4, 5).reshape(B * n_crop, C, T, H, W)


def average_logits(logits: Tensor, n_crop: int) -> Tensor:
    """(B*n_crop, classes) -> (B, classes), mean over the crops of a sample (finetune.py:54-61)."""
    if n_crop == 1:
        return logits
    return logits.view(logits.shape[0] // n_crop, n_crop, -1).mean(dim=1)


def load_moco_checkpoint(model: nn.Module, checkpoint_path: str, device=None) -> "torch.nn.modules.module._IncompatibleKeys":
    """finetune.py:273-310: keep ``encoder_q.*`` of a pretext checkpoint (or ``module.*`` / bare keys of a third-party one),
    drop the classifier-like names, load non-strictly.  ``model`` is the unwrapped MultiTaskWrapper."""
    cp = torch.load(checkpoint_path, map_location=device, weights_only=False)
    if "model" in cp and "arch" in cp:
        state, prefix = cp["model"], "encoder_q."
    else:
        state = cp["state_dict"] if "state_dict" in cp else cp
        prefix = "module." if next(iter(state.keys())).startswith("module") else ""
    keep = {k[len(prefix):]: v for k, v in state.items()
            if k.startswith(prefix) and not any(k.startswith(f"{prefix}{b}") for b in BLACKLIST)}
    msg = model.load_state_dict(keep, strict=False)
    logger.warning("Missing keys: %s, Unexpected keys: %s", msg.missing_keys, msg.unexpected_keys)
    return msg


def train_step(model: nn.Module, criterion: nn.Module, optimizer: torch.optim.Optimizer, clip: Tensor, target: Tensor) -> Dict:
    """One optimisation step as finetune.py:326-338 runs it (n_crop = 1 in training)."""
    output = model(clip)
    loss = criterion(output, target)
    optimizer.zero_grad()
    loss.backward()
    optimizer.step()
    return {"loss": loss.detach(), "output": output.detach()}


@torch.no_grad()
def validate_step(model: nn.Module, criterion: nn.Module, clip: Tensor, target: Tensor, n_crop: int = 1) -> Dict:
    """finetune.py:95-116 under ``model.eval()``: crops -> logits -> mean over crops -> loss."""
    output = average_logits(model(reshape_clip(clip, n_crop)), n_crop)
    return {"loss": criterion(output, target), "output": output}
