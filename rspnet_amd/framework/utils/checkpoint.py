"""Checkpoint files of the pretext run — same names and atomic-rename protocol as
/root/reference/framework/utils/checkpoint.py:13-62: ``checkpoint.pth.tar`` written through a ``.next.`` temp file, hard
links ``model_best.pth.tar`` and ``checkpoint_epoch_{N}.pth.tar`` (every keep_interval epochs past `milestone`)."""
import os
from pathlib import Path

import torch


class CheckpointManager:
    def __init__(self, experiment_dir, keep_interval=None, filename="checkpoint.pth.tar", milestone=0):
        self.experiment_dir = Path(experiment_dir)
        self.filename = filename
        self.keep_interval = keep_interval
        self.milestone = milestone

    def save(self, state: dict, is_best: bool, epoch: int):
        path = self.experiment_dir / self.filename
        tmp = self.experiment_dir / f".next.{self.filename}"
        try:
            torch.save(state, tmp)          # a failed save must leave the previous checkpoint intact
        except BaseException:
            if tmp.exists():
                tmp.unlink()
            raise
        tmp.rename(path)
        if is_best:
            best = self.experiment_dir / "model_best.pth.tar"
            if best.exists():
                best.unlink()
            os.link(path, best)
        if self.keep_interval is not None and epoch % self.keep_interval == 0 and epoch > self.milestone:
            keep = self.experiment_dir / f"checkpoint_epoch_{epoch}.pth.tar"
            if keep.exists():
                keep.unlink()
            os.link(path, keep)
