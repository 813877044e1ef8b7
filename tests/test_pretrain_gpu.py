"""GPU: the pretrain driver end to end on one MI355X — config → model → epochs → checkpoint → resume."""
import json
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _args(tmp_path, **kw):
    cfg = json.load(open("rspnet_amd/config/pretrain/c3d.json"))
    cfg.update(batch_size=4, num_epochs="2", log_interval=2)
    cfg["moco"]["k"] = 64
    cfg["spatial_transforms"]["size"] = 32
    p = tmp_path / "cfg.json"
    json.dump(cfg, open(p, "w"))
    a = dict(config=str(p), ext_config=None, experiment_dir=str(tmp_path / "exp"), load_checkpoint=None, load_model=None,
             debug=False, world_size=1, seed=0, no_scale_lr=False, steps_per_epoch=3, run_dir=str(tmp_path / "exp" / "run_0_t"),
             cont=False)
    a.update(kw)
    return types.SimpleNamespace(**a)


def test_pretrain_runs_checkpoints_and_resumes(tmp_path):
    from rspnet_amd.pretrain import main_worker
    stats = main_worker(0, _args(tmp_path), "")
    assert stats["loss"] == stats["loss"] and stats["clips_per_s"] > 0
    ck = torch.load(tmp_path / "exp" / "checkpoint.pth.tar", weights_only=False)
    assert set(ck) == {"epoch", "arch", "model", "best_loss", "optimizer", "scheduler"} and ck["epoch"] == 2
    assert (tmp_path / "exp" / "model_best.pth.tar").exists()
    run = tmp_path / "exp" / "run_0_t"                       # run dir layout (framework/arguments.py:60-81)
    assert {"config.json", "run.sh", "experiment.log"} <= {f.name for f in run.iterdir()}
    assert ck["model"]["queue_ptr"].item() == (2 * 3 * 4) % 64
    assert ck["model"]["encoder_q.encoder.bn1.num_batches_tracked"].item() == 6      # q: +1 per step
    assert ck["model"]["encoder_k.encoder.bn1.num_batches_tracked"].item() == 12     # k: +2 per step (two key passes)
    # resume: already at num_epochs -> nothing to do, state restored
    stats2 = main_worker(0, _args(tmp_path, load_checkpoint=str(tmp_path / "exp" / "checkpoint.pth.tar")), "")
    assert stats2 is None
    # arch mismatch is refused like pretrain.py:112-116
    ck["arch"] = "resnet18"
    torch.save(ck, tmp_path / "bad.pth.tar")
    with pytest.raises(ValueError):
        main_worker(0, _args(tmp_path, load_model=str(tmp_path / "bad.pth.tar")), "")


def test_pretrain_with_uint8_loader_and_fused_augmentation(tmp_path):
    """uint8 clips -> CPU random crop -> FusedGPUCollateFn -> pretext step: the data path of SURVEY.md §8f-2 feeding §8a."""
    from rspnet_amd.pretrain import main_worker
    stats = main_worker(0, _args(tmp_path, loader="uint8"), "")
    assert stats["loss"] == stats["loss"] and 0 < stats["loss"] < 50 and stats["clips_per_s"] > 0


def test_launcher_counts_gpus_without_touching_the_runtime():
    """pretrain.py's parent decides the world size from sysfs / *_VISIBLE_DEVICES (the launcher must stay GPU-free); on this box
    it must agree with what the HIP runtime reports."""
    import torch
    from rspnet_amd.pretrain import visible_gpu_count
    assert visible_gpu_count() == torch.cuda.device_count() >= 1
