"""Pretext-training driver for rspnet_amd — the `pretrain.py` surface of the reference (config keys, CLI flag names,
checkpoint layout, LR policy) around the MI355X-native step.

Mirrors /root/reference/pretrain.py:31-336 for what touches the hot path: Engine construction (:33-110: model factory,
Loss(margin=2.0, A, M), LR scaling, SGD, CosineAnnealingLR per epoch with eta_min = lr/1000), checkpoint load with the arch
check (:112-132), the train loop (:147-197: forward, loss, zero_grad/backward/step, top-k accuracies), epoch loop and
checkpoint dict (:220-260), one process per GPU started with mp.spawn and a tcp://127.0.0.1 rendezvous (:263-336).
The data pipeline (decord decode + GPU augmentation) is out of scope: the loader here is any iterable
of (clip_q, clip_k) device tensors; `SyntheticClips` stands in for it.  The config is the resolved JSON the reference
saves as run_*/config.json (rspnet_amd/config/pretrain/*.json ship the four shipped pretext configs).
"""
from __future__ import annotations

import argparse
import json
import logging
import math
import os
import re
import socket
import sys
import time
from datetime import datetime
from pathlib import Path
from shlex import quote

import torch
import torch.distributed as dist

from .framework.utils.checkpoint import CheckpointManager
from .framework.utils.environment import scale_learning_rate
from .moco import Loss, ModelFactory
from .optim import SGD
from .utils.moco import replace_moco_k_in_config

logger = logging.getLogger(__name__)


def accuracy(output: torch.Tensor, target: torch.Tensor, topk=(1,)):
    """Top-k hit rate x100 (framework/metrics/classification.py:6-20); stays on device (no sync)."""
    maxk = max(topk)
    _, pred = output.topk(maxk, 1, True, True)
    correct = pred.t().eq(target.view(1, -1).expand_as(pred.t()))
    return [correct[:k].reshape(-1).float().sum(0) * (100.0 / target.size(0)) for k in topk]


class SyntheticClips:
    """Stand-in data loader: `steps` batches of N(0,1) clip pairs (B,3,T,H,W), generated once on the device."""

    def __init__(self, batch_size, T, size, steps, device, seed=1234):
        g = torch.Generator(device=device).manual_seed(seed)
        self.q = torch.randn(batch_size, 3, T, size, size, device=device, generator=g)
        self.k = torch.randn(batch_size, 3, T, size, size, device=device, generator=g)
        self.steps = steps

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            yield self.q, self.k


class SyntheticVideoClips:
    """Decode-free stand-in for the reference's video pipeline that still exercises its GPU half: every step yields B samples
    of two uint8 (T, h, w, 3) crops -- a fixed pool of synthetic "videos", a RawVideoRandomCrop window (scale 0.4..1, aspect
    3/4..4/3; transforms_spatial.py:43-80) drawn per clip on the CPU as the reference does -- and hands them to
    rspnet_amd.augment.FusedGPUCollateFn (ToTensor, Resize, grayscale, colour jitter, flip, normalise in one HIP launch group),
    i.e. the loader returns (clip_q, clip_k) device tensors exactly like DataLoader(collate_fn=SequentialGPUCollateFn(...))."""

    def __init__(self, batch_size, T, size, steps, device, seed=1234, src_hw=(128, 171), mean=(0.485, 0.456, 0.406),
                 std=(0.229, 0.224, 0.225), aug_plus=False, pool=8):
        import random as _random
        from .augment import FusedGPUCollateFn
        g = torch.Generator().manual_seed(seed)
        self.videos = [torch.randint(0, 256, (T, src_hw[0], src_hw[1], 3), dtype=torch.uint8, generator=g) for _ in range(pool)]
        self.collate = FusedGPUCollateFn(size, mean, std, target_transform=False, device=device, aug_plus=aug_plus)
        self.batch_size, self.steps = batch_size, steps
        self.rng = _random.Random(seed)

    def _crop(self, clip):
        H, W = clip.shape[1], clip.shape[2]
        for _ in range(10):
            area = self.rng.uniform(0.4, 1.0) * H * W
            ar = math.exp(self.rng.uniform(math.log(3 / 4), math.log(4 / 3)))
            w, h = int(round(math.sqrt(area * ar))), int(round(math.sqrt(area / ar)))
            if 0 < w <= W and 0 < h <= H:
                i, j = self.rng.randint(0, H - h), self.rng.randint(0, W - w)
                return clip[:, i:i + h, j:j + w, :].contiguous()
        return clip

    def __len__(self):
        return self.steps

    def __iter__(self):
        for _ in range(self.steps):
            batch = []
            for _b in range(self.batch_size):
                v = self.videos[self.rng.randrange(len(self.videos))]
                batch.append(([self._crop(v), self._crop(v)], 0))
            (clip_q, clip_k), _ = self.collate(batch)
            yield clip_q, clip_k


class Engine:
    def __init__(self, args, cfg: dict, local_rank: int, train_loader=None):
        self.args, self.cfg, self.local_rank = args, cfg, local_rank
        self._stepper, self._first_epoch = None, 0
        self.device = torch.device("cuda", local_rank)
        self.model = ModelFactory(cfg).build_moco_diffloss(device=self.device)
        self.criterion = Loss(margin=2.0, A=float(cfg["loss_lambda"]["A"]), M=float(cfg["loss_lambda"]["M"]))
        self.batch_size = int(cfg["batch_size"])
        self.learning_rate = float(cfg["optimizer"]["lr"])
        if not args.no_scale_lr:
            self.learning_rate = scale_learning_rate(self.learning_rate, args.world_size, self.batch_size)
        o = cfg["optimizer"]
        # all parameters, frozen encoder_k included, exactly as pretrain.py:65-72 does: the checkpoint's 'optimizer' entry
        # (param_groups[0]['params'] indices) is then interchangeable with the reference's
        self.optimizer = SGD(self.model.parameters(), lr=self.learning_rate,
                             momentum=float(o["momentum"]), dampening=float(o["dampening"]),
                             weight_decay=float(o["weight_decay"]), nesterov=bool(o["nesterov"]))
        self.num_epochs = int(cfg["num_epochs"])          # the jsonnet value is the STRING '200' (moco-train-base.jsonnet:18)
        self.scheduler = torch.optim.lr_scheduler.CosineAnnealingLR(self.optimizer, T_max=self.num_epochs,
                                                                    eta_min=self.learning_rate / 1000)
        self.arch = cfg["arch"]
        self.checkpoint = (CheckpointManager(args.experiment_dir, keep_interval=int(cfg["checkpoint_interval"]))
                           if local_rank == 0 else None)
        self.log_interval = int(cfg["log_interval"])
        self.current_epoch = 0
        self.best_loss = math.inf
        T, size = int(cfg["temporal_transforms"]["size"]), int(cfg["spatial_transforms"]["size"])
        if train_loader is None and getattr(args, "loader", "tensor") == "uint8":
            train_loader = SyntheticVideoClips(self.batch_size, T, size, args.steps_per_epoch, self.device,
                                               seed=args.seed + local_rank, aug_plus=bool(cfg.get("moco", {}).get("aug_plus", False)))
        self.train_loader = train_loader or SyntheticClips(self.batch_size, T, size, args.steps_per_epoch, self.device,
                                                           seed=args.seed + local_rank)

    # ---- checkpoints (pretrain.py:112-132) ---------------------------------------------------------------------------
    def _load_ckpt_file(self, path):
        states = torch.load(path, map_location=self.device, weights_only=False)
        if states["arch"] != self.arch:
            raise ValueError(f'Loading checkpoint arch {states["arch"]} does not match current arch {self.arch}')
        return states

    def load_checkpoint(self, path):
        states = self._load_ckpt_file(path)
        self.model.module.load_state_dict(states["model"])
        self.optimizer.load_state_dict(states["optimizer"])
        self.scheduler.load_state_dict(states["scheduler"])
        self.current_epoch = states["epoch"]
        self.best_loss = states["best_loss"]

    def load_model(self, path):
        self.model.module.load_state_dict(self._load_ckpt_file(path)["model"])

    # ---- training (pretrain.py:147-260) ------------------------------------------------------------------------------
    def train_epoch(self):
        sums = torch.zeros(4, device=self.device)
        n = 0
        t0 = time.perf_counter()
        if self._stepper is None and self.device.type == "cuda" and not getattr(self.args, "no_graph", False):
            # the five statements below run through the stepper — replayed as linear HIP graphs on three streams, with the
            # data-parallel collectives between them at more than one rank, where the host cannot issue the step fast enough; the
            # eager loop with its side streams otherwise (rspnet_amd/graph_step.py)
            from .graph_step import GraphedPretextStep
            self._stepper = GraphedPretextStep(self.model, self.criterion, self.optimizer)
        for it, (clip_q, clip_k) in enumerate(self.train_loader):
            if it == 2 and self.current_epoch == self._first_epoch:
                # modules, layer plans and descriptor caches are long-lived: park them in the permanent generation so a full
                # garbage collection cannot pause the host for tens of ms while the GPU runs dry (measured: 40-75 ms, bench.py)
                import gc
                gc.collect()
                gc.freeze()
            if self._stepper is not None:
                loss, loss_A, loss_M, output, ranking_logits = self._stepper(clip_q, clip_k)
                target = torch.zeros(output[0].shape[0], dtype=torch.long, device=output[0].device)   # labels_A (:540)
            else:
                output, target, ranking_logits, ranking_target = self.model(clip_q, clip_k)
                loss, loss_A, loss_M = self.criterion(output, target, ranking_logits, ranking_target)
                self.optimizer.zero_grad()
                loss.backward()
                self.optimizer.step()
            acc1_A, acc5_A = accuracy(output[0], target, topk=(1, 5))
            acc1_M, = accuracy(torch.cat(ranking_logits, dim=1), target, topk=(1,))
            sums += torch.stack([loss.detach(), loss_A, loss_M, acc1_A])
            n += 1
            if self.local_rank == 0 and (it + 1) % self.log_interval == 0:    # the only host sync (pretrain.py:177-185)
                m = (sums / n).tolist()
                logger.info("epoch %d it %d loss %.4f loss_A %.4f loss_M %.4f acc1_A %.2f acc1_M %.2f", self.current_epoch,
                            it + 1, m[0], m[1], m[2], m[3], float(acc1_M))
        torch.cuda.synchronize(self.device)
        dt = time.perf_counter() - t0
        mean = (sums / max(n, 1)).tolist()
        return {"loss": mean[0], "loss_A": mean[1], "loss_M": mean[2], "acc1_A": mean[3],
                "clips_per_s": n * self.batch_size * self.args.world_size / dt}

    def run(self):
        num_epochs = 1 if self.args.debug else self.num_epochs
        self._first_epoch = self.current_epoch
        self.model.train()
        stats = None
        while self.current_epoch < num_epochs:
            stats = self.train_epoch()
            self.scheduler.step()
            self.current_epoch += 1
            self.model.sync_buffers()
            if self.local_rank == 0:
                is_best = stats["loss"] < self.best_loss
                self.best_loss = min(self.best_loss, stats["loss"])
                self.checkpoint.save({"epoch": self.current_epoch, "arch": self.arch,
                                      "model": self.model.module.state_dict(), "best_loss": self.best_loss,
                                      "optimizer": self.optimizer.state_dict(), "scheduler": self.scheduler.state_dict()},
                                     is_best, self.current_epoch)
                logger.info("epoch %d done: %s", self.current_epoch, json.dumps(stats))
        return stats


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


RUN_DIR_NAME_REGEX = re.compile(r"^run_(\d+)_")


def resolve_run_dir(args) -> Path:
    """EXP/run_{id}_{timestamp}: id = 1 + the highest existing run id (framework/arguments.py:64-78)."""
    if args.run_dir is not None:
        return Path(args.run_dir)
    exp = Path(args.experiment_dir)
    run_id = -1
    if exp.exists():
        for prev in exp.iterdir():
            m = RUN_DIR_NAME_REGEX.match(prev.name)
            if m is not None:
                run_id = max(run_id, int(m.group(1)))
    return exp / f"run_{run_id + 1}_{datetime.now().strftime('%Y%m%d_%H%M%S')}"


def resolve_continue(args):
    """--continue: newest run's config.json and EXP/checkpoint.pth.tar (arguments.py:59-86)."""
    if not args.cont:
        return
    exp = Path(args.experiment_dir)
    if not exp.exists():
        raise EnvironmentError(f'Experiment directory "{exp}" does not exists.')
    if args.config is None:
        best = -1
        for run in exp.iterdir():
            m = RUN_DIR_NAME_REGEX.match(run.name)
            if m is not None and int(m.group(1)) > best and run.is_dir() and (run / "config.json").exists():
                best = int(m.group(1))
                args.config = str(run / "config.json")
        if args.config is None:
            raise EnvironmentError("No previous run config found")
        logger.info('Continue using previous config: "%s"', args.config)
    if args.load_checkpoint is None:
        ckpt = exp / "checkpoint.pth.tar"
        if ckpt.exists():
            args.load_checkpoint = str(ckpt)
            logger.info('Continue using previous checkpoint: "%s"', ckpt)
        else:
            logger.warning("No previous checkpoint found")


def save_run_files(args, cfg: dict):
    """run dir contents: config.json (framework/config.py:78-81), run.sh (framework/arguments.py:49-58), experiment.log."""
    run_dir = Path(args.run_dir)
    run_dir.mkdir(parents=True, exist_ok=True)
    with open(run_dir / "config.json", "w") as f:
        json.dump(cfg, f, indent=2)
    with open(run_dir / "run.sh", "w") as f:
        f.write(f"cd {quote(os.getcwd())}\n")
        for env in ("HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
            if os.environ.get(env) is not None:
                f.write(f"export {env}={quote(os.environ[env])}\n")
        f.write(sys.executable + " " + " ".join(quote(a) for a in sys.argv) + "\n")


def main_worker(local_rank: int, args, dist_url: str):
    logging.basicConfig(level=logging.DEBUG if args.debug else logging.INFO, format="%(asctime)s %(message)s")
    if local_rank == 0 and args.run_dir is not None:
        Path(args.run_dir).mkdir(parents=True, exist_ok=True)
        logging.getLogger().addHandler(logging.FileHandler(Path(args.run_dir) / "experiment.log"))   # framework/logging.py:31
    import random
    import numpy as np
    seed = args.seed + local_rank                                   # utils/reproduction.py initialize_seed (pretrain.py:266-267)
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    torch.cuda.set_device(local_rank)
    forced = args.world_size <= 1 and bool(os.environ.get("RSP_FORCE_COLLECTIVES"))      # one rank, RCCL path on (see MoCoDiffLossTwoFc)
    if args.world_size > 1 or forced:
        dist.init_process_group("nccl", init_method=dist_url or f"tcp://127.0.0.1:{_free_port()}", rank=local_rank,
                                world_size=max(args.world_size, 1), device_id=torch.device("cuda", local_rank))
    with open(args.config) as f:
        cfg = json.load(f)
    for snippet in args.ext_config or []:                           # -x overlays: JSON objects merged on top
        _merge(cfg, json.loads(snippet))
    replace_moco_k_in_config(cfg)
    if local_rank == 0:
        Path(args.experiment_dir).mkdir(parents=True, exist_ok=True)
        save_run_files(args, cfg)
    engine = Engine(args, cfg, local_rank)
    if args.load_model is not None:
        engine.load_model(args.load_model)
    if args.load_checkpoint is not None:
        engine.load_checkpoint(args.load_checkpoint)
    stats = engine.run()
    if args.world_size > 1 or forced:
        dist.barrier()
        dist.destroy_process_group()
    return stats


def _merge(base: dict, over: dict):
    for k, v in over.items():
        if isinstance(v, dict) and isinstance(base.get(k), dict):
            _merge(base[k], v)
        else:
            base[k] = v


def parse_args(argv=None):
    ap = argparse.ArgumentParser(description="RSPNet pretext training on MI355X (flag names follow the reference's arguments.py)")
    ap.add_argument("-c", "--config", default=None, help="resolved config JSON (e.g. rspnet_amd/config/pretrain/c3d.json)")
    ap.add_argument("-x", "--ext-config", action="append", help="JSON object merged over the config (may repeat)")
    ap.add_argument("-e", "--experiment-dir", required=True)
    ap.add_argument("--load-checkpoint", default=None)
    ap.add_argument("--load-model", default=None)
    ap.add_argument("-d", "--debug", action="store_true", help="1 epoch, DEBUG logging, allows --ws 1 (pretrain.py:312-316)")
    ap.add_argument("--ws", "--world-size", dest="world_size", type=int, default=None)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--no-scale-lr", action="store_true")
    ap.add_argument("--steps-per-epoch", type=int, default=100, help="synthetic loader length")
    ap.add_argument("--no-graph", action="store_true", help="never replay the step as captured HIP graphs (default: only when the host is the limiter)")
    ap.add_argument("--loader", choices=("tensor", "uint8"), default="tensor",
                    help="tensor: fixed N(0,1) device clips; uint8: synthetic uint8 videos -> CPU random crop -> fused GPU augmentation")
    ap.add_argument("--run-dir", default=None, help="default: EXP/run_{id}_{timestamp}")
    ap.add_argument("--continue", dest="cont", action="store_true", help="use the previous run's config and EXP/checkpoint.pth.tar")
    args = ap.parse_args(argv)
    resolve_continue(args)
    if args.config is None:
        ap.error("-c/--config is required (or --continue with a previous run)")
    args.run_dir = str(resolve_run_dir(args))     # resolved once, before the workers are spawned
    if args.world_size is None:
        args.world_size = visible_gpu_count()
    return args


def visible_gpu_count() -> int:
    """Number of GPUs this launcher spawns ranks for — what the reference asks torch.cuda.device_count() for (pretrain.py:318) —
    found WITHOUT touching the HIP runtime in the parent: ranks are fresh child processes and the launcher itself must stay
    GPU-free (a process that has initialised the GPU must never be re-exec'ed or forked on this platform).  A short-lived CHILD
    interpreter is asked for torch.cuda.device_count(): it sees exactly what a rank will see (HIP_/ROCR_/CUDA_VISIBLE_DEVICES,
    container device filtering).  Only if that child cannot be run, the count falls back to the KFD topology intersected with the
    *_VISIBLE_DEVICES lists.  Raises when no GPU is visible."""
    import subprocess
    n = None
    try:
        r = subprocess.run([sys.executable, "-c", "import torch; print(torch.cuda.device_count())"], capture_output=True, text=True,
                           timeout=300)
        if r.returncode == 0:
            n = int(r.stdout.strip().splitlines()[-1])
    except (OSError, ValueError, IndexError, subprocess.TimeoutExpired):
        n = None
    if n is None:
        n = _kfd_gpu_count()
        for var in ("ROCR_VISIBLE_DEVICES", "HIP_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES"):      # each list filters the previous one
            v = os.environ.get(var)
            if v is not None:
                n = min(n, len([t for t in v.split(",") if t.strip() != ""]))
    if n <= 0:
        raise EnvironmentError("rspnet_amd.pretrain: no GPU is visible to this process (check HIP_VISIBLE_DEVICES / "
                               "ROCR_VISIBLE_DEVICES and the container's /dev/kfd, /dev/dri access); pass --ws to override")
    return n


def _kfd_gpu_count() -> int:
    """GPUs in the KFD topology (nodes with simd_count > 0)."""
    import glob
    n = 0
    for path in glob.glob("/sys/class/kfd/kfd/topology/nodes/*/properties"):
        try:
            with open(path) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
            if int(props.get("simd_count", "0")) > 0:
                n += 1
        except (OSError, ValueError):
            continue
    return n


def main(argv=None):
    args = parse_args(argv)
    # Unlike the reference (which needs >= 2 ranks for shuffle-BN unless --debug), one GPU is a supported configuration.
    if args.world_size <= 1:
        return main_worker(0, args, "")
    url = f"tcp://127.0.0.1:{_free_port()}"
    torch.multiprocessing.spawn(main_worker, args=(args, url), nprocs=args.world_size)


if __name__ == "__main__":
    main()
