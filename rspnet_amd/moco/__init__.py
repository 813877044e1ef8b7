"""ModelFactory for the pretext model — same config keys as /root/reference/moco/__init__.py:14-55."""
import torch
from torch import distributed as dist
from torch import nn

from ..models import get_model_class
from .builder_diffspeed_diffloss import Loss, MoCoDiffLossTwoFc
from .split_wrapper import MultiTaskWrapper


def _get(cfg, dotted, default=None):
    """Read 'a.b' from a pyhocon ConfigTree (get) or a plain nested dict (the build ships resolved JSON)."""
    if hasattr(cfg, "get_config") or hasattr(cfg, "get_string"):
        return cfg.get(dotted, default)
    cur = cfg
    for part in dotted.split("."):
        if not isinstance(cur, dict) or part not in cur:
            return default
        cur = cur[part]
    return cur


class DataParallelPretext(nn.Module):
    """Stands where the reference wraps the model in DistributedDataParallel (moco/__init__.py:49-53): exposes
    `.module`, forwards calls, and makes rank 0's initial parameters/buffers global (DDP constructor broadcast).
    Gradient averaging happens inside the model's own backward (bucketed RCCL all-reduce over the flat gradient
    buffer).

    Buffers: torch's DistributedDataParallel (default broadcast_buffers=True, which the reference keeps) re-broadcasts rank 0's
    buffers (BN running statistics, num_batches_tracked, queue, queue_ptr) before EVERY forward.  Here the queue / queue_ptr /
    num_batches_tracked are identical on all ranks by construction (every rank enqueues the same all-gathered keys), and the BN
    running statistics — views into one flat buffer — are broadcast from rank 0 at the top of every forward, one small collective
    (`MoCoDiffLossTwoFc._broadcast_running_stats`): every rank's `state_dict()` follows the one it has under DDP step by step
    (tests/test_distributed_cpu.py, three chained steps against the reference under 2-rank DDP).  `broadcast_buffers=False` leaves
    the statistics rank-local between `sync_buffers()` calls, as DDP(broadcast_buffers=False) would: train-mode numerics never read
    them and rank 0's checkpoint is the same."""

    def __init__(self, module: MoCoDiffLossTwoFc, broadcast_buffers: bool = True):
        super().__init__()
        self.module = module
        module.broadcast_buffers = bool(broadcast_buffers)
        if module._dp()[2]:
            module._prepare()
            with torch.no_grad():
                dist.broadcast(module._flat.q_flat, src=0)
                dist.broadcast(module._flat.k_flat, src=0)
                for b in module.buffers():
                    dist.broadcast(b, src=0)
            module._state_loaded()
            module.setup_side_group()          # collective: every rank constructs the wrapper, as with DDP

    @torch.no_grad()
    def sync_buffers(self):
        """Broadcast rank 0's buffers (what DDP's broadcast_buffers does before each forward)."""
        if self.module._dp()[2]:
            for b in self.module.buffers():
                dist.broadcast(b, src=0)
            self.module._ptr_host = None

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)


class ModelFactory:
    def __init__(self, cfg):
        self.cfg = cfg

    def build_moco_diffloss(self, device=None, force_collectives=None):
        """force_collectives: see MoCoDiffLossTwoFc (run the data-parallel collectives in a world of one rank)."""
        moco_dim = int(_get(self.cfg, "moco.dim"))
        moco_t = float(_get(self.cfg, "moco.t"))
        moco_k = int(_get(self.cfg, "moco.k"))
        moco_m = float(_get(self.cfg, "moco.m"))
        moco_fc_type = str(_get(self.cfg, "moco.fc_type"))
        moco_diff_speed = list(_get(self.cfg, "moco.diff_speed"))
        base_model_class = get_model_class(**dict(_get(self.cfg, "model")))

        def model_class(num_classes=128):
            return MultiTaskWrapper(base_model_class, num_classes=num_classes, fc_type=moco_fc_type, finetune=False,
                                    groups=1)

        model = MoCoDiffLossTwoFc(model_class, dim=moco_dim, K=moco_k, m=moco_m, T=moco_t, diff_speed=moco_diff_speed,
                                  force_collectives=force_collectives)
        if device is None:
            device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else None
        if device is not None:
            model.to(device)
        return DataParallelPretext(model)
