"""Backbone registry — same entry point and arch names as /root/reference/models/__init__.py:16-75."""
from typing import Callable

from torch import nn


def get_model_class(**kwargs) -> Callable[[int], nn.Module]:
    """``model_class(num_classes=N)`` factory for ``cfg['model']``; raises ValueError for unknown archs
    (models/__init__.py:73)."""
    arch = kwargs.get("arch")
    if arch == "c3d":
        from .c3d import C3D
        return C3D
    if arch in ("resnet18", "resnet34", "resnet50"):
        from . import resnet
        return getattr(resnet, arch)
    if arch == "s3dg":
        from .s3dg import S3D_G
        return S3D_G
    if arch == "r2plus1d-vcop":
        from .r2plus1d_vcop import R2Plus1DNet
        return lambda num_classes=128: R2Plus1DNet((1, 1, 1, 1), with_classifier=True, num_classes=num_classes)
    raise ValueError(f'Unknown model architecture "{arch}"')


class ModelFactory:
    """Downstream (fine-tune) model factory — same entry points as /root/reference/models/__init__.py:76-143.
    ``cfg`` is the resolved config (dict or pyhocon ConfigTree): keys ``model.arch``, ``dataset.num_classes``, ``only_train_fc``."""

    def __init__(self, cfg):
        self.cfg = cfg

    def _get(self, dotted, default=None):
        node = self.cfg
        for part in dotted.split("."):
            if not hasattr(node, "get") or part not in node:
                return default
            node = node[part]
        return node

    def _post_process_model(self, model: nn.Module):
        """``only_train_fc``: freeze everything but the classifier and keep the backbone in eval mode (:82-104)."""
        if self._get("only_train_fc", False):
            for param in model.parameters():
                param.requires_grad = False
            fc_module = next((getattr(model, n) for n in ("fc", "new_fc") if hasattr(model, n)), None)
            if fc_module is None:
                raise Exception('"only_train_fc" specified, but no fc layer found')
            for param in fc_module.parameters():
                param.requires_grad = True
            orig_train = model.train

            def override_train(mode=True):
                orig_train(mode=False)
                fc_module.train(mode)

            model.train = override_train
        return model

    def build_multitask_wrapper(self, local_rank: int) -> nn.Module:
        """MultiTaskWrapper(finetune=True) on cuda:local_rank, wrapped in DistributedDataParallel when a process group is up
        (:125-143); with a single process it is wrapped in a pass-through holder so that ``model.module`` exists either way."""
        import torch
        import torch.distributed as dist
        from ..moco.split_wrapper import MultiTaskWrapper
        model_cfg = dict(self._get("model"))
        model = MultiTaskWrapper(get_model_class(**model_cfg), num_classes=int(self._get("dataset.num_classes")), finetune=True)
        model = self._post_process_model(model)
        model = model.to(torch.device("cuda", local_rank))
        if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
            return nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], find_unused_parameters=True)
        return _SingleProcess(model)


class _SingleProcess(nn.Module):
    """What DistributedDataParallel is to one rank: forwards to ``.module`` (checkpoints are saved from ``.module``)."""

    def __init__(self, module: nn.Module):
        super().__init__()
        self.module = module

    def forward(self, *args, **kwargs):
        return self.module(*args, **kwargs)

    def train(self, mode: bool = True):
        self.training = mode
        self.module.train(mode)
        return self
