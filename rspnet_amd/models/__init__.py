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
