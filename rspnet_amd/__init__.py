"""rspnet_amd — MI355X-native implementation of RSPNet's pretext-training hot path.

Host side mirrors the reference's Python surface for this path (``moco.ModelFactory``,
``moco.builder_diffspeed_diffloss.{MoCoDiffLossTwoFc, Loss}``, ``moco.split_wrapper.MultiTaskWrapper``,
``models.get_model_class``); all arithmetic runs in ``librspnet_hip.so`` (include/rspnet_hip.h).
"""
__version__ = "0.1.0"
