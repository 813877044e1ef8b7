"""Learning-rate scaling — /root/reference/framework/utils/environment.py:13-16 (lr * world_size * batch / 64)."""


def scale_learning_rate(lr: float, world_size: int, batch_size: int, base_batch_size: int = 64) -> float:
    return lr * world_size * batch_size / base_batch_size
