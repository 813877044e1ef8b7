"""Queue-size trimming — /root/reference/utils/moco.py:8-21 (changes numerics: K must be a multiple of the global batch)."""
import torch


def trim_moco_k(k: int, batch_size: int, world_size: int) -> int:
    total_batch_size = batch_size * world_size
    return k // total_batch_size * total_batch_size


def replace_moco_k_in_config(cfg, moco_k_key="moco.k", batch_size_key="batch_size", world_size=None):
    """Ensure K is a multiple of batch_size * #GPUs.  The reference uses torch.cuda.device_count() (not --ws); pass
    world_size to override.  Works on a pyhocon ConfigTree (put) or on the resolved nested dict."""
    ws = world_size if world_size is not None else max(torch.cuda.device_count(), 1)

    def get(d, dotted):
        if hasattr(d, "get_int"):
            return d.get_int(dotted)
        for part in dotted.split("."):
            d = d[part]
        return int(d)

    k = trim_moco_k(get(cfg, moco_k_key), get(cfg, batch_size_key), ws)
    if hasattr(cfg, "put"):
        cfg.put(moco_k_key, k)
    else:
        d = cfg
        parts = moco_k_key.split(".")
        for part in parts[:-1]:
            d = d[part]
        d[parts[-1]] = k
    return k
