"""Multi-GPU plumbing: one process per GPU, frames sharded by interleaved 8x8 pixel tiles.

The reference distributes by samples-per-pixel and sum-reduces full frames over MPI
(/root/reference/ray-tracing-cuda/utils.cu:111-130, 181-242).  Here every rank renders
its own tiles of the frame (RNG subsequence == global pixel index, so the image does not
depend on the number of ranks) and the only exchange is one gather of the tile-major
radiance buffers to rank 0 over RCCL/xGMI (``torch.distributed`` backend "nccl"); the
gloo backend runs the same code on CPU tensors for tests.
"""
import ctypes as C

import numpy as np

from . import Frame, lib, work_items


def env_rank_world():
    import os
    return int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))


def gather_to_root(t, dst=0):
    """Gather equal-sized tensors to ``dst``; returns the concatenation on dst, None elsewhere."""
    import torch
    import torch.distributed as dist
    if not dist.is_initialized():
        return t
    world = dist.get_world_size()
    if dist.get_rank() == dst:
        parts = [torch.empty_like(t) for _ in range(world)]
        dist.gather(t, gather_list=parts, dst=dst)
        return torch.cat(parts, 0)
    dist.gather(t, gather_list=None, dst=dst)
    return None


def shard_pixel_map(height, width, rank, world_size):
    """Global pixel index of every work item of a shard (-1 for padding); host arithmetic only."""
    from . import pixel_map
    return pixel_map(Frame(height, width, 1, 1, 0, rank, world_size))


def untile_host(all_tiles, height, width, world_size):
    """Assemble the row-major image from the gathered tile-major buffers on the host.

    ``all_tiles``: array (world_size * items, C).  Pure index shuffling (no rendering
    arithmetic); the device version is rtmi_untile."""
    all_tiles = np.asarray(all_tiles)
    c = all_tiles.shape[1] if all_tiles.ndim == 2 else 1
    flat = all_tiles.reshape(-1, c)
    img = np.zeros((height * width, c), dtype=all_tiles.dtype)
    items = flat.shape[0] // world_size
    for r in range(world_size):
        pm = shard_pixel_map(height, width, r, world_size)
        assert pm.size == items
        ok = pm >= 0
        img[pm[ok]] = flat[r * items:(r + 1) * items][ok]
    return img.reshape(height, width, c)
