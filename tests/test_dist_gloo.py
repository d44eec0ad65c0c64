"""N > 1 path on CPU: world_size-2 (and 3) gloo groups exercise exactly the exchange the
multi-GPU job performs — each rank holds the tile-major buffer of its pixel shard, rank 0
gathers them (rtmi.dist.gather_to_root) and un-tiles the frame.  The per-pixel values come
from the oracle's full frame, so the assembled frame must equal it bit for bit; no rendering
happens in this test (there is no CPU render path in the product)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import common


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, h, w, full, out_path):
    import sys
    here = os.path.dirname(os.path.abspath(__file__))
    sys.path.insert(0, os.path.join(os.path.dirname(here), "ray-tracing-cuda_amd"))
    from rtmi.dist import gather_to_root, shard_pixel_map, untile_host
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    pm = shard_pixel_map(h, w, rank, world)
    flat = full.reshape(h * w, 3)
    tiles = np.zeros((pm.size, 3), dtype=np.float32)
    tiles[pm >= 0] = flat[pm[pm >= 0]]
    tiles[pm < 0] = -7.0  # padding must never reach the image
    allt = gather_to_root(torch.from_numpy(tiles), 0)
    # second exchange: per-rank ray totals summed like bench.py does
    tot = torch.tensor([float((pm >= 0).sum())], dtype=torch.float64)
    dist.all_reduce(tot, op=dist.ReduceOp.SUM)
    if rank == 0:
        img = untile_host(allt.numpy(), h, w, world)
        np.save(out_path, img)
        assert int(tot.item()) == h * w
    else:
        assert allt is None
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,h,w", [(2, 24, 32), (3, 20, 28), (2, 19, 45)])
def test_gather_and_untile_reassemble_the_frame(tmp_path, world, h, w):
    rgb, _, _, _, _ = common.oracle_render("cornell_box", h, w, 1, 4)
    out = str(tmp_path / "img.npy")
    mp.spawn(_worker, args=(world, _free_port(), h, w, rgb, out), nprocs=world, join=True)
    img = np.load(out)
    assert np.array_equal(img, rgb)


def test_shard_image_is_independent_of_world_size():
    """RNG subsequence == global pixel index, so a pixel's value cannot depend on which rank
    renders it: rendering each shard's pixels separately on the oracle reproduces the frame."""
    import oraclelib
    from rtmi.dist import shard_pixel_map
    h, w, spp, depth = 16, 24, 2, 6
    full, rays, _, _, _ = common.oracle_render("mixed", h, w, spp, depth)
    for world in (2, 3):
        acc = np.zeros_like(full)
        for r in range(world):
            pm = shard_pixel_map(h, w, r, world)
            ids = pm[pm >= 0].astype(np.int32)
            b = common.build_scene(oraclelib.OracleBuilder(common.scene_seed("mixed")), "mixed", w / h)
            part, _, _, _ = b.render(h, w, spp, depth, pixel_ids=ids)
            acc.reshape(-1, 3)[ids] = part.reshape(-1, 3)[ids]
        assert np.array_equal(acc, full)
