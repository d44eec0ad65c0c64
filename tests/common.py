"""Shared helpers: build one named scene on either implementation and render it."""
import numpy as np

from rtmi import scenes

_MESH = {}


def small_mesh(n=9):
    if n not in _MESH:
        _MESH[n] = scenes.procedural_bunny_mesh(n)  # 12*n*n triangles (972 by default)
    return _MESH[n]


def build_scene(b, name, aspect=1.0, **kw):
    """Run scene program `name` on builder `b` (oracle's or product's)."""
    if name == "cornell_box":
        scenes.cornell_box(b, aspect)
    elif name == "spheres":
        scenes.spheres(b, aspect)
    elif name == "bunny":
        scenes.bunny(b, aspect, kw.get("faces", small_mesh()), k_min=kw.get("k_min", 2048))
    elif name == "birthday":
        scenes.birthday(b, aspect, kw.get("earthmap", scenes.procedural_earthmap(64, 128)))
    elif name == "sky_only":
        scenes.sky_only(b, aspect)
    elif name == "furnace":
        scenes.furnace(b, aspect, kw.get("rho", 0.5))
    elif name == "mixed":
        scenes.mixed(b, aspect)
    else:
        raise KeyError(name)
    return b


def scene_seed(name):
    return scenes.SCENE_SEEDS.get(name, 1024)


def oracle_render(name, h, w, spp, depth, post=True, seed=None, threads=None, **kw):
    import oraclelib
    seed = scene_seed(name) if seed is None else seed
    b = build_scene(oraclelib.OracleBuilder(seed), name, w / h, **kw)
    rgb, rays, states, total = b.render(h, w, spp, depth, post=post, threads=threads)
    return rgb, rays, states, total, b


def gpu_render(name, h, w, spp, depth, post=True, seed=None, world_size=1, **kw):
    """Render through librtmi (C ABI) on cuda:0; world_size > 1 renders every shard on the
    one GPU and assembles them as an RCCL gather would."""
    import torch
    import rtmi
    seed = scene_seed(name) if seed is None else seed
    b = build_scene(rtmi.SceneBuilder(seed), name, w / h, **kw).commit()
    tiles, counts, states = [], [], []
    total = 0
    for r in range(world_size):
        R = rtmi.Renderer(b, h, w, spp, depth, post, rank=r, world_size=world_size)
        R.init_rng()
        R.render()
        total += R.total_rays()
        tiles.append(R.tiles)
        counts.append(R.ray_counts)
        states.append(R.states)
    img, cnt = R.untile(torch.cat(tiles, 0).contiguous(), torch.cat(counts, 0).contiguous())
    torch.cuda.synchronize()
    return img.cpu().numpy(), cnt.cpu().numpy().astype(np.uint32), states, total, b


def rel_l2(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return float(np.sqrt(((a - b) ** 2).sum()) / max(np.sqrt((b ** 2).sum()), 1e-30))
