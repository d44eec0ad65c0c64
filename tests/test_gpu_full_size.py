"""BASELINE.json full-size configurations on the GPU, checked through properties that do not
need a full CPU render: (a) a random sample of pixels of the FULL config against the oracle
(the oracle renders just those pixels: same seed, same subsequence, all 1024 spp), (b)
shard invariance, (c) run-to-run determinism, (d) ray accounting."""
import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu


def _oracle_pixels(name, h, w, spp, depth, ids, **kw):
    import oraclelib
    b = common.build_scene(oraclelib.OracleBuilder(common.scene_seed(name)), name, w / h, **kw)
    rgb, rays, _, _ = b.render(h, w, spp, depth, pixel_ids=ids)
    return rgb.reshape(-1, 3)[ids], rays.reshape(-1)[ids]


def test_c2_cornell_1024sq_1024spp_depth50_sampled_pixels():
    """configs[1] at full size: 512 random pixels (plus the corners) bit-exact vs the oracle."""
    h = w = 1024
    spp, depth = 1024, 50
    g_rgb, g_rays, _, g_total, _ = common.gpu_render("cornell_box", h, w, spp, depth)
    rng = np.random.default_rng(2026)
    ids = np.unique(np.concatenate([rng.integers(0, h * w, 512), [0, w - 1, (h - 1) * w, h * w - 1]])).astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("cornell_box", h, w, spp, depth, ids)
    assert np.array_equal(g_rays.reshape(-1)[ids], o_rays)
    assert np.array_equal(g_rgb.reshape(-1, 3)[ids], o_rgb)
    assert g_total == int(g_rays.astype(np.uint64).sum())
    assert g_rays.min() >= spp and g_rays.max() <= spp * (depth + 1)
    assert np.isfinite(g_rgb).all() and g_rgb.min() >= 0 and g_rgb.max() <= 1


def test_c2_shape_shard_invariance_and_determinism():
    """1024x1024 frame at reduced spp: 1 shard == 8 shards == a second run, bit for bit."""
    h = w = 1024
    a = common.gpu_render("cornell_box", h, w, 4, 50)
    b = common.gpu_render("cornell_box", h, w, 4, 50, world_size=8)
    c = common.gpu_render("cornell_box", h, w, 4, 50)
    for other in (b, c):
        assert np.array_equal(a[0], other[0]) and np.array_equal(a[1], other[1]) and a[3] == other[3]


def test_c3_bunny_mesh_1024sq_sampled_pixels():
    """configs[2] shape (procedural stand-in mesh, 69,312 faces, reference leaf size 2048) at
    1024x1024 with reduced spp (16 of 512); sampled pixels bit-exact vs the oracle."""
    from rtmi import scenes
    h = w = 1024
    spp, depth = 16, 10
    faces = scenes.procedural_bunny_mesh()
    g_rgb, g_rays, _, g_total, _ = common.gpu_render("bunny", h, w, spp, depth, faces=faces)
    rng = np.random.default_rng(7)
    # bias the sample towards the mesh (centre of the frame)
    ii = rng.integers(300, 724, 200)
    jj = rng.integers(300, 724, 200)
    ids = np.unique(np.concatenate([ii * w + jj, rng.integers(0, h * w, 56)])).astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("bunny", h, w, spp, depth, ids, faces=faces)
    assert np.array_equal(g_rays.reshape(-1)[ids], o_rays)
    assert np.array_equal(g_rgb.reshape(-1, 3)[ids], o_rgb)
    assert (g_rays.reshape(-1)[ids] > spp).any()  # some sampled pixels do bounce off the mesh


def test_c3_outlier_spreading_and_queue_order_change_no_pixel():
    """At 64 spp the mesh frame goes through the scheduler: the 2-spp probe, the longest-first tile
    order and -- its cost distribution being skewed -- the outlier tiles spread one pixel per 16 lanes
    with cooperative searches.  None of it may change a value: the frame equals the image-order
    render bit for bit (image, ray counts, ray total), and sampled pixels equal the oracle."""
    import rtmi
    from rtmi import scenes
    h = w = 1024
    spp, depth = 64, 10
    faces = scenes.procedural_bunny_mesh()
    try:
        assert rtmi.lib().rtmi_set_schedule(0) == 0
        plain = common.gpu_render("bunny", h, w, spp, depth, faces=faces)
        assert rtmi.lib().rtmi_set_schedule(1) == 0
        sched = common.gpu_render("bunny", h, w, spp, depth, faces=faces)
    finally:
        rtmi.lib().rtmi_set_schedule(1)
    assert np.array_equal(plain[0], sched[0]) and np.array_equal(plain[1], sched[1]) and plain[3] == sched[3]
    rays = sched[1].reshape(-1)
    tiles = rays.reshape(h // 8, 8, w // 8, 8).sum(axis=(1, 3)).reshape(-1)
    assert tiles.max() >= 3 * tiles.mean()  # the distribution that switches the outlier spreading on
    heavy = np.argsort(-rays)[:24].astype(np.int32)  # the heaviest pixels are the ones that were spread
    ids = np.unique(np.concatenate([heavy, np.random.default_rng(3).integers(0, h * w, 24).astype(np.int32)]))
    o_rgb, o_rays = _oracle_pixels("bunny", h, w, spp, depth, ids, faces=faces)
    assert np.array_equal(rays[ids], o_rays)
    assert np.array_equal(sched[0].reshape(-1, 3)[ids], o_rgb)


def test_c4_c5_shapes_sampled_pixels():
    """configs[3] (cornell 2048^2) and configs[4] (birthday 4096^2) frame shapes at reduced spp,
    rendered as 8 shards; sampled pixels vs the oracle (birthday within the texel tolerance)."""
    from rtmi import scenes
    rng = np.random.default_rng(11)
    h = w = 2048
    g_rgb, g_rays, _, _, _ = common.gpu_render("cornell_box", h, w, 2, 50, world_size=8)
    ids = np.unique(rng.integers(0, h * w, 128)).astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("cornell_box", h, w, 2, 50, ids)
    assert np.array_equal(g_rays.reshape(-1)[ids], o_rays) and np.array_equal(g_rgb.reshape(-1, 3)[ids], o_rgb)
    h = w = 4096
    tex = scenes.procedural_earthmap(256, 512)
    g_rgb, g_rays, _, _, _ = common.gpu_render("birthday", h, w, 1, 10, world_size=8, earthmap=tex)
    ids = np.unique(rng.integers(0, h * w, 128)).astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("birthday", h, w, 1, 10, ids, earthmap=tex)
    assert np.array_equal(g_rays.reshape(-1)[ids], o_rays)
    assert common.rel_l2(g_rgb.reshape(-1, 3)[ids], o_rgb) <= 1e-3


def test_million_face_mesh_sampled_pixels():
    """1.08 M faces: 2,047 reference-tree nodes (only the first 512 are staged in LDS), a search tree
    ten levels deep (31 of the 32 stack entries the kernel has).  512x512 at 64 spp goes through the
    scheduler; sampled pixels, the heaviest included, equal the oracle bit for bit."""
    from rtmi import scenes
    h = w = 512
    spp, depth = 64, 10
    faces = scenes.procedural_bunny_mesh(300)
    assert faces.shape[0] == 1_080_000
    g_rgb, g_rays, _, g_total, b = common.gpu_render("bunny", h, w, spp, depth, faces=faces)
    st = b.stats()
    assert st["bvh_nodes"] == 2047 and st["bvh_faces"] >= 1_080_000
    rays = g_rays.reshape(-1)
    ids = np.unique(np.concatenate([np.argsort(-rays)[:10], np.random.default_rng(1).integers(0, h * w, 14),
                                    [256 * w + 256, 250 * w + 260]])).astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("bunny", h, w, spp, depth, ids, faces=faces)
    assert np.array_equal(rays[ids], o_rays)
    assert np.array_equal(g_rgb.reshape(-1, 3)[ids], o_rgb)
    assert rays[ids].max() > 2 * spp


# ----------------------------------------------------------------------------------------------
# BASELINE.json configs[2..4] at their FULL sample counts.  The oracle renders only the sampled
# pixels (same seed, same cuRAND subsequence = global pixel index, every sample of the pixel), so
# each sampled pixel carries its RNG state through all 512 / 4096 / 8192 samples on both sides.
def _gpu_shard(name, h, w, spp, depth, rank, world, **kw):
    """One rank's shard of the frame through the C ABI: tile-major rgb, ray counts, pixel map."""
    import torch
    import rtmi
    b = common.build_scene(rtmi.SceneBuilder(common.scene_seed(name)), name, w / h, **kw).commit()
    R = rtmi.Renderer(b, h, w, spp, depth, True, rank=rank, world_size=world)
    R.init_rng()
    R.render()
    total = R.total_rays()
    torch.cuda.synchronize()
    pm = rtmi.pixel_map(R.frame)
    return R.tiles.cpu().numpy(), R.ray_counts.cpu().numpy().astype(np.uint32), pm, total


def test_c3_bunny_full_512spp_sampled_pixels():
    """configs[2] at full size: 1024x1024 x512 spp, depth 10, the stand-in mesh (69,312 faces,
    reference leaf size 2048).  Sampled pixels -- the heaviest chains of the frame included -- are
    bit-exact against the oracle."""
    from rtmi import scenes
    h = w = 1024
    spp, depth = 512, 10
    faces = scenes.procedural_bunny_mesh()
    g_rgb, g_rays, _, g_total, _ = common.gpu_render("bunny", h, w, spp, depth, faces=faces)
    rays = g_rays.reshape(-1)
    rng = np.random.default_rng(70)
    ids = np.unique(np.concatenate([np.argsort(-rays.astype(np.int64))[:6], rng.integers(300, 724, 40) * w + rng.integers(300, 724, 40),
                                    rng.integers(0, h * w, 18)])).astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("bunny", h, w, spp, depth, ids, faces=faces)
    assert np.array_equal(rays[ids], o_rays)
    assert np.array_equal(g_rgb.reshape(-1, 3)[ids], o_rgb)
    assert g_total == int(rays.astype(np.uint64).sum())
    assert rays[ids].max() > 8 * spp  # chains that bounce to the depth limit inside the mesh are in the sample


def test_c4_cornell_2048sq_4096spp_shard0_of_8_sampled_pixels():
    """configs[3] at full size: cornell_box 2048x2048 x4096 spp, depth 50, rank 0's shard of an
    8-GPU job (every 8th 8x8 tile).  Sampled pixels of the shard are bit-exact against the oracle."""
    h = w = 2048
    spp, depth = 4096, 50
    tiles, rays, pm, total = _gpu_shard("cornell_box", h, w, spp, depth, 0, 8)
    rng = np.random.default_rng(44)
    qs = np.unique(rng.integers(0, pm.size, 96))
    qs = qs[pm[qs] >= 0]
    ids = pm[qs].astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("cornell_box", h, w, spp, depth, ids)
    assert np.array_equal(rays[qs], o_rays)
    assert np.array_equal(tiles[qs], o_rgb)
    assert total == int(rays.astype(np.uint64).sum())
    assert rays.min() >= spp or (pm < 0).any()


def test_c5_birthday_4096sq_8192spp_shard0_of_8_sampled_pixels():
    """configs[4] at full size: birthday 4096x4096 x8192 spp (depth 10, the reference's
    TRACE_DEPTH_LIMIT), rank 0's shard of an 8-GPU job.  Ray counts of the sampled pixels are exact;
    colours are held to the north-star tolerance (1e-3 relative L2) because the image-textured sphere
    goes through acosf/atan2f of two different libms (DESIGN.md "Oracle and parity status")."""
    from rtmi import scenes
    h = w = 4096
    spp, depth = 8192, 10
    tex = scenes.procedural_earthmap(1024, 2048)
    tiles, rays, pm, total = _gpu_shard("birthday", h, w, spp, depth, 0, 8, earthmap=tex)
    rng = np.random.default_rng(45)
    qs = np.unique(rng.integers(0, pm.size, 64))
    qs = qs[pm[qs] >= 0]
    ids = pm[qs].astype(np.int32)
    o_rgb, o_rays = _oracle_pixels("birthday", h, w, spp, depth, ids, earthmap=tex)
    assert np.array_equal(rays[qs], o_rays)
    assert common.rel_l2(tiles[qs], o_rgb) <= 1e-3
    assert total == int(rays.astype(np.uint64).sum())
