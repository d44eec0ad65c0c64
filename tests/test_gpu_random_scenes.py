"""Randomised-scene parity: seeded random worlds mixing every primitive and material kind,
random cameras (pinhole and defocus), frame shapes, depth limits and post/raw modes, rendered
through the C ABI and compared bit for bit with the oracle.  Complements the fixed scenes: it
exercises list orders, run boundaries (sphere / triangle / sky interleavings), nested boxes,
single triangles between parallelograms, several BVHs per world and material-table sizes on
both sides of the uint8/uint16 id-stack switch."""
import numpy as np
import pytest

import oraclelib
import rtmi
from rtmi.scenes import v3, PI_D

pytestmark = pytest.mark.gpu


def random_world(b, rng, n_objects, many_materials=False):
    def col(lo=0.05, hi=0.95):
        return v3(*rng.uniform(lo, hi, 3))

    mats = []
    n_mats = 300 if many_materials else int(rng.integers(3, 9))
    for _ in range(n_mats):
        k = rng.integers(0, 10)
        if k < 5:
            mats.append(b.lambertian(col()))
        elif k < 7:
            mats.append(b.metal(col(0.3, 1.0), float(rng.choice([0.0, rng.uniform(0.05, 0.9), 1.7]))))
        elif k < 9:
            mats.append(b.dielectric(col(0.7, 1.0), float(rng.uniform(1.1, 2.0))))
        else:
            mats.append(b.diffuse_light(b.constant_texture(col(1.0, 6.0))))
    light = b.diffuse_light(b.constant_texture(v3(5, 5, 5)))

    def pick():
        return mats[int(rng.integers(0, len(mats)))]

    b.sphere(v3(0, -100.5, -1), 100.0, mats[0])
    sky_at = int(rng.integers(0, n_objects))
    for i in range(n_objects):
        if i == sky_at:
            b.sky()
        c = rng.uniform(-2.5, 2.5, 3)
        c[1] = abs(c[1]) * 0.6
        c[2] -= 2.0
        k = rng.integers(0, 6)
        if k == 0:
            b.sphere(v3(*c), float(rng.uniform(0.15, 0.6)), pick())
        elif k == 1:
            e1, e2 = rng.uniform(-0.8, 0.8, 3), rng.uniform(-0.8, 0.8, 3)
            b.parallelogram([v3(*c), v3(*(c + e1)), v3(*(c + e2))], pick())
        elif k == 2:
            e = rng.uniform(0.2, 0.7, 3)
            b.parallelepiped([v3(*c), v3(c[0] + e[0], c[1], c[2]), v3(c[0], c[1] + e[1], c[2]),
                              v3(c[0], c[1], c[2] + e[2])], pick())
        elif k == 3:
            b.triangle([v3(*c), v3(*(c + rng.uniform(-0.9, 0.9, 3))), v3(*(c + rng.uniform(-0.9, 0.9, 3)))], pick())
        elif k == 4:
            ang = np.float32(rng.uniform(0, 3.0))
            off = v3(*c)
            from rtmi.scenes import rotate_y
            b.parallelepiped_lengths(v3(*rng.uniform(0.2, 0.7, 3)), pick(), lambda p, a=ang, o=off: rotate_y(p, a) + o)
        else:
            n = int(rng.integers(1, 40))
            base = rng.uniform(-0.4, 0.4, (n, 1, 3)) + c
            faces = (base + rng.uniform(-0.25, 0.25, (n, 3, 3))).astype(np.float32)
            b.bvh(faces, pick(), k_min=int(rng.choice([2, 8, 2048])))
    b.parallelogram([v3(-1, 3.5, -3), v3(1, 3.5, -3), v3(-1, 3.5, -1)], light)


@pytest.mark.parametrize("seed", list(range(12)))
def test_random_world_bit_exact(seed):
    import torch
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(9, 41)), int(rng.integers(9, 57))
    spp, depth = int(rng.integers(1, 7)), int(rng.choice([1, 3, 10, 25, 64]))
    post = bool(rng.integers(0, 2))
    n_objects = int(rng.integers(3, 14))
    many = seed % 6 == 5
    defocus = seed % 4 == 3
    state = rng.bit_generator.state
    results = []
    for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
        rng.bit_generator.state = state  # both builders see the same random stream
        b = make(77 + seed)
        if defocus:
            b.camera_defocus(v3(0, 1.0, 2.5), v3(0, 0.4, -2), v3(0, 1, 0), PI_D / 3, w / h, 0.2, 4.0)
        else:
            b.camera_pinhole(v3(0, 1.0, 2.5), v3(0, 0.4, -2), v3(0, 1, 0), PI_D / 3, w / h)
        random_world(b, rng, n_objects, many)
        results.append(b)
    o, p = results
    o_rgb, o_rays, o_states, o_total = o.render(h, w, spp, depth, post=post)
    p.commit()
    R = rtmi.Renderer(p, h, w, spp, depth, post).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    assert R.total_rays() == o_total
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays)
    assert np.array_equal(img.cpu().numpy(), o_rgb), np.abs(img.cpu().numpy() - o_rgb).max()


def _layer_stack(n_layers, rng, jitter):
    """Quads perpendicular to the x axis, one after another along it: a ray travelling along x
    finds a hit in every layer, i.e. in far more reference leaves than the kernel's per-lane
    candidate list has slots."""
    faces = []
    for k in range(n_layers):
        x = -2.0 + 4.0 * k / max(n_layers - 1, 1)
        dx = rng.uniform(-jitter, jitter, 4)
        p = [np.array([x + dx[0], -0.6, -2.6]), np.array([x + dx[1], 0.9, -2.6]),
             np.array([x + dx[2], 0.9, -1.2]), np.array([x + dx[3], -0.6, -1.2])]
        faces.append([p[0], p[1], p[2]])
        faces.append([p[0], p[2], p[3]])
    return np.asarray(faces, dtype=np.float32)


@pytest.mark.parametrize("case", [
    # (layers, leaf size, material, jitter): leaves of 1..3 faces give 2x..0.7x leaves per layer
    # floor: a sphere selects the double-precision kernel variant, a parallelogram the float one
    (40, 1, "dielectric", 0.0, "sphere"), (40, 2, "dielectric", 0.02, "pgram"), (25, 3, "dielectric", 0.05, "pgram"),
    (12, 2, "lambertian", 0.0, "pgram"), (64, 2, "metal", 0.01, "sphere"), (40, 2048, "dielectric", 0.02, "pgram"),
    (40, 1, "dielectric", 0.0, "pgram"),
])
def test_many_hit_leaves_along_one_ray(case):
    """Mesh rays that hold hits in tens of reference leaves at once (more than kHitSlots):
    the search defers the leaves beyond its list to further passes; the result must still be
    the reference's in-order walk, bit for bit."""
    import torch
    n_layers, k_min, kind, jitter, floor = case
    h, w, spp, depth = 24, 40, 3, 30
    results = []
    for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
        rng = np.random.default_rng(4242)
        b = make(5)
        # the camera looks along +x through the whole stack
        b.camera_pinhole(v3(-4.5, 0.2, -1.9), v3(0, 0.15, -1.9), v3(0, 1, 0), PI_D / 4, w / h)
        if kind == "dielectric":
            m = b.dielectric(v3(0.95, 0.97, 0.99), 1.0)  # index 1: goes straight on, every layer is met
        elif kind == "metal":
            m = b.metal(v3(0.9, 0.9, 0.9), 0.3)
        else:
            m = b.lambertian(v3(0.7, 0.6, 0.5))
        b.bvh(_layer_stack(n_layers, rng, jitter), m, k_min=k_min)
        if floor == "sphere":
            b.sphere(v3(0, -100.7, -2), 100.0, b.lambertian(v3(0.5, 0.5, 0.5)))
        else:
            b.parallelogram([v3(-50, -0.7, -50), v3(50, -0.7, -50), v3(-50, -0.7, 50)], b.lambertian(v3(0.5, 0.5, 0.5)))
        b.sky()
        results.append(b)
    o, p = results
    o_rgb, o_rays, o_states, o_total = o.render(h, w, spp, depth, post=False)
    p.commit()
    R = rtmi.Renderer(p, h, w, spp, depth, False).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    assert o_total > h * w * spp * (2 if kind == "dielectric" else 1)  # paths do continue through the stack
    assert R.total_rays() == o_total
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays)
    assert np.array_equal(img.cpu().numpy(), o_rgb), np.abs(img.cpu().numpy() - o_rgb).max()


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_scheduler_paths_agree_on_random_mesh_worlds(seed):
    """Mesh-heavy random worlds rendered six ways -- image order, forced longest-first (probe,
    outlier pixels spread one per 16 lanes, their waves working for them alone), as two interleaved
    shards, with the outlier spreading switched off, with one outlier pixel per wave, and without the
    helper-wave rule -- must give the same image, ray counts and ray total bit for bit.  (The oracle
    comparison of such worlds is test_random_world_bit_exact; this one is large enough for the
    scheduler to matter: 192x256 at 64 spp.)"""
    import torch
    from rtmi.scenes import procedural_bunny_mesh
    rng = np.random.default_rng(500 + seed)
    h, w, spp, depth = 192, 256, 64, int(rng.choice([6, 12, 30]))
    b = rtmi.SceneBuilder(90 + seed)
    b.camera_pinhole(v3(0.0, 0.16, -0.55), v3(0.0, 0.1, 0.0), v3(0, 1, 0), PI_D / 4, w / h)
    mats = [b.lambertian(v3(1, 1, 1)), b.metal(v3(0.9, 0.85, 0.8), 0.05), b.dielectric(v3(0.95, 0.95, 1.0), 1.4)]
    mesh = procedural_bunny_mesh(int(rng.integers(10, 24)))
    for k in range(int(rng.integers(1, 4))):
        off = np.array([0.17 * (k - 1), 0.0, 0.05 * k], dtype=np.float32)
        b.bvh((mesh + off).astype(np.float32), mats[int(rng.integers(0, 3))], k_min=int(rng.choice([4, 64, 2048])))
    if seed % 2:
        b.sphere(v3(0, -100.0 + 0.03, 0), 100.0, b.lambertian(v3(0.5, 0.6, 0.5)))
    else:
        b.parallelogram([v3(-5, 0.03, -5), v3(5, 0.03, -5), v3(-5, 0.03, 5)], b.lambertian(v3(0.5, 0.6, 0.5)))
    b.sky()
    b.commit()

    def run(world, **opts):
        tiles, counts, total = [], [], 0
        o = rtmi.render_opts(**opts)
        for r in range(world):
            R = rtmi.Renderer(b, h, w, spp, depth, True, rank=r, world_size=world).init_rng()
            R.render(opts=o)
            total += R.total_rays()
            tiles.append(R.tiles), counts.append(R.ray_counts)
        img, cnt = R.untile(torch.cat(tiles, 0).contiguous(), torch.cat(counts, 0).contiguous())
        torch.cuda.synchronize()
        return img.cpu().numpy(), cnt.cpu().numpy(), total

    plain = run(1, schedule=0)
    forced = run(1, schedule=2)
    sharded = run(2, schedule=2)
    solo = run(1, schedule=2, sparse_stride=1)    # outlier pixels packed 64 to a wave
    few = run(1, schedule=2, sparse_stride=64)    # ... or one to a wave
    shared = run(1, schedule=2, exclusive=0)      # ... or sharing their wave with ordinary pixels
    for other in (forced, sharded, solo, few, shared):
        assert np.array_equal(plain[0], other[0]) and np.array_equal(plain[1], other[1]) and plain[2] == other[2]
    assert plain[2] > h * w * spp * 1.2


def _nested_world(b, rng):
    """Lists inside lists (hitable_list.cuh:8), with coincident surfaces across nesting levels so that
    the tie rules of hitable_list.cu:12-20 decide pixels: the nested call accepts its first hit
    unconditionally and the parent then demands a strictly nearer one."""
    def col():
        return v3(*rng.uniform(0.1, 0.9, 3))
    mats = [b.lambertian(col()), b.lambertian(col()), b.metal(col(), float(rng.uniform(0, 0.4))), b.dielectric(v3(1, 1, 1), 1.5),
            b.diffuse_light(b.constant_texture(v3(3, 3, 3)))]

    def wall(z, m, dx=0.0):
        b.parallelogram([v3(-1.5 + dx, -0.2, z), v3(1.5 + dx, -0.2, z), v3(-1.5 + dx, 1.8, z)], mats[m])

    b.sky()
    b.list_begin()                      # A
    wall(-2.0, 0)                       #   back wall, material 0 ...
    b.list_begin()                      #   B inside A
    wall(-2.0, 1, 0.7)                  #     ... overlapped by a coincident wall of material 1 (a tie, first wins)
    b.sphere(v3(-0.6, 0.5, -1.0), 0.45, mats[2])
    b.list_begin()                      #     C inside B
    b.parallelepiped([v3(0.3, 0.0, -1.4), v3(0.9, 0.0, -1.4), v3(0.3, 0.7, -1.4), v3(0.3, 0.0, -0.8)], mats[3])
    b.parallelogram([v3(0.3, 0.0, -0.8), v3(0.9, 0.0, -0.8), v3(0.3, 0.7, -0.8)], mats[1])  # coincident with a box face
    b.list_end()
    b.list_end()
    b.triangle([v3(-1.4, 1.2, -1.9), v3(-0.4, 1.2, -1.9), v3(-0.9, 1.9, -1.9)], mats[4])
    b.list_end()
    b.list_begin()                      # an empty list, then one holding only the floor
    b.list_end()
    b.list_begin()
    b.sphere(v3(0, -100.2, -1), 100.0, mats[0])
    b.list_end()
    wall(-2.0, 2, -0.9)                 # and a coincident wall at world level, after the nested ones
    for _ in range(int(rng.integers(0, 3))):
        b.sphere(v3(*rng.uniform(-1, 1, 2), -1.0 + float(rng.uniform(-0.3, 0.3))), float(rng.uniform(0.1, 0.3)), mats[int(rng.integers(0, 4))])


@pytest.mark.parametrize("seed", [0, 1, 2, 3])
def test_nested_lists_match_the_nested_oracle(seed):
    """The library inlines nested HitableLists; the oracle builds them as real nested objects whose
    Hit() calls recurse (hitable_list.cu:7-25).  Image, ray counts and ray total must agree bit for
    bit -- including where coincident surfaces of different nesting levels tie."""
    import torch
    h, w, spp, depth = 40, 52, 4, 12
    res = []
    for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
        rng = np.random.default_rng(900 + seed)
        b = make(31 + seed)
        b.camera_pinhole(v3(0, 0.8, 2.2), v3(0, 0.6, -1), v3(0, 1, 0), PI_D / 3, w / h)
        _nested_world(b, rng)
        res.append(b)
    o, p = res
    o_rgb, o_rays, _, o_total = o.render(h, w, spp, depth)
    p.commit()
    R = rtmi.Renderer(p, h, w, spp, depth).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    assert R.total_rays() == o_total
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays)
    assert np.array_equal(img.cpu().numpy(), o_rgb)


def _clump(n, rng, spread):
    """n faces around one point: centroids nearly coincide, so the search tree's splits fall back
    to medians and the tree is as deep as it gets for n faces."""
    c = rng.uniform(-0.2, 0.2, (n, 1, 3)) * spread + np.array([0.0, 0.3, -2.0])
    return (c + rng.uniform(-0.5, 0.5, (n, 3, 3))).astype(np.float32)


@pytest.mark.parametrize("case", [
    # (frame h, w, spp, depth, faces per mesh, reference leaf size, meshes): frames of at most 16 pixels keep every
    # search at <= 16 rays, i.e. on the top-table start; 5 x 13 starts dense and thins out towards the end of the frame
    (3, 5, 9, 12, 600, 2, 1), (4, 4, 7, 20, 1, 2048, 3), (2, 8, 6, 12, 2500, 1, 2), (2, 2, 40, 30, 300, 4, 2),
    (5, 13, 5, 12, 900, 3, 2), (3, 5, 9, 12, 70, 2048, 1),
])
def test_sparse_waves_start_from_the_top_table(case):
    """Waves with a handful of rays (tiny frames) on deep, clumped meshes with small reference leaves:
    the search starts from the mesh's table of sub-trees, runs in its four- and two-lanes-per-entry
    modes, and the replay reads leaf-path rows of 8, 16 and 32 words, from LDS and from global memory."""
    import torch
    h, w, spp, depth, n_faces, k_min, n_meshes = case
    results = []
    for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
        rng = np.random.default_rng(977)
        b = make(31)
        b.camera_pinhole(v3(0.1, 0.4, 1.2), v3(0, 0.3, -2.0), v3(0, 1, 0), PI_D / 5, w / h)
        mats = [b.dielectric(v3(0.95, 0.97, 0.99), 1.2), b.metal(v3(0.9, 0.85, 0.8), 0.2), b.lambertian(v3(0.8, 0.7, 0.6))]
        for i in range(n_meshes):
            b.bvh(_clump(n_faces, rng, 0.05 if i == 0 else 1.0), mats[i % 3], k_min=k_min)
        b.parallelogram([v3(-50, -0.7, -50), v3(50, -0.7, -50), v3(-50, -0.7, 50)], b.lambertian(v3(0.5, 0.5, 0.5)))
        b.sky()
        results.append(b)
    o, p = results
    o_rgb, o_rays, o_states, o_total = o.render(h, w, spp, depth, post=False)
    p.commit()
    R = rtmi.Renderer(p, h, w, spp, depth, False).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    assert o_total > h * w * spp  # the meshes are in view
    assert R.total_rays() == o_total
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays)
    assert np.array_equal(img.cpu().numpy(), o_rgb), np.abs(img.cpu().numpy() - o_rgb).max()
