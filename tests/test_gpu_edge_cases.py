"""Edge cases of the trace path through the C ABI: worlds without Sky (a true miss is black,
ray_tracing.cu:23-25), an empty world, an empty mesh, zero samples, a pitched texture upload."""
import ctypes as C

import numpy as np
import pytest

import oraclelib
import rtmi
from rtmi.scenes import v3, PI_D

pytestmark = pytest.mark.gpu


def both(fill, seed=5):
    out = []
    for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
        b = make(seed)
        b.camera_pinhole(v3(0, 0.5, 3), v3(0, 0.3, 0), v3(0, 1, 0), PI_D / 3, 1.25)
        fill(b)
        out.append(b)
    return out


def render_both(fill, h=24, w=30, spp=3, depth=10, post=True):
    import torch
    o, p = both(fill)
    o_rgb, o_rays, _, o_total = o.render(h, w, spp, depth, post=post)
    p.commit()
    R = rtmi.Renderer(p, h, w, spp, depth, post).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    assert R.total_rays() == o_total
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays)
    assert np.array_equal(img.cpu().numpy(), o_rgb)
    return o_rgb, o_rays


def test_world_without_sky_misses_are_black():
    def fill(b):
        b.sphere(v3(0, 0.3, 0), 0.5, b.lambertian(v3(0.8, 0.3, 0.3)))
        b.parallelogram([v3(-1, 2, -1), v3(1, 2, -1), v3(-1, 2, 1)], b.diffuse_light(b.constant_texture(v3(3, 3, 3))))
    rgb, rays = render_both(fill)
    assert (rgb[0, 0] == 0).all() and rgb.max() > 0


def test_empty_world():
    rgb, rays = render_both(lambda b: None)
    assert rgb.max() == 0 and (rays == 3).all()


def test_empty_mesh_and_single_face_mesh():
    def fill(b):
        b.sky()
        m = b.lambertian(v3(0.5, 0.5, 0.9))
        b.bvh(np.zeros((0, 3, 3), dtype=np.float32), m)
        b.bvh(np.array([[[-1, 0, 0], [1, 0, 0], [0, 1.5, 0]]], dtype=np.float32), m)
    rgb, rays = render_both(fill)
    assert (rays > 3).any()


def test_zero_samples_writes_zeros():
    import torch
    p = rtmi.SceneBuilder(1)
    p.camera_pinhole(v3(0, 0, 3), v3(0, 0, 0), v3(0, 1, 0), 1.0, 1.0)
    p.sky()
    p.commit()
    R = rtmi.Renderer(p, 16, 16, 0, 10).init_rng()
    R.tiles.fill_(7.0)
    R.render()
    torch.cuda.synchronize()
    assert float(R.tiles.abs().max()) == 0.0 and R.total_rays() == 0


def test_pitched_texture_rows():
    """rtmi_image_texture honours pitch_bytes (cudaMemcpy2D semantics, scenes/birthday.cu:81-83)."""
    import torch
    from rtmi import scenes
    tex = scenes.procedural_earthmap(16, 24)
    padded = np.zeros((16, 40, 4), dtype=np.uint8)
    padded[:, :24] = tex
    padded[:, 24:] = 99  # padding bytes must never be sampled

    def fill_o(b):
        b.sky()
        b.sphere(v3(0, 0.3, 0), 0.9, b.lambertian_tex(b.image_texture(tex)))

    o = oraclelib.OracleBuilder(3)
    o.camera_pinhole(v3(0, 0.5, 3), v3(0, 0.3, 0), v3(0, 1, 0), PI_D / 3, 1.0)
    fill_o(o)
    o_rgb, o_rays, _, _ = o.render(32, 32, 4, 10)
    p = rtmi.SceneBuilder(3)
    p.camera_pinhole(v3(0, 0.5, 3), v3(0, 0.3, 0), v3(0, 1, 0), PI_D / 3, 1.0)
    p.sky()
    L = rtmi.lib()
    t = L.rtmi_image_texture(p.h, padded.ctypes.data_as(C.POINTER(C.c_uint8)), 16, 24, 40 * 4)
    assert t >= 0
    p.sphere(v3(0, 0.3, 0), 0.9, p.lambertian_tex(t))
    p.commit()
    R = rtmi.Renderer(p, 32, 32, 4, 10).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays)
    rel = np.sqrt(((img.cpu().numpy() - o_rgb) ** 2).sum() / (o_rgb ** 2).sum())
    assert rel <= 1e-3


@pytest.mark.parametrize("floor", ["sphere", "pgram"])
def test_mesh_with_more_reference_nodes_than_the_lds_stage_holds(floor):
    """2,700 faces in leaves of <= 2 give ~2,700 reference-tree nodes: only the first 512 are
    staged in LDS, the replay of the box tests reads the rest from global memory.  Two meshes in
    one world, so node / face / search-tree indices of the second are rebased."""
    from rtmi.scenes import procedural_bunny_mesh
    mesh = procedural_bunny_mesh(15)  # 12 * 15 * 15 faces around (-0.017, 0.11, 0)

    def fill(b):
        b.camera_pinhole(v3(0.04, 0.14, -0.45), v3(0.04, 0.1, 0.02), v3(0, 1, 0), PI_D / 5, 1.25)
        b.bvh(mesh, b.lambertian(v3(0.9, 0.9, 0.9)), k_min=2)
        shifted = (mesh + np.array([0.12, 0.0, 0.05], dtype=np.float32)).astype(np.float32)
        b.bvh(shifted, b.metal(v3(0.9, 0.8, 0.7), 0.1), k_min=3)
        if floor == "sphere":
            b.sphere(v3(0, -100.0 + 0.03, 0), 100.0, b.lambertian(v3(0.4, 0.6, 0.4)))
        else:
            b.parallelogram([v3(-5, 0.03, -5), v3(5, 0.03, -5), v3(-5, 0.03, 5)], b.lambertian(v3(0.4, 0.6, 0.4)))
        b.sky()
    rgb, rays = render_both(fill, h=40, w=50, spp=2, depth=12, post=False)
    assert (rays > 2).mean() > 0.3


def test_colours_with_a_sign_bit_keep_the_emitted_plus_product_addition():
    """ray_tracing.cu:51 folds `emitted + attenuation * result` with emitted == 0 for everything
    that scatters; the kernel drops the addition unless some material colour has its sign bit set,
    because only a product of -0 notices it (0 + -0 = +0).  Here colours are negative and -0."""
    def fill(b):
        b.sphere(v3(0, -100.5, 0), 100.0, b.lambertian(v3(-0.0, 0.6, 0.5)))
        b.sphere(v3(-0.6, 0.3, 0), 0.5, b.lambertian(v3(0.8, -0.25, 0.3)))
        b.sphere(v3(0.6, 0.3, 0), 0.5, b.metal(v3(-0.0, -0.0, 0.9), 0.2))
        b.parallelogram([v3(-1, 2, -1), v3(1, 2, -1), v3(-1, 2, 1)], b.diffuse_light(b.constant_texture(v3(3, -0.0, 2))))
        b.sky()
    rgb, rays = render_both(fill, h=32, w=40, spp=4, depth=12, post=False)
    assert np.signbit(rgb).any() or (rgb < 0).any()


def _tiny_sheet():
    rng = np.random.default_rng(17)
    n = 24
    g = (np.arange(n + 1, dtype=np.float64) / n - 0.5) * 0.04
    X, Z = np.meshgrid(g, g, indexing="ij")
    Y = 0.002 * np.sin(40.0 * X) * np.cos(55.0 * Z) + rng.uniform(-2e-4, 2e-4, X.shape)
    P = np.stack([X, Y, Z], axis=-1).astype(np.float32)
    p00, p10, p01, p11 = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
    return np.concatenate([np.stack([p00, p10, p11], axis=2).reshape(-1, 3, 3),
                           np.stack([p00, p11, p01], axis=2).reshape(-1, 3, 3)], axis=0).astype(np.float32)


@pytest.mark.parametrize("dist,mode", [(1e3, "direct"), (1e3, "grazing"), (1e3, "mirror"), (4e3, "mirror"), (2e4, "mirror")])
def test_tiny_faces_seen_from_far_away(dist, mode):
    """Faces of size ~1e-3 near the world origin hit by rays that start 1e3..2e4 away: seen directly
    (as far as the reference's binary32 camera frame can resolve such a view at all), at grazing
    incidence, and by rays coming back from a distant mirror.  At that range the binary32
    Moller-Trumbore test (utils.cu:49-85) accepts rays that miss the exact triangle by about
    4e-7 x distance -- as much as a whole face here -- and the reference, which scans every face of
    an entered leaf, reports those hits.  The search boxes are widened in proportion to the ray
    origin's distance (MESH_DIST_SLACK), so the mesh search must still find every face the test
    accepts: image, ray counts and ray total equal the oracle's."""
    import torch
    faces = _tiny_sheet()
    h, w, spp, depth = 28, 36, 2, 4
    res = []
    for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
        b = make(9)
        grey = b.lambertian(v3(0.8, 0.8, 0.8))
        if mode == "mirror":
            # camera just above the sheet looking up at a mirror `dist` away: the rays come back down
            b.camera_pinhole(v3(0.002, 0.05, 0.001), v3(0.002, 1.0, 0.001), v3(0, 0, 1), float(2.0 * np.arctan(0.012 / dist)), w / h)
            b.parallelogram([v3(-50, dist, -50), v3(50, dist, -50), v3(-50, dist, 50)], b.metal(v3(0.9, 0.9, 0.9), 0.0))
        else:
            elev = 0.02 if mode == "grazing" else 0.6
            pos = v3(0.3 * dist * 0.01, dist * np.sin(elev), dist * np.cos(elev))
            b.camera_pinhole(pos, v3(0.0031, 0, 0.0017), v3(0, 1, 0), float(2.0 * np.arctan(0.03 / dist)), w / h)
        b.bvh(faces, grey, k_min=64)
        b.sky()
        res.append(b)
    o, p = res
    o_rgb, o_rays, _, o_total = o.render(h, w, spp, depth)
    p.commit()
    R = rtmi.Renderer(p, h, w, spp, depth).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    assert o_total > h * w * spp * (2.2 if mode == "mirror" else 1.0)  # the sheet is hit at all
    assert R.total_rays() == o_total
    assert np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays)
    assert np.array_equal(img.cpu().numpy(), o_rgb)
