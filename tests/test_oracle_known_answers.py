"""Known-answer pins of the CPU oracle: analytic facts that the
reference's own source states or implies, checked on the oracle's functions directly.
Citations are paths under /root/reference/ray-tracing-cuda/."""
import math

import numpy as np
import pytest

import common
import oraclelib
from rtmi import scenes
from rtmi.scenes import v3

f32 = np.float32


# ---------------------------------------------------------------- k2: utils.cu:111-113
def test_get_workload_partitions_spp():
    L = oraclelib.lib()
    for spp in [1, 7, 20, 100, 200, 1024]:
        for world in [1, 2, 3, 4, 7, 8, 16]:
            parts = [L.orc_get_workload(r, world, spp) for r in range(world)]
            assert sum(parts) == spp
            assert max(parts) - min(parts) <= 1
            assert parts == sorted(parts, reverse=True)


# ---------------------------------------------------------------- k3: sphere.cu:56-58
@pytest.mark.parametrize("normal,uv", [
    ((1, 0, 0), (0.50, 0.50)), ((-1, 0, 0), (0.00, 0.50)),
    ((0, 1, 0), (0.50, 1.00)), ((0, -1, 0), (0.50, 0.00)),
    ((0, 0, 1), (0.25, 0.50)), ((0, 0, -1), (0.75, 0.50)),
])
def test_sphere_uv_table(normal, uv):
    b = oraclelib.OracleBuilder(1)
    m = b.lambertian(v3(1, 1, 1))
    b.sphere(v3(0, 0, 0), 1.0, m)
    n = np.array(normal, dtype=np.float32)
    hit, rec, _ = b.probe_hit(3 * n, -n)
    assert hit
    assert rec[0] == pytest.approx(2.0, abs=1e-6)
    assert np.allclose(rec[3:6], n, atol=1e-6)
    assert rec[1] == pytest.approx(uv[0], abs=1e-6) and rec[2] == pytest.approx(uv[1], abs=1e-6)


def test_sphere_far_root_from_inside_and_range():
    b = oraclelib.OracleBuilder(1)
    m = b.lambertian(v3(1, 1, 1))
    b.sphere(v3(0, 0, 0), 2.0, m)
    hit, rec, _ = b.probe_hit(v3(0, 0, 0), v3(0, 0, 1))  # near root is negative -> far root
    assert hit and rec[0] == pytest.approx(2.0, abs=1e-6)
    assert np.allclose(rec[3:6], [0, 0, 1])  # normal always outward (sphere.cu:26)
    assert not b.probe_hit(v3(0, 0, 0), v3(0, 0, 1), 1e-3, 1.5)[0]  # outside [t_from, t_to]
    assert b.probe_hit(v3(0, 0, 0), v3(0, 0, 1), 1e-3, 2.0)[0]  # inclusive upper bound


# ---------------------------------------------------------------- k4: parallelogram.cu:19-21
def test_parallelogram_uv_corners_and_normal_faces_ray():
    b = oraclelib.OracleBuilder(1)
    m = b.lambertian(v3(1, 1, 1))
    # p0 <0,1>, p1 <1,1>, p2 <0,0>, p3 = p1+p2-p0 <1,0>
    b.parallelogram([v3(0, 1, 0), v3(1, 1, 0), v3(0, 0, 0)], m)
    e = 0.01
    for (x, y), (u, v) in {(e, 1 - 2 * e): (0, 1), (1 - 2 * e, 1 - e): (1, 1), (e, 2 * e): (0, 0),
                           (1 - e, 2 * e): (1, 0)}.items():
        hit, rec, _ = b.probe_hit(v3(x, y, 5), v3(0, 0, -1))
        assert hit and rec[0] == pytest.approx(5.0, abs=1e-5)
        assert rec[1] == pytest.approx(u, abs=0.05) and rec[2] == pytest.approx(v, abs=0.05)
        assert np.allclose(rec[3:6], [0, 0, 1])  # faces the ray (utils.cu:80)
        hit2, rec2, _ = b.probe_hit(v3(x, y, -5), v3(0, 0, 1))
        assert hit2 and np.allclose(rec2[3:6], [0, 0, -1])
    assert not b.probe_hit(v3(1.5, 0.5, 5), v3(0, 0, -1))[0]
    assert not b.probe_hit(v3(0.5, 0.5, 5), v3(1, 0, 0))[0]  # parallel: |det| < 1e-7


def test_parallelepiped_has_six_faces_of_a_box():
    b = oraclelib.OracleBuilder(1)
    m = b.lambertian(v3(1, 1, 1))
    b.parallelepiped([v3(0, 0, 0), v3(2, 0, 0), v3(0, 3, 0), v3(0, 0, 4)], m)
    c = np.array([1, 1.5, 2], dtype=np.float32)
    for d, t in [((1, 0, 0), 1), ((-1, 0, 0), 1), ((0, 1, 0), 1.5), ((0, -1, 0), 1.5), ((0, 0, 1), 2), ((0, 0, -1), 2)]:
        hit, rec, _ = b.probe_hit(c, np.array(d, dtype=np.float32))
        assert hit and rec[0] == pytest.approx(t, abs=1e-5)
        assert np.allclose(rec[3:6], -np.array(d))  # from inside, the normal opposes the ray


def test_parallelepiped_lengths_matches_point_form_under_identity():
    a = oraclelib.OracleBuilder(1)
    m = a.lambertian(v3(1, 1, 1))
    a.parallelepiped_lengths(v3(2, 3, 4), m, lambda p: p)
    b = oraclelib.OracleBuilder(1)
    m = b.lambertian(v3(1, 1, 1))
    b.parallelepiped([v3(0, 0, 0), v3(2, 0, 0), v3(0, 3, 0), v3(0, 0, 4)], m)
    rng = np.random.default_rng(0)
    for _ in range(200):
        o = rng.uniform(-6, 6, 3).astype(np.float32)
        d = rng.normal(size=3).astype(np.float32)
        ha, ra, _ = a.probe_hit(o, d)
        hb, rb, _ = b.probe_hit(o, d)
        assert ha == hb
        if ha:
            assert ra[0] == rb[0] and np.array_equal(ra[3:6], rb[3:6])


# ---------------------------------------------------------------- tie rules (quirk g7)
def test_list_first_entry_wins_ties():
    b = oraclelib.OracleBuilder(1)
    m0 = b.lambertian(v3(1, 0, 0))
    m1 = b.lambertian(v3(0, 1, 0))
    P = [v3(0, 1, 0), v3(1, 1, 0), v3(0, 0, 0)]
    b.parallelogram(P, m0)
    b.parallelogram(P, m1)
    hit, _, mat = b.probe_hit(v3(0.3, 0.6, 2), v3(0, 0, -1))
    assert hit and mat == m0  # hitable_list.cu:18: strict '<' keeps the first


def test_sky_is_an_object_at_1e9_and_true_miss_is_black():
    b = oraclelib.OracleBuilder(1)
    b.sky()
    hit, rec, _ = b.probe_hit(v3(0, 0, 0), v3(0, 1, 0))
    assert hit and rec[0] == 1e9
    assert not b.probe_hit(v3(0, 0, 0), v3(0, 1, 0), 1e-3, 1e8)[0]  # sky.cu:21
    # a world without Sky: misses are black (ray_tracing.cu:23-25)
    rgb, rays, _, total, _ = _render_custom(lambda bb: bb.sphere(v3(0, 0, -50), 1.0, bb.lambertian(v3(1, 1, 1))), 8, 8, 2)
    assert total == 8 * 8 * 2 and rgb.max() == 0.0


def _render_custom(fill, h, w, spp, depth=10, seed=3):
    b = oraclelib.OracleBuilder(seed)
    b.camera_pinhole(v3(0, 0, 0), v3(0, 0, -1), v3(0, 1, 0), scenes.PI_D / 2, w / h)
    fill(b)
    return b.render(h, w, spp, depth) + (b,)


def test_bvh_leaf_last_face_wins_ties():
    """bvh.cuh:129-133: 't <= t_to' inclusive and unconditional replace -> the LAST of two
    coincident faces supplies the record (seen through its texture coordinates)."""
    b = oraclelib.OracleBuilder(1)
    m = b.lambertian(v3(1, 1, 1))
    tri = np.array([[0, 0, 0], [1, 0, 0], [0, 1, 0]], dtype=np.float32)
    faces = np.stack([tri, tri])
    uvs = np.array([[0.1, 0.1, 0.1, 0.1, 0.1, 0.1], [0.9, 0.9, 0.9, 0.9, 0.9, 0.9]], dtype=np.float32)
    b.bvh(faces, m, uvs=uvs)
    hit, rec, _ = b.probe_hit(v3(0.2, 0.2, 1), v3(0, 0, -1))
    assert hit and rec[1] == pytest.approx(0.9, abs=1e-6) and rec[2] == pytest.approx(0.9, abs=1e-6)


# ---------------------------------------------------------------- k7: material single-ray checks
def test_metal_mirror_reflection():
    b = oraclelib.OracleBuilder(1)
    m = b.metal(v3(0.7, 0.6, 0.5), 0.0)
    d = np.array([1, -1, 0], dtype=np.float32) / f32(math.sqrt(2))
    st = oraclelib.rng_init(1, 1)[0].copy()
    before = st.copy()
    sc, out = b.probe_scatter(m, v3(0, 1, 0), d, 1.0, v3(0, 1, 0), st)
    assert sc and np.allclose(out[0:3], [0.7, 0.6, 0.5])
    assert np.allclose(out[6:9], np.array([1, 1, 0]) / math.sqrt(2), atol=1e-6)
    assert np.array_equal(st, before)  # fuzz 0 consumes no RNG (metal.cu:19-23)
    assert not b.probe_scatter(m, v3(0, 1, 0), d, 1.0, v3(0, -1, 0), st)[0]  # dot >= 0 -> false


def test_dielectric_straight_through_tir_and_no_rng():
    b = oraclelib.OracleBuilder(1)
    m = b.dielectric(v3(1, 1, 1), 1.5)
    st = oraclelib.rng_init(1, 1)[0].copy()
    before = st.copy()
    sc, out = b.probe_scatter(m, v3(0, 2, 0), v3(0, -1, 0), 1.0, v3(0, 1, 0), st)
    assert sc and np.allclose(out[6:9], [0, -1, 0], atol=1e-6)
    # Snell at 45 degrees entering glass: sin(theta_t) = sin(45)/1.5
    d = np.array([1, -1, 0], dtype=np.float32) / f32(math.sqrt(2))
    sc, out = b.probe_scatter(m, v3(0, 2, 0), d, 1.0, v3(0, 1, 0), st)
    assert sc and out[6] == pytest.approx(math.sin(math.pi / 4) / 1.5, abs=1e-6)
    # total internal reflection from inside -> refract returns 0 -> Scatter false (dielectric.cu:30)
    g = np.array([math.sin(1.2), math.cos(1.2), 0], dtype=np.float32)
    assert not b.probe_scatter(m, v3(0, 0, 0), g, 1.0, v3(0, 1, 0), st)[0]
    assert np.array_equal(st, before)


def test_lambertian_rejection_sampling_consumes_triples_and_stays_in_hemisphere():
    b = oraclelib.OracleBuilder(1)
    m = b.lambertian(v3(0.2, 0.4, 0.6))
    st = oraclelib.rng_init(11, 1)[0].copy()
    n = v3(0, 1, 0)
    for _ in range(200):
        d0 = int(st[0])
        sc, out = b.probe_scatter(m, v3(0, 1, 0), v3(0, -1, 0), 1.0, n, st)
        assert sc and np.allclose(out[0:3], [0.2, 0.4, 0.6])
        draws = ((int(st[0]) - d0) & 0xffffffff) // 362437
        assert draws % 3 == 0 and draws >= 3
        assert abs(np.linalg.norm(out[6:9]) - 1) < 1e-5 and out[7] >= -1e-6
    assert not b.probe_scatter(m, v3(0, 1, 0), v3(0, 1, 0), 1.0, n, st)[0]  # back side -> false


# ---------------------------------------------------------------- camera (camera.cu:24-38,57-70)
def test_camera_frame_and_rays():
    b = oraclelib.OracleBuilder(1)
    scenes.cornell_box(b, 1.0)
    cam = b.camera_get()
    pos, llc, hor, ver, u, v, w = cam
    assert np.allclose(w, [0, 0, -1]) and np.allclose(u, [-1, 0, 0]) and np.allclose(v, [0, 1, 0])
    hh = math.tan(math.pi * 2 / 9 / 2)
    assert np.allclose(hor, [-2 * hh, 0, 0], atol=1e-6) and np.allclose(ver, [0, 2 * hh, 0], atol=1e-6)
    r = b.probe_camera_ray(0.0, 0.0)  # image centre looks along -w
    assert np.allclose(r[0:3], pos) and np.allclose(r[3:6], [0, 0, 1], atol=1e-6)
    r = b.probe_camera_ray(-1.0, -1.0)  # lower-left corner
    want = llc - pos
    assert np.allclose(r[3:6], want / np.linalg.norm(want), atol=1e-6)


def test_defocus_camera_draws_two_numbers_from_a_square():
    b = oraclelib.OracleBuilder(1)
    b.camera_defocus(v3(0, 0, 5), v3(0, 0, 0), v3(0, 1, 0), scenes.PI_D / 4, 1.0, 0.5, 5.0)
    st = oraclelib.rng_init(2, 1)[0].copy()
    for _ in range(100):
        d0 = int(st[0])
        r = b.probe_camera_ray(0.1, -0.2, st)
        assert ((int(st[0]) - d0) & 0xffffffff) // 362437 == 2
        off = r[0:3] - np.array([0, 0, 5])
        # u = +x, v = +y for this frame; offsets lie in (0, r] x (0, r] (camera.cu:74-77)
        assert 0 < off[0] <= 0.25 and 0 < off[1] <= 0.25 and abs(off[2]) < 1e-6


# ---------------------------------------------------------------- k1: sky-only frame
def test_sky_only_frame_is_the_gradient():
    h, w, spp = 16, 24, 4
    rgb, rays, _, total, b = common.oracle_render("sky_only", h, w, spp, 10)
    assert total == h * w * spp and (rays == spp).all()
    for i in [0, 5, 15]:
        for j in [0, 11, 23]:
            x = (j + 0.5) / w * 2 - 1
            y = (h - i + 0.5) / h * 2 - 1  # quirk g1: (H - i), not (H - 1 - i)
            d = b.probe_camera_ray(x, y)[3:6]
            t = 0.5 * (d[1] + 1.0)
            want = np.sqrt(np.clip((1 - t) * np.ones(3) + t * np.array([0.5, 0.7, 1.0]), 0, 1))
            assert np.allclose(rgb[i, j], want, atol=0.03)
    assert np.all(rgb[..., 2] > 0.9999)


# ---------------------------------------------------------------- k5: furnace closed form
@pytest.mark.parametrize("rho", [0.5, 0.25])
def test_furnace_exact_values(rho):
    """A convex Lambertian sphere of albedo rho inside a radiance-1 enclosure: every path is
    'sphere then light' (value rho, 2 rays) or 'light' (value 1, 1 ray); all sums are exact in
    binary32, so each pixel must equal sqrt((rho*a + (spp-a)) / spp) with a = rays - spp."""
    h = w = 16
    spp = 8
    rgb, rays, _, _, _ = common.oracle_render("furnace", h, w, spp, 10, rho=rho)
    a = rays.astype(np.float32) - f32(spp)
    assert a.min() >= 0 and a.max() <= spp and (a > 0).any() and (a == 0).any()
    want = np.sqrt((f32(rho) * a + (f32(spp) - a)) / f32(spp), dtype=np.float32)
    for c in range(3):
        assert np.array_equal(rgb[..., c], want)


def test_depth_limit_zero_is_black_with_one_query_per_sample():
    rgb, rays, _, total, _ = common.oracle_render("cornell_box", 8, 8, 3, 0)
    assert rgb.max() == 0.0 and (rays == 3).all()  # ray_tracing.cu:22-26: query first, then the depth test


def test_spheres_scene_layout_counts_and_pixel0_stream():
    """scenes/spheres.cu: '.length() > 0.9' is GLM's component count, so all 22*22 small spheres
    exist (quirk g4): 4 + Sky + 484 = 489 entries; the layout consumed 6 or 7 draws per sphere
    from pixel 0's stream (quirk g5)."""
    b = oraclelib.OracleBuilder(10086)
    fresh = b.state0.copy()
    scenes.spheres(b, 1.0)
    draws = ((int(b.state0[0]) - int(fresh[0])) & 0xffffffff) // 362437
    assert 484 * 6 <= draws <= 484 * 7
    import rtmi
    p = rtmi.SceneBuilder(10086)
    scenes.spheres(p, 1.0)
    assert p.stats()["world"] == 489 and p.stats()["spheres"] == 488
