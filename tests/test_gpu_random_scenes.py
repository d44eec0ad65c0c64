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
