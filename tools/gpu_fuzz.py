"""One-off campaign: many seeded random worlds (tests/test_gpu_random_scenes.py's generator), each
rendered by the oracle and through the C ABI, compared bit for bit.  usage: gpu_fuzz.py <first> <count>"""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oraclelib, rtmi
from rtmi.scenes import v3, PI_D
from test_gpu_random_scenes import random_world

first, count = int(sys.argv[1]), int(sys.argv[2])
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(1000 + seed)
    h, w = int(rng.integers(9, 41)), int(rng.integers(9, 57))
    spp, depth = int(rng.integers(1, 7)), int(rng.choice([1, 3, 10, 25, 64]))
    post = bool(rng.integers(0, 2))
    n_objects = int(rng.integers(3, 14))
    many = seed % 6 == 5
    defocus = seed % 4 == 3
    state = rng.bit_generator.state
    res = []
    for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
        rng.bit_generator.state = state
        b = make(77 + seed)
        if defocus:
            b.camera_defocus(v3(0, 1.0, 2.5), v3(0, 0.4, -2), v3(0, 1, 0), PI_D / 3, w / h, 0.2, 4.0)
        else:
            b.camera_pinhole(v3(0, 1.0, 2.5), v3(0, 0.4, -2), v3(0, 1, 0), PI_D / 3, w / h)
        random_world(b, rng, n_objects, many)
        res.append(b)
    o, p = res
    o_rgb, o_rays, _, o_total = o.render(h, w, spp, depth, post=post)
    p.commit()
    R = rtmi.Renderer(p, h, w, spp, depth, post).init_rng()
    R.render()
    img, cnt = R.untile()
    torch.cuda.synchronize()
    ok = R.total_rays() == o_total and np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays) and \
        np.array_equal(img.cpu().numpy(), o_rgb)
    if not ok:
        bad += 1
        print("MISMATCH seed", seed, h, w, spp, depth, post, n_objects, flush=True)
print("seeds %d..%d: %d mismatches, %.1fs" % (first, first + count - 1, bad, time.time() - t0), flush=True)

# ---- second campaign: triangle soups (many overlapping faces, tiny reference leaves, every material)
if len(sys.argv) > 3:
    bad = 0
    t0 = time.time()
    for seed in range(first, first + int(sys.argv[3])):
        rng = np.random.default_rng(50000 + seed)
        h, w = int(rng.integers(16, 49)), int(rng.integers(16, 65))
        spp, depth = int(rng.integers(1, 5)), int(rng.choice([2, 8, 20, 50]))
        n_mesh = int(rng.integers(1, 4))
        soups = []
        for _ in range(n_mesh):
            n = int(rng.choice([1, 5, 40, 300, 1500]))
            c = rng.uniform(-1.2, 1.2, 3); c[1] = abs(c[1]) * 0.5 + 0.2; c[2] -= 2.0
            spread = float(rng.choice([0.05, 0.4, 1.0]))
            base = rng.uniform(-spread, spread, (n, 1, 3)) + c
            size = float(rng.choice([0.05, 0.3, 0.9]))
            f = (base + rng.uniform(-size, size, (n, 3, 3))).astype(np.float32)
            if rng.integers(0, 3) == 0:  # duplicated faces: equal t, ties by reference index
                f = np.concatenate([f, f[: max(1, n // 3)]], 0)
            soups.append((f, int(rng.choice([1, 2, 3, 8, 64, 2048])), int(rng.integers(0, 4))))
        floor_sphere = bool(rng.integers(0, 2))
        res = []
        for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
            b = make(300 + seed)
            b.camera_pinhole(v3(0, 0.8, 1.8), v3(0, 0.4, -2), v3(0, 1, 0), PI_D / 3, w / h)
            ms = [b.lambertian(v3(0.8, 0.7, 0.6)), b.metal(v3(0.9, 0.9, 0.8), 0.1), b.dielectric(v3(1, 1, 1), 1.0),
                  b.dielectric(v3(0.9, 1, 0.9), 1.5)]
            for f, kmin, mi in soups:
                b.bvh(f, ms[mi], k_min=kmin)
            if floor_sphere:
                b.sphere(v3(0, -100.5, -1), 100.0, ms[0])
            else:
                b.parallelogram([v3(-30, -0.5, -30), v3(30, -0.5, -30), v3(-30, -0.5, 30)], ms[0])
            b.sky()
            res.append(b)
        o, p = res
        o_rgb, o_rays, _, o_total = o.render(h, w, spp, depth, post=False)
        p.commit()
        R = rtmi.Renderer(p, h, w, spp, depth, False).init_rng()
        R.render()
        img, cnt = R.untile()
        torch.cuda.synchronize()
        ok = R.total_rays() == o_total and np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays) and \
            np.array_equal(img.cpu().numpy(), o_rgb, equal_nan=True)
        if not ok:
            bad += 1
            print("SOUP MISMATCH seed", seed, h, w, spp, depth, [(s[0].shape[0], s[1], s[2]) for s in soups], flush=True)
    print("soup seeds %d..%d: %d mismatches, %.1fs" % (first, first + int(sys.argv[3]) - 1, bad, time.time() - t0), flush=True)

# ---- third campaign: small meshes seen from far away (tests/test_gpu_round3.py: far_view_world), the regime of
# binary32 false accepts outside the leaves' exact boxes; worlds with faces thinner than 1.8 degrees (their nodes widen the search boxes for them: scene.hip face_slack_exponent)
# (rtmi_scene_sliver_faces) are counted apart
if len(sys.argv) > 4:
    import test_gpu_round3 as t3
    bad = bad_sliver = n_sliver = 0
    t0 = time.time()
    for seed in range(first, first + int(sys.argv[4])):
        fill, cam, h, w, spp, depth, what = t3.far_view_world(seed)
        g, o = t3.render_pair(fill, h, w, spp, depth, post=False, seed=500 + seed, camera=cam)
        slivers = what["slivers"]
        n_sliver += slivers > 0
        ok = g[2] == o[2] and np.array_equal(g[1], o[1]) and np.array_equal(g[0], o[0], equal_nan=True)
        if not ok:
            if slivers:
                bad_sliver += 1
            else:
                bad += 1
            print("FAR MISMATCH seed", seed, h, w, spp, depth, what, flush=True)
    print("far-view seeds %d..%d: %d mismatches in worlds without sliver faces, %d in the %d worlds with them, %.1fs" %
          (first, first + int(sys.argv[4]) - 1, bad, bad_sliver, n_sliver, time.time() - t0), flush=True)
