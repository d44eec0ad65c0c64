"""Experiment (GPU box): thin faces seen from far away, mesh first in the world list -- differing pixels GPU vs oracle
per (apex angle, distance).  Found the +infinity crossing-time marker bug of round 3 (trace_helpers.h)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import oraclelib, rtmi
from rtmi.scenes import v3

def sliver_fan(angle_deg, n=64, size=2e-3, seed=3):
    rng = np.random.default_rng(seed)
    faces = []
    a = np.deg2rad(angle_deg)
    for i in range(n):
        c = np.array([rng.uniform(-0.02, 0.02), rng.uniform(-1e-3, 1e-3), rng.uniform(-0.02, 0.02)])
        th = rng.uniform(0, 2 * np.pi)
        u = np.array([np.cos(th), 0.05 * rng.uniform(-1, 1), np.sin(th)])
        v = np.array([np.cos(th + a), 0.05 * rng.uniform(-1, 1), np.sin(th + a)])
        faces.append([c, c + size * u, c + size * v])
    return np.asarray(faces, dtype=np.float32)

for ang in (5.0, 1.4, 0.5, 0.1):
    for dist in (1e2, 1e3, 4e3, 2e4):
        faces = sliver_fan(ang)
        h, w, spp, depth = 28, 36, 4, 3
        res = []
        for make in (oraclelib.OracleBuilder, rtmi.SceneBuilder):
            b = make(9)
            pos = v3(0.3 * dist * 0.01, dist * np.sin(0.6), dist * np.cos(0.6))
            b.camera_pinhole(pos, v3(0.0, 0, 0.0), v3(0, 1, 0), float(2.0 * np.arctan(0.03 / dist)), w / h)
            b.bvh(faces, b.lambertian(v3(0.8, 0.8, 0.8)), k_min=8)
            b.sky()
            res.append(b)
        o, p = res
        o_rgb, o_rays, _, o_total = o.render(h, w, spp, depth)
        p.commit()
        R = rtmi.Renderer(p, h, w, spp, depth).init_rng(); R.render(); img, cnt = R.untile(); torch.cuda.synchronize()
        bad = int((cnt.cpu().numpy().astype(np.uint32) != o_rays).sum()) + int((img.cpu().numpy() != o_rgb).any(axis=2).sum())
        print("angle %.1f deg dist %g: oracle rays %d (hits beyond sky: %d), differing pixels %d" % (ang, dist, o_total, o_total - h*w*spp, bad), flush=True)
