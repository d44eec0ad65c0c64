"""Ad-hoc: frame times of small synthetic worlds that select the kernel variants no benchmark workload uses --
render_kernel<1> (a short run of spheres), <3> (spheres + world-list triangles: scenes.mixed), <7> (a grouped sphere run
+ triangles).  usage: gpu_variant_time.py [size=768] [spp=128]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import numpy as np
import torch
import rtmi
from rtmi import scenes
from rtmi.scenes import v3, PI_D

size = int(sys.argv[1]) if len(sys.argv) > 1 else 768
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128


def few_spheres(b):
    rng = np.random.default_rng(3)
    b.camera_pinhole(v3(0, 1.2, 4.0), v3(0, 0.5, -1), v3(0, 1, 0), PI_D / 3, 1.0)
    b.sphere(v3(0, -1000, 0), 1000.0, b.lambertian(v3(0.5, 0.5, 0.5)))
    for i in range(20):
        m = b.lambertian(v3(*rng.uniform(0.2, 0.9, 3))) if i % 3 else (b.metal(v3(0.8, 0.8, 0.8), 0.1) if i % 2 else b.dielectric(v3(1, 1, 1), 1.5))
        b.sphere(v3(rng.uniform(-3, 3), rng.uniform(0.2, 0.6), rng.uniform(-4, 1)), float(rng.uniform(0.2, 0.5)), m)
    b.sky()


def grouped_plus_tris(b):
    rng = np.random.default_rng(4)
    b.camera_pinhole(v3(0, 1.5, 5.0), v3(0, 0.5, -1), v3(0, 1, 0), PI_D / 3, 1.0)
    b.parallelogram([v3(-6, 0, -8), v3(6, 0, -8), v3(-6, 0, 3)], b.lambertian(v3(0.5, 0.6, 0.5)))
    for k in range(5):
        b.parallelepiped_lengths(v3(0.6, 0.6 + 0.2 * k, 0.6), b.lambertian(v3(*rng.uniform(0.3, 0.9, 3))),
                                 lambda p, k=k: p + np.array([-3 + 1.5 * k, 0, -5], dtype=np.float32))
    for i in range(200):
        b.sphere(v3(rng.uniform(-4, 4), rng.uniform(0.1, 0.3), rng.uniform(-4, 2)), float(rng.uniform(0.08, 0.2)),
                 b.lambertian(v3(*rng.uniform(0.2, 0.9, 3))))
    b.sky()


for name, fill in (("few_spheres", few_spheres), ("mixed", lambda b: scenes.mixed(b, 1.0)), ("grouped_plus_tris", grouped_plus_tris)):
    b = rtmi.SceneBuilder(7)
    fill(b)
    b.commit()
    R = rtmi.Renderer(b, size, size, spp, 10).init_rng()
    pristine = R.states.clone()
    times = []
    for it in range(4):
        R.states.copy_(pristine)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
        times.append(e0.elapsed_time(e1))
    print("%-18s %dx%d x%d: %.2f ms (min of %s), %.0f Mrays/s, lanes %d" % (name, size, size, spp, min(times[1:]), ["%.2f" % t for t in times[1:]],
          R.total_rays() / min(times[1:]) / 1e3, R.launch_shape()["blocks"] * R.launch_shape()["threads"]), flush=True)
