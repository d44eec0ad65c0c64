#!/usr/bin/env python3
"""Diagnostic (GPU box): run the -DRTMI_CHECK_MARGINS build (tools/ab_build.sh check "-DRTMI_CHECK_MARGINS") over the
list scenes and report how many sampled closest-hit queries it answered a second time without the per-lane culls
and how many of those disagreed with the culled answer (must be 0).
usage: RTMI_LIB_PATH=ray-tracing-cuda_amd/lib/librtmi_check.so tools/gpu_check_margins.py"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rtmi
from rtmi import scenes
from rtmi.scenes import v3, PI_D
import bench
import test_gpu_round3 as t3

assert "check" in os.path.basename(rtmi.LIB_PATH), "run with RTMI_LIB_PATH=.../librtmi_check.so"
L = rtmi.lib()
out = {}


def counters(b):
    c = (C.c_ulonglong * 40)()
    assert L.rtmi_debug_counters(b.h, c, None) == 0
    return int(c[33]), int(c[34])


def run(tag, b, h, w, spp, depth):
    b.commit()
    R = rtmi.Renderer(b, h, w, spp, depth).init_rng()
    R.render(opts=rtmi.render_opts(schedule=0))  # (one launch: the probe pass would reset the counters)
    torch.cuda.synchronize()
    redone, bad = counters(b)
    out[tag] = {"rays": R.total_rays(), "re_done": redone, "disagreements": bad}


for name, side, spp, depth in (("cornell_box", 256, 64, 50), ("spheres", 256, 16, 8), ("birthday", 128, 16, 10)):
    b = bench.build_scene(rtmi.SceneBuilder(scenes.SCENE_SEEDS[name]), name, 1.0)
    run("%s_%dx%dx%d_d%d" % (name, side, side, spp, depth), b, side, side, spp, depth)
for n, boxes in ((100, 0), (300, 5)):
    b = rtmi.SceneBuilder(11)
    b.camera_pinhole(v3(0, 1.0, 3.0), v3(0, 0.6, -1), v3(0, 1, 0), PI_D / 3, 1.0)
    t3._quilt(b, n, np.random.default_rng(4000 + n + boxes), boxes)
    run("quilt_%d_%d" % (n, boxes), b, 96, 96, 16, 12)
# far views: a long list and a sphere cloud seen from 1e3 .. 1e4 away (the distance slack's regime)
for dist in (1e3, 1e4):
    b = rtmi.SceneBuilder(11)
    b.camera_pinhole(v3(0.3 * dist, 0.5 * dist, dist), v3(0, 0.8, -1.5), v3(0, 1, 0), float(2 * np.arctan(3.5 / dist)), 1.0)
    t3._quilt(b, 100, np.random.default_rng(4100), 0)
    run("quilt_100_from_%g" % dist, b, 96, 96, 8, 6)
    b = rtmi.SceneBuilder(11)
    b.camera_pinhole(v3(0.3 * dist, 0.5 * dist, dist), v3(0, 0.8, -1.5), v3(0, 1, 0), float(2 * np.arctan(3.5 / dist)), 1.0)
    rng = np.random.default_rng(7)
    mats = [b.lambertian(v3(*rng.uniform(0.2, 0.9, 3))) for _ in range(4)] + [b.metal(v3(0.9, 0.9, 0.9), 0.0)]
    for _ in range(200):
        b.sphere(v3(rng.uniform(-2.5, 2.5), rng.uniform(0.1, 2.2), rng.uniform(-4, 1)), float(rng.uniform(0.02, 0.3)), mats[int(rng.integers(0, 5))])
    b.sky()
    run("spheres_200_from_%g" % dist, b, 96, 96, 8, 6)
print(json.dumps(out, indent=1))
sys.exit(1 if any(v["disagreements"] for v in out.values()) else 0)
