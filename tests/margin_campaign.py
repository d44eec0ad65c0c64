#!/usr/bin/env python3
"""The every-query margin check (GPU box; run by tests/test_gpu_margins.py in a process of its own, because it loads a
DIAGNOSTIC build of the library: `make -C ray-tracing-cuda_amd/csrc check1`, i.e. -DRTMI_CHECK_MARGINS
-DRTMI_CHECK_EVERY=1, which __graft_entry__.build() builds next to the product).  That build answers every
closest-hit query a second time WITHOUT any cull, padded bound or distance slack -- the plain walk of the world list
(hitable_list.cu:7-25), the plain sphere loop, and for meshes the reference's own walk of its own tree (bvh.cuh:123-158,
bvh.cu:6-30) -- and counts the rays re-done (rtmi_debug_counters word 33) and the disagreements (word 34, must be 0).
Worlds: the four scene programs' worlds at small frames, quilts of 100 / 300 pairs near and from 1e3 / 1e4 away, sphere
clouds from afar, the bunny stand-in, and seeded adversarial worlds of tests/test_gpu_round3.py -- far views with and
without slivers, grazing sheets, needle meshes, needle lists, far sphere clouds (counts: RTMI_CHECK_FAR / _GRAZE /
_NEEDLES / _SPHERES / _NEEDLE_LISTS).  Prints one JSON object; exit status 1 on any disagreement.
usage: RTMI_LIB_PATH=ray-tracing-cuda_amd/lib/librtmi_check1.so python tests/margin_campaign.py [scenes|meshes|lists]"""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))  # (tests/ -> the repository)
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


SCENES_EVERY = "scenes" in sys.argv[1:]  # (with a -DRTMI_CHECK_EVERY=1 build: every query of the four scene programs' worlds)
if SCENES_EVERY:
    big = os.environ.get("RTMI_CHECK_SCALE") == "big"
    for name, side, spp, depth in ((("cornell_box", 1024, 64, 50), ("spheres", 1024, 16, 8), ("birthday", 1024, 16, 10), ("bunny", 384, 16, 10))
                                   if big else (("cornell_box", 256, 64, 50), ("spheres", 256, 16, 8), ("birthday", 256, 16, 10), ("bunny", 192, 8, 10))):
        b = bench.build_scene(rtmi.SceneBuilder(scenes.SCENE_SEEDS[name]), name, 1.0)
        run("%s_%dx%dx%d_d%d_every_query" % (name, side, side, spp, depth), b, side, side, spp, depth)
    print(json.dumps(out, indent=1))
    sys.exit(1 if any(v["disagreements"] for v in out.values()) else 0)
ONLY_MESHES = "meshes" in sys.argv[1:]  # (with a -DRTMI_CHECK_EVERY=1 build: every query of the small mesh worlds)
for name, side, spp, depth in (() if ONLY_MESHES else (("cornell_box", 256, 64, 50), ("spheres", 256, 16, 8), ("birthday", 128, 16, 10))):
    b = bench.build_scene(rtmi.SceneBuilder(scenes.SCENE_SEEDS[name]), name, 1.0)
    run("%s_%dx%dx%d_d%d" % (name, side, side, spp, depth), b, side, side, spp, depth)
for n, boxes in (() if ONLY_MESHES else ((100, 0), (300, 5))):
    b = rtmi.SceneBuilder(11)
    b.camera_pinhole(v3(0, 1.0, 3.0), v3(0, 0.6, -1), v3(0, 1, 0), PI_D / 3, 1.0)
    t3._quilt(b, n, np.random.default_rng(4000 + n + boxes), boxes)
    run("quilt_%d_%d" % (n, boxes), b, 96, 96, 16, 12)
# far views: a long list and a sphere cloud seen from 1e3 .. 1e4 away (the distance slack's regime)
for dist in (() if ONLY_MESHES else (1e3, 1e4)):
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
# ---- meshes: the second answer is the reference's own walk of its tree (closest_hit.h: bvh_reference_walk)
if "lists" not in sys.argv[1:]:
    b = bench.build_scene(rtmi.SceneBuilder(scenes.SCENE_SEEDS["bunny"]), "bunny", 1.0)
    if ONLY_MESHES:
        run("bunny_48x48x2_d10", b, 48, 48, 2, 10)
    else:
        run("bunny_128x128x8_d10", b, 128, 128, 8, 10)


    def campaign(tag, worlds):
        tot = {"worlds": 0, "re_done": 0, "disagreements": 0, "worlds_disagreeing": []}
        for seed, make in worlds:
            b = rtmi.SceneBuilder(500 + seed)
            h, w, spp, depth = make(b)
            b.commit()
            R = rtmi.Renderer(b, h, w, spp, depth).init_rng()
            R.render(opts=rtmi.render_opts(schedule=0))
            torch.cuda.synchronize()
            redone, bad = counters(b)
            tot["worlds"] += 1
            tot["re_done"] += redone
            tot["disagreements"] += bad
            if bad:
                tot["worlds_disagreeing"].append(seed)
        out[tag] = tot


    def far(seed):
        fill, cam, h, w, spp, depth, what = t3.far_view_world(seed)

        def make(b):
            cam(b)
            fill(b)
            return h, w, max(spp, 4), depth
        return make, what["slivers"]


    def graze(seed):
        fill, h, w, spp, depth = t3.grazing_world(seed)

        def make(b):
            fill(b)
            return h, w, spp, depth
        return make


    def needle_list(seed):
        fill, h, w, spp, depth = t3.needle_list_world(seed)

        def make(b):
            fill(b)
            return h, w, spp, depth
        return make


    def sphere_cloud(seed):
        fill, h, w, spp, depth = t3.far_sphere_cloud(seed)

        def make(b):
            fill(b)
            return h, w, spp, depth
        return make


    def needle(seed):
        fill, h, w, spp, depth = t3.needle_world(seed)

        def make(b):
            fill(b)
            return h, w, spp, depth
        return make

    n_far = int(os.environ.get("RTMI_CHECK_FAR", "60"))
    fars = [(s,) + far(s) for s in range(n_far)]
    campaign("far_views_without_slivers", [(s, m) for s, m, sl in fars if not sl])
    campaign("far_views_with_slivers", [(s, m) for s, m, sl in fars if sl])
    campaign("far_views_known_sliver_cases", [(s, far(s)[0]) for s in (1527, 1674, 1675, 1774)])
    campaign("grazing_views", [(s, graze(s)) for s in range(int(os.environ.get("RTMI_CHECK_GRAZE", "40")))])
    campaign("needles", [(s, needle(s)) for s in range(int(os.environ.get("RTMI_CHECK_NEEDLES", "40")))])
    campaign("far_sphere_clouds", [(s, sphere_cloud(s)) for s in range(int(os.environ.get("RTMI_CHECK_SPHERES", "40")))])
    campaign("needle_lists", [(s, needle_list(s)) for s in range(int(os.environ.get("RTMI_CHECK_NEEDLE_LISTS", "40")))])
print(json.dumps(out, indent=1))
hard = [k for k, v in out.items() if v["disagreements"]]
sys.exit(1 if hard else 0)
