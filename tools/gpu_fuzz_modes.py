"""One-off campaign (GPU box): the scheduling modes of round 4 on seeded random list worlds (tests/test_gpu_random_scenes.py's
generator) at frames large enough to be scheduled (the fuzz worlds of tools/gpu_fuzz.py are a few hundred pixels: thin
frames, no probe).  Every world is rendered by the plain queue in image order and then under every other mode -- longest-
first queue, wave priorities, planned chains (also with one workgroup per CU: long chains and many take-overs), thin
frames -- and all results must agree bit for bit (image, per-pixel ray counts, final RNG states, ray total); every
`oracle_every`-th world is also rendered by the oracle.  usage: gpu_fuzz_modes.py <first> <count> [oracle_every=8]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch
import oraclelib
import rtmi
from rtmi.scenes import v3, PI_D
from test_gpu_random_scenes import random_world

first, count = int(sys.argv[1]), int(sys.argv[2])
oracle_every = int(sys.argv[3]) if len(sys.argv) > 3 else 8
MODES = (dict(schedule=0, plan=0, wave_priority=0, lane_stride=1),
         dict(schedule=2, plan=0, wave_priority=0, lane_stride=1),
         dict(schedule=2, plan=0, wave_priority=16, lane_stride=1),
         dict(schedule=2, plan=2, wave_priority=16, lane_stride=1),
         dict(schedule=2, plan=2, wave_priority=2, lane_stride=1, blocks_per_cu=1),
         dict(schedule=2, plan=2, wave_priority=64, lane_stride=1, probe_spp=1, blocks_per_cu=2),
         dict(schedule=2, plan=0, wave_priority=8, lane_stride=4),
         dict(schedule=2, plan=2, wave_priority=16, lane_stride=1, first_pass=0),
         dict(schedule=2, plan=0, wave_priority=16, lane_stride=1, first_pass=3),
         dict())
bad = 0
t0 = time.time()
for seed in range(first, first + count):
    rng = np.random.default_rng(9000 + seed)
    h, w = int(rng.integers(64, 161)), int(rng.integers(64, 201))
    spp, depth = int(rng.choice([64, 64, 96, 128])), int(rng.choice([3, 10, 25]))
    post = bool(rng.integers(0, 2))
    n_objects = int(rng.integers(3, 14))
    many = seed % 3 == 2
    state = rng.bit_generator.state

    def world(make):
        rng.bit_generator.state = state
        b = make(77 + seed)
        b.camera_pinhole(v3(0, 1.0, 2.5), v3(0, 0.4, -2), v3(0, 1, 0), PI_D / 3, w / h)
        random_world(b, rng, n_objects, many)
        return b

    p = world(rtmi.SceneBuilder).commit()
    want = None
    for kw in MODES:
        R = rtmi.Renderer(p, h, w, spp, depth, post).init_rng()
        R.render(opts=rtmi.render_opts(**kw))
        R.check()
        got = (R.tiles.cpu().numpy(), R.ray_counts.cpu().numpy(), R.states.cpu().numpy(), R.total_rays())
        if want is None:
            want = got
        elif got[3] != want[3] or not all(np.array_equal(x, y, equal_nan=True) for x, y in zip(got[:3], want[:3])):
            bad += 1
            print("MISMATCH seed", seed, kw, flush=True)
    if oracle_every and seed % oracle_every == 0:
        o = world(oraclelib.OracleBuilder)
        o_rgb, o_rays, _, o_total = o.render(h, w, spp, depth, post=post)
        R = rtmi.Renderer(p, h, w, spp, depth, post).init_rng()
        R.render()
        img, cnt = R.untile()
        torch.cuda.synchronize()
        if not (R.total_rays() == o_total and np.array_equal(cnt.cpu().numpy().astype(np.uint32), o_rays) and
                np.array_equal(img.cpu().numpy(), o_rgb, equal_nan=True)):
            bad += 1
            print("ORACLE MISMATCH seed", seed, flush=True)
print("mode seeds %d..%d (%d modes each, oracle every %d): %d mismatches, %.1fs" % (first, first + count - 1, len(MODES), oracle_every, bad, time.time() - t0), flush=True)
sys.exit(1 if bad else 0)
