#!/usr/bin/env python3
"""Diagnostic (GPU box, stats build level 9: `RTMI_STATS_LEVEL=9 tools/mesh_stats.sh build` on the dev box first):
where the time of ONE shard of a list-scene frame goes -- per-wave start / end / queries and per-pixel ray counts.
usage: RTMI_LIB_PATH=.../librtmi_stats.so tools/gpu_shard_waves.py [workload c4] [r/G 0/8] [spp] [probe_spp] [bpc]
Writes gpurun_out/shard_waves_<tag>.npz and prints the digest."""
import ctypes as C
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import rtmi
import bench

wl = sys.argv[1] if len(sys.argv) > 1 else "c4"
r, G = (int(x) for x in (sys.argv[2] if len(sys.argv) > 2 else "0/8").split("/"))
w = dict(bench.WORKLOADS[wl])
if len(sys.argv) > 3 and int(sys.argv[3]) > 0:
    w["spp"] = int(sys.argv[3])
prio = int(sys.argv[4]) if len(sys.argv) > 4 else 0
bpc = int(sys.argv[5]) if len(sys.argv) > 5 else 0
from rtmi import scenes
seed = scenes.SCENE_SEEDS[w["scene"]]
scene = bench.build_scene(rtmi.SceneBuilder(seed), w["scene"], 1.0).commit()
R = rtmi.Renderer(scene, w["size"], w["size"], w["spp"], w["depth"], True, rank=r, world_size=G).init_rng()
opts = rtmi.render_opts(probe_spp=prio, blocks_per_cu=bpc)
shape = R.launch_shape(opts)
pristine = R.states.clone()
R.render(opts=opts)
torch.cuda.synchronize()
R.states.copy_(pristine)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
R.render(opts=opts)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
ws = np.zeros((16384, 16), dtype=np.uint64)
L = rtmi.lib()
L.rtmi_debug_wave_stats.argtypes = [C.c_void_p, C.c_size_t]
assert L.rtmi_debug_wave_stats(ws.ctypes.data_as(C.c_void_p), ws.nbytes) == 0
n_waves = shape["blocks"] * shape["threads"] // 64
ws = ws[:n_waves]
counts = R.ray_counts.cpu().numpy().astype(np.int64)
life, q, t0, t1 = ws[:, 0].astype(np.float64), ws[:, 11].astype(np.float64), ws[:, 12].astype(np.float64), ws[:, 13].astype(np.float64)
start = t0.min()
end = (t1 - start) / 100.0  # microseconds (100 MHz)
beg = (t0 - start) / 100.0
rays = float(counts.sum())
tile = counts.reshape(-1, 64)
d = {"workload": wl, "shard": [r, G], "spp": w["spp"], "probe_spp": prio, "shape": shape, "kernel_ms_stats_build": ms,
     "rays": rays, "waves": int(n_waves), "wave_queries_sum": float(q.sum()),
     "lane_utilisation": rays / (q.sum() * 64.0),
     "wave_end_us_percentiles": {p: float(np.percentile(end, p)) for p in (1, 10, 25, 50, 75, 90, 99, 100)},
     "wave_begin_us_max": float(beg.max()),
     "wave_queries_percentiles": {p: float(np.percentile(q, p)) for p in (1, 10, 50, 90, 99, 100)},
     "cycles_per_query_percentiles": {p: float(np.percentile(life / np.maximum(q, 1), p)) for p in (1, 10, 50, 90, 99)},
     "pixel_rays": {"mean": float(counts.mean()), "max": int(counts.max()),
                    "percentiles": {p: float(np.percentile(counts, p)) for p in (1, 10, 25, 50, 75, 90, 99)}},
     "tile_max_over_mean": {p: float(np.percentile(tile.max(1) / np.maximum(tile.mean(1), 1), p)) for p in (10, 50, 90, 99)},
     "tile_rays_over_mean_tile": {p: float(np.percentile(tile.sum(1) / tile.sum(1).mean(), p)) for p in (1, 10, 25, 50, 75, 90, 99, 100)}}
# the frame's timeline: live waves and iteration rate over time (20 bins)
T = end.max()
bins = np.linspace(0, T, 21)
d["live_waves_at"] = [int(((beg <= t) & (end > t)).sum()) for t in bins[:-1]]
tag = "%s_s%dof%d_p%d_b%d%s" % (wl, r, G, prio, bpc, os.environ.get("RTMI_TOOL_TAG", ""))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "shard_waves_%s.npz" % tag), ws=ws, counts=counts.astype(np.uint32))
print(json.dumps(d))
