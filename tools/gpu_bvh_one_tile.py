"""Ad-hoc: render ONE tile of the bunny frame (a single wavefront) so per-wave PMC counters
are easy to read.  usage: gpu_bvh_one_tile.py <tile_row> <tile_col> [spp]"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
ty, tx = int(sys.argv[1]), int(sys.argv[2]); spp = int(sys.argv[3]) if len(sys.argv) > 3 else 16
h = w = 1024
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, scenes.procedural_bunny_mesh()); b.commit()
n_tiles = (h // 8) * (w // 8)
R = rtmi.Renderer(b, h, w, spp, 10, rank=ty * (w // 8) + tx, world_size=n_tiles).init_rng()
R.render(); torch.cuda.synchronize(); R.init_rng()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
print("tile", ty, tx, "spp", spp, "%.2f ms" % e0.elapsed_time(e1), R.total_rays(), "rays; max rays/pixel", int(R.ray_counts.max()))
