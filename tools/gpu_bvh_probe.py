"""Ad-hoc timing probe for the BVH path (not a pytest file)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rtmi, common
from rtmi import scenes

faces = scenes.procedural_bunny_mesh()
for (h, w, spp, kmin) in [(256, 256, 16, 2048), (256, 256, 16, 100000), (256, 256, 16, 128), (1024, 1024, 4, 2048), (1024, 1024, 16, 2048)]:
    b = rtmi.SceneBuilder(10086); scenes.bunny(b, w / h, faces, k_min=kmin); b.commit()
    R = rtmi.Renderer(b, h, w, spp, 10).init_rng()
    R.render(); torch.cuda.synchronize()
    R.init_rng()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); rays = R.total_rays()
    _, cnt = R.untile(); cnt = cnt.cpu().numpy()
    print(f"{h}x{w} spp{spp} kmin{kmin}: {ms:.1f} ms, {rays} rays, {rays/ms/1e3:.1f} Mrays/s, stats {b.stats()['bvh_nodes']} ref nodes; pixels with bounces {(cnt>spp).mean():.3f}", flush=True)
