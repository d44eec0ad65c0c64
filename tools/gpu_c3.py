"""Ad-hoc: time the C3-shaped frame (mesh scene, 1024x1024) at a given spp / depth."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 10
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0
per_cu = int(sys.argv[4]) if len(sys.argv) > 4 else 0
h = w = 1024
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, scenes.procedural_bunny_mesh()); b.commit()
rtmi.lib().rtmi_set_launch(per_cu, threads)
R = rtmi.Renderer(b, h, w, spp, depth).init_rng()
for it in range(2):
    R.init_rng(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); rays = R.total_rays()
    print(f"C3 1024x1024 spp{spp} depth{depth} threads{threads or 'dflt'} per_cu{per_cu or 'auto'}: {ms:.1f} ms, {rays} rays, {rays/ms/1e3:.1f} Mrays/s", flush=True)
