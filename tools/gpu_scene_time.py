"""Ad-hoc: time one of the scene programs on the GPU.  usage: gpu_scene_time.py <scene> <h> <w> <spp> <depth>"""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
name = sys.argv[1]; h, w, spp, depth = (int(x) for x in sys.argv[2:6])
b = rtmi.SceneBuilder(scenes.SCENE_SEEDS.get(name, 1024))
if name == "bunny":
    scenes.bunny(b, w / h, scenes.procedural_bunny_mesh())
elif name == "birthday":
    scenes.birthday(b, w / h, scenes.procedural_earthmap())
else:
    getattr(scenes, name)(b, w / h)
b.commit()
R = rtmi.Renderer(b, h, w, spp, depth).init_rng()
for it in range(3):
    R.init_rng(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1); rays = R.total_rays()
    print(f"{name} {h}x{w} spp{spp} depth{depth}: {ms:.2f} ms, {rays} rays, {rays/ms/1e3:.1f} Mrays/s", flush=True)
