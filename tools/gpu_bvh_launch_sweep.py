"""Ad-hoc probe: mesh-scene frame time against the launch shape (not a pytest file)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import rtmi
from rtmi import scenes

faces = scenes.procedural_bunny_mesh()
h = w = 1024; spp = 16
b = rtmi.SceneBuilder(10086); scenes.bunny(b, w / h, faces, k_min=2048); b.commit()
R = rtmi.Renderer(b, h, w, spp, 10).init_rng()
for threads in (256, 128, 64):
    for per_cu in (0, 1, 2, 4, 8):
        rc = rtmi.lib().rtmi_set_launch(per_cu, threads)
        assert rc == 0
        R.init_rng(); R.render(); torch.cuda.synchronize()
        R.init_rng()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1); rays = R.total_rays()
        print(f"threads {threads} blocks/CU {per_cu or 'auto'}: {ms:.1f} ms, {rays/ms/1e3:.1f} Mrays/s", flush=True)
rtmi.lib().rtmi_set_launch(0, 0)
