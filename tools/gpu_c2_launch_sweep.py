"""Ad-hoc: C2 frame time against blocks per CU (not a pytest file)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
b = rtmi.SceneBuilder(1024); scenes.cornell_box(b, 1.0); b.commit()
R = rtmi.Renderer(b, 1024, 1024, spp, 50).init_rng()
for per_cu in [int(x) for x in sys.argv[2:]] or [0, 4, 5, 6, 7, 8]:
    assert rtmi.lib().rtmi_set_launch(per_cu, 256) == 0
    best = 1e9
    for it in range(2):
        R.init_rng(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    rays = R.total_rays()
    print(f"blocks/CU {per_cu or 'auto'}: {best:.1f} ms, {rays/best/1e3:.0f} Mrays/s", flush=True)
rtmi.lib().rtmi_set_launch(0, 0)
