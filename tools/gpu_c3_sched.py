"""Ad-hoc: mesh frame time against the work-queue order."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, scenes.procedural_bunny_mesh()); b.commit()
R = rtmi.Renderer(b, 1024, 1024, spp, 10).init_rng()
for mode in (0, 1, 2):
    rtmi.lib().rtmi_set_schedule(mode)
    for it in range(2):
        R.init_rng(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render(); e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1)
    print(f"schedule {mode}: {ms:.1f} ms", flush=True)
rtmi.lib().rtmi_set_schedule(1)
cnt = R.ray_counts.view(-1, 64).sum(1).cpu().numpy()
import numpy as np
print("tile ray sums: max %d, p99 %d, median %d, mean %.0f, tiles %d" % (cnt.max(), np.percentile(cnt, 99), np.median(cnt), cnt.mean(), cnt.size))
