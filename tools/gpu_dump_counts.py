"""Ad-hoc: dump per-pixel ray counts of the C2 frame (64 spp, and the 2-spp estimate) in work-item
order to gpurun_out/ for tools/sim_schedule.py."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import numpy as np, torch, rtmi
from rtmi import scenes
b = rtmi.SceneBuilder(1024); scenes.cornell_box(b, 1.0); b.commit()
R = rtmi.Renderer(b, 1024, 1024, 64, 50).init_rng(); R.render(); torch.cuda.synchronize()
np.save(os.path.join(ROOT, "gpurun_out", "c2_counts_64spp.npy"), R.ray_counts.cpu().numpy().astype(np.uint32))
rtmi.lib().rtmi_set_schedule(0)
R2 = rtmi.Renderer(b, 1024, 1024, 2, 50).init_rng(); R2.render(); torch.cuda.synchronize()
np.save(os.path.join(ROOT, "gpurun_out", "c2_counts_2spp.npy"), R2.ray_counts.cpu().numpy().astype(np.uint32))
print("saved")
