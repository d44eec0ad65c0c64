"""C3 (mesh scene, 1024x1024 x512 spp, depth 10): N timed renders in one process -> median / min / max of the kernel time
(the frame's end is a serial chain whose length depends on which wave draws which outlier pixel: single runs scatter
by +-10 %).  usage: [RTMI_LIB_PATH=...] tools/gpu_c3_stats.py [N=9] [spp=512]"""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import statistics
import torch
import rtmi
from rtmi import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 9
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 512
b = rtmi.SceneBuilder(10086)
scenes.bunny(b, 1.0, scenes.procedural_bunny_mesh())
b.commit()
R = rtmi.Renderer(b, 1024, 1024, spp, 10).init_rng()
pristine = R.states.clone()
ms = []
for it in range(n + 1):
    R.states.copy_(pristine)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    R.render()
    e1.record()
    torch.cuda.synchronize()
    if it:
        ms.append(e0.elapsed_time(e1))
print("C3 x%d: median %.1f  min %.1f  max %.1f ms  (%s)  lib %s" % (n, statistics.median(ms), min(ms), max(ms),
      " ".join("%.0f" % m for m in ms), os.path.basename(os.environ.get("RTMI_LIB_PATH", "librtmi.so"))))
