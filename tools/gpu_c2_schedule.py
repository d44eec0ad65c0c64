"""Ad-hoc: does the longest-first schedule (2-spp probe + tile order) pay for itself on the C2 frame?"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
b = rtmi.SceneBuilder(1024); scenes.cornell_box(b, 1.0); b.commit()
R = rtmi.Renderer(b, 1024, 1024, 1024, 50).init_rng()
pristine = R.states.clone()
for sched in (1, 0, 1, 0):
    o = rtmi.render_opts(schedule=sched)
    ts = []
    for i in range(3):
        R.states.copy_(pristine)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render(opts=o); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1))
    print("schedule=%d: %s ms (rays %d)" % (sched, " ".join("%.1f" % t for t in ts), R.total_rays()))
