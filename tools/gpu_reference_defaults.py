import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
def t(name, h, w, spp, depth, **kw):
    b = rtmi.SceneBuilder(scenes.SCENE_SEEDS.get(name, 1024))
    if name == "bunny": scenes.bunny(b, w / h, scenes.procedural_bunny_mesh())
    elif name == "birthday": scenes.birthday(b, w / h, scenes.procedural_earthmap())
    else: getattr(scenes, name)(b, w / h)
    b.commit()
    R = rtmi.Renderer(b, h, w, spp, depth).init_rng()
    pr = R.states.clone()
    res = []
    for it in range(4):
        R.states.copy_(pr)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); R.render(opts=rtmi.render_opts(**kw)); e1.record(); torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1))
    print("%-12s %dx%d x%d %-40s %.2f ms" % (name, h, w, spp, kw, min(res[1:])), flush=True)
for name, spp in (("bunny", 20), ("cornell_box", 200), ("spheres", 100), ("birthday", 200)):
    for kw in (dict(), dict(schedule=2), dict(schedule=2, plan=0), dict(schedule=0)):
        t(name, 720, 1280, spp, 10, **kw)
