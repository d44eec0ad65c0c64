import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "ray-tracing-cuda_amd"))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "tools"))
import torch, rtmi
from rtmi import scenes
def t(name, h, w, spp, depth, variants):
    b = rtmi.SceneBuilder(scenes.SCENE_SEEDS.get(name, 1024))
    if name == "bunny": scenes.bunny(b, w / h, scenes.procedural_bunny_mesh())
    elif name == "birthday": scenes.birthday(b, w / h, scenes.procedural_earthmap())
    else: getattr(scenes, name)(b, w / h)
    b.commit()
    R = rtmi.Renderer(b, h, w, spp, depth).init_rng()
    pr = R.states.clone()
    out = []
    for tag, kw in variants:
        res = []
        for it in range(5):
            R.states.copy_(pr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); R.render(opts=rtmi.render_opts(**kw)); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1))
        out.append("%s %.2f" % (tag, sorted(res[1:])[1]))
    print("%-12s %dx%d x%d d%d: %s ms" % (name, h, w, spp, depth, " | ".join(out)), flush=True)
V = [("default", dict()), ("unscheduled", dict(schedule=0)), ("scheduled", dict(schedule=2, plan=0))]
for name, h, w, depth in (("cornell_box", 1024, 1024, 50), ("cornell_box", 720, 1280, 10), ("spheres", 1024, 1024, 8), ("birthday", 1024, 1024, 10), ("cornell_box", 2048, 2048, 10)):
    for spp in (8, 16, 24, 32, 48, 64, 128):
        t(name, h, w, spp, depth, V)
