"""One-off measurement (GPU box): what the scheduler's first pass costs a frame that is rendered from the queue -- each
frame unscheduled (image order, one launch), with a first pass of 1 sample per pixel and of 2 (the default), kernel
milliseconds, best of three renders after a warm-up.  usage: python3 tools/gpu_first_pass.py"""
import sys, os
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "ray-tracing-cuda_amd"))
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
        for it in range(4):
            R.states.copy_(pr)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); R.render(opts=rtmi.render_opts(**kw)); e1.record(); torch.cuda.synchronize()
            res.append(e0.elapsed_time(e1))
        out.append("%s %.2f" % (tag, min(res[1:])))
    print("%-12s %dx%d x%d d%d: %s ms" % (name, h, w, spp, depth, " | ".join(out)), flush=True)
V = [("unscheduled", dict(schedule=0)), ("first pass 1", dict(probe_spp=1)), ("first pass 2", dict(probe_spp=2))]
for name, h, w, spp, depth in (("cornell_box", 720, 1280, 200, 10), ("spheres", 720, 1280, 100, 10), ("birthday", 720, 1280, 200, 10),
                               ("bunny", 720, 1280, 20, 10), ("cornell_box", 2048, 2048, 100, 10), ("cornell_box", 1024, 1024, 64, 50),
                               ("spheres", 1024, 1024, 64, 8), ("spheres", 2048, 2048, 64, 8), ("birthday", 2048, 2048, 200, 10),
                               ("bunny", 1024, 1024, 512, 10), ("bunny", 1024, 1024, 64, 10)):
    t(name, h, w, spp, depth, V)
