"""Ad-hoc: commit and render a ~1M-face mesh (builder time, memory, a small parity sample)."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch, rtmi
from rtmi import scenes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
t = time.time(); faces = scenes.procedural_bunny_mesh(n); print("mesh", faces.shape[0], "faces in %.1fs" % (time.time() - t), flush=True)
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, faces)
t = time.time(); b.commit(); print("commit %.2fs" % (time.time() - t), b.stats(), flush=True)
R = rtmi.Renderer(b, 512, 512, 64, 10).init_rng()
for it in range(2):
    R.init_rng(); torch.cuda.synchronize(); t = time.time(); R.render(); torch.cuda.synchronize(); dt = time.time() - t
print("render 512x512x64: %.1f ms, %d rays" % (dt * 1e3, R.total_rays()), flush=True)
if len(sys.argv) > 2:
    import oraclelib, common
    ids = np.random.default_rng(1).integers(0, 512 * 512, 12).astype(np.int32)
    ids = np.concatenate([ids, np.array([256 * 512 + 256, 250 * 512 + 260], dtype=np.int32)])
    ob = oraclelib.OracleBuilder(10086); scenes.bunny(ob, 1.0, faces)
    t = time.time(); o_rgb, o_rays, _, _ = ob.render(512, 512, 64, 10, pixel_ids=ids); o_rgb, o_rays = o_rgb.reshape(-1, 3)[ids], o_rays.reshape(-1)[ids]; print("oracle %d pixels in %.1fs" % (ids.size, time.time() - t))
    img, cnt = R.untile()
    g = img.cpu().numpy().reshape(-1, 3)[ids]; gr = cnt.cpu().numpy().reshape(-1)[ids]
    print("bit-exact:", bool(np.array_equal(g, o_rgb) and np.array_equal(gr.astype(np.uint32), o_rays)))
    for i in range(ids.size):
        if not (np.array_equal(g[i], o_rgb[i]) and gr[i] == o_rays[i]):
            print("pixel", int(ids[i]), divmod(int(ids[i]), 512), "gpu", g[i], int(gr[i]), "oracle", o_rgb[i], int(o_rays[i]))
