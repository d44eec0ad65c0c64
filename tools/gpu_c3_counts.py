"""Ad-hoc: per-pixel ray-count distribution of the C3 frame (who is the longest serial chain?)."""
import sys, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import numpy as np, torch, rtmi
from rtmi import scenes
spp = int(sys.argv[1]) if len(sys.argv) > 1 else 512
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, scenes.procedural_bunny_mesh()); b.commit()
R = rtmi.Renderer(b, 1024, 1024, spp, 10).init_rng()
R.render(); torch.cuda.synchronize()
img, cnt = R.untile()
c = cnt.cpu().numpy().astype(np.int64)
print("rays total", c.sum(), "max/pixel", c.max(), "p99.9", np.percentile(c, 99.9), "p99", np.percentile(c, 99), "median", np.median(c))
t = c.reshape(128, 8, 128, 8).sum(axis=(1, 3))
print("tile max", t.max(), "tile p99", np.percentile(t, 99), "tile median", np.median(t), "tiles >= 2x mean", (t >= 2 * t.mean()).sum(), "of", t.size)
for thr in (1000, 2000, 3000, 4000, 5000):
    print("pixels with >=", thr, "rays:", (c >= thr).sum())
if len(sys.argv) > 2:  # dump the full-spp counts and the 2-spp probe counts for offline scheduling studies
    out = sys.argv[2]
    np.save(os.path.join(out, "c3_counts.npy"), c.astype(np.int32))
    P = rtmi.Renderer(b, 1024, 1024, 2, 10).init_rng()
    P.render(); torch.cuda.synchronize()
    _, pc = P.untile()
    np.save(os.path.join(out, "c3_probe.npy"), pc.cpu().numpy().astype(np.int32))
