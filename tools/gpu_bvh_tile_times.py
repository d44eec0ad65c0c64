"""Ad-hoc: find the slow tiles of the bunny frame by rendering interleaved shards separately."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import rtmi
from rtmi import scenes
faces = scenes.procedural_bunny_mesh()
h = w = 1024; spp = 4
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, faces); b.commit()
G = 512
times = []
for r in range(G):
    R = rtmi.Renderer(b, h, w, spp, 10, rank=r, world_size=G).init_rng()
    torch.cuda.synchronize(); t = time.perf_counter(); R.render(); torch.cuda.synchronize()
    times.append(time.perf_counter() - t)
times = np.array(times) * 1e3
order = np.argsort(-times)
print("total %.1f ms, median %.3f, top:" % (times.sum(), np.median(times)), [(int(i), round(float(times[i]), 2)) for i in order[:8]])
# drill into the slowest shard: which of its tiles / pixels
r = int(order[0])
pm = rtmi.pixel_map(rtmi.make_frame(h, w, spp, rank=r, world_size=G))
R = rtmi.Renderer(b, h, w, spp, 10, rank=r, world_size=G).init_rng(); R.render(); torch.cuda.synchronize()
cnt = R.ray_counts.cpu().numpy()
tiles = pm.reshape(-1, 64)
for ti in range(tiles.shape[0]):
    px = tiles[ti][tiles[ti] >= 0]
    if px.size:
        print("tile", ti, "pixel0 (i,j)=", divmod(int(px[0]), w), "rays in tile", int(cnt[ti*64:(ti+1)*64].sum()), "max rays/pixel", int(cnt[ti*64:(ti+1)*64].max()))
