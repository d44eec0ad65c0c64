"""Ad-hoc GPU probe (not a pytest file): parity of the HIP path vs the oracle on small frames."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import common

names = sys.argv[1:] or ["sky_only", "cornell_box", "spheres", "mixed", "furnace", "bunny", "birthday"]
for name in names:
    h, w, spp, depth = 64, 64, 8, 10
    kw = {"k_min": 64} if name == "bunny" else {}
    t = time.time(); o_rgb, o_rays, _, o_tot, _ = common.oracle_render(name, h, w, spp, depth, **kw); to = time.time() - t
    t = time.time(); g_rgb, g_rays, _, g_tot, _ = common.gpu_render(name, h, w, spp, depth, **kw); tg = time.time() - t
    same_rays = float((o_rays == g_rays).mean())
    print(f"{name:12s} relL2={common.rel_l2(g_rgb, o_rgb):.3e} maxabs={np.abs(g_rgb - o_rgb).max():.3e} "
          f"bitexact_px={float((g_rgb == o_rgb).all(axis=2).mean()):.4f} rays_equal={same_rays:.4f} "
          f"rays o={o_tot} g={g_tot}  t_oracle={to:.2f}s t_gpu={tg:.2f}s", flush=True)
