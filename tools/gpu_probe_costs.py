"""Ad-hoc: the scheduler's per-pixel probe costs of the C3 frame next to the full-frame ray counts and,
per tile, what the frame's wave-query counts make of it.  Dumps npy files for offline study."""
import sys, os, ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))
import numpy as np, torch, rtmi
from rtmi import scenes
out = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out"
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, scenes.procedural_bunny_mesh()); b.commit()
R = rtmi.Renderer(b, 1024, 1024, 512, 10).init_rng()
need = rtmi.lib().rtmi_render_scratch_bytes(C.byref(R.frame))
scratch = torch.zeros(need // 4 + 16, dtype=torch.int32, device=R.device)
o = rtmi.render_opts(schedule=2)
o.d_scratch = scratch.data_ptr(); o.scratch_bytes = need
R.render(opts=o); torch.cuda.synchronize()
n = R.items
probe = scratch[n * 6: n * 7].clone()
img, cnt = R.untile()
_, pc = R.untile(all_counts=probe)
c = cnt.cpu().numpy(); p = pc.cpu().numpy()
np.save(os.path.join(out, "c3_counts.npy"), c.astype(np.int32)); np.save(os.path.join(out, "c3_probe_cost.npy"), p.astype(np.int32))
tp = p.reshape(128, 8, 128, 8).sum(axis=(1, 3)); tc = c.reshape(128, 8, 128, 8).sum(axis=(1, 3))
print("probe cost per pixel: max", p.max(), "mean %.1f" % p.mean(), " tile cost: max", tp.max(), "mean %.0f" % tp.mean(),
      "p50 %.0f p90 %.0f p99 %.0f" % tuple(np.percentile(tp, [50, 90, 99])))
print("tiles >= 2x mean:", (tp >= 2 * tp.mean()).sum(), " >= 1.5x p90:", (tp >= 1.5 * np.percentile(tp, 90)).sum(),
      " >= 2x p90:", (tp >= 2 * np.percentile(tp, 90)).sum())
