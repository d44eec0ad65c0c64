#!/bin/bash
# Diagnostic (GPU box, stats build): how many pairs survive the cull of the world-list scan on C2-like frames.
ROOT=$(cd "$(dirname "$0")/.." && pwd)
# RTMI_TOOL_BPC / RTMI_TOOL_THREADS: workgroups per CU / lanes per workgroup (1 / 64 = every wave alone on its CU);
# RTMI_TOOL_SIZE: frame side; RTMI_TOOL_LEVEL: stats build (1 sections, 2 + the culled scan's parts)
RTMI_LIB_PATH=$ROOT/ray-tracing-cuda_amd/lib/librtmi_stats${RTMI_TOOL_LEVEL:-1}.so python3 - "${1:-cornell_box}" "${2:-64}" <<'PY'
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
name, spp = sys.argv[1], int(sys.argv[2])
b = rtmi.SceneBuilder(1024); getattr(scenes, name)(b, 1.0) if name != "birthday" else scenes.birthday(b, 1.0, scenes.procedural_earthmap(256, 512)); b.commit()
side = int(os.environ.get("RTMI_TOOL_SIZE", "1024"))
R = rtmi.Renderer(b, side, side, spp, 50).init_rng()
opts = rtmi.render_opts(blocks_per_cu=int(os.environ.get("RTMI_TOOL_BPC", "0")), threads_per_block=int(os.environ.get("RTMI_TOOL_THREADS", "0")))
print("launch", R.launch_shape(opts))
R.render(opts=opts); torch.cuda.synchronize()
out = (C.c_ulonglong * 40)()
rtmi.lib().rtmi_debug_counters(b.h, out, None)
rays, wq, bits, iters, crays = out[1], out[4], out[30], out[31], out[32]
life = out[26]
names = ["gen", "list", "search", "replay", "shade", "cull", "tasks+tests+fold", "fold"]
for i, n in enumerate(names):
    if out[17 + i]:
        print("  %-18s %5.1f%% of wave time, %7.0f cycles per wave_query" % (n, 100.0 * out[17 + i] / life, out[17 + i] / max(wq, 1)))
print("  shade: up to the material record (divergent lanes stamp separately: upper bound) %.0f, up to before the fold %.0f cycles per wave_query" % (out[25] / max(wq, 1), out[6] / max(wq, 1)))
print("  wave life mean %.1f Mcyc max %.1f Mcyc, %d waves; cycles per wave_query %.0f" % (life / max(out[28], 1) / 1e6, out[27] / 1e6, out[28], life / max(wq, 1)))
print("rays %d wave_queries %d: candidate pairs per ray %.2f (of %d lane-chunks), wave iterations per query %.2f" % (rays, wq, bits / max(crays, 1), crays, iters / max(wq, 1)))
PY
