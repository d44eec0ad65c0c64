#!/bin/bash
# Diagnostic (GPU box): wave-level step counts of the mesh search on the C3 frame.
# Builds the kernels with -DRTMI_STATS into ray-tracing-cuda_amd/lib/librtmi_stats.so (run the
# build part on the dev box: `tools/mesh_stats.sh build`), then `tools/mesh_stats.sh run [spp]`.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-run}" = build ]; then
  make -C $ROOT/ray-tracing-cuda_amd/csrc OUT=../lib/stats OBJ=../lib/stats/obj EXTRA=-DRTMI_STATS=${RTMI_STATS_LEVEL:-1}
  cp $ROOT/ray-tracing-cuda_amd/lib/stats/librtmi.so $ROOT/ray-tracing-cuda_amd/lib/librtmi_stats.so
  exit 0
fi
RTMI_LIB_PATH=$ROOT/ray-tracing-cuda_amd/lib/librtmi_stats.so python3 - "${2:-512}" <<'PY'
import sys, os, ctypes as C
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", os.getcwd()), "ray-tracing-cuda_amd"))
import torch, rtmi
from rtmi import scenes
spp = int(sys.argv[1])
b = rtmi.SceneBuilder(10086); scenes.bunny(b, 1.0, scenes.procedural_bunny_mesh()); b.commit()
R = rtmi.Renderer(b, 1024, 1024, spp, 10).init_rng()
bpc = int(os.environ.get("RTMI_TOOL_BPC", "0"))
R.render(opts=rtmi.render_opts(blocks_per_cu=bpc) if bpc else None); torch.cuda.synchronize()
out = (C.c_ulonglong * 40)()
rtmi.lib().rtmi_debug_counters(b.h, out, None)
names = ["queue", "rays", "abandoned", "-", "wave_queries", "searches", "node_steps", "face_steps", "nodes_popped",
         "blocks_popped", "insert_rounds", "hist<=1", "hist<=4", "hist<=8", "hist<=12", "hist<=20", "hist>20",
         "cyc_gen", "cyc_list", "cyc_search", "cyc_replay", "cyc_shade", "cyc_node_fetch", "cyc_node_test", "cyc_node_push",
         "cyc_face", "cyc_life_sum", "cyc_life_max", "waves", "calib"]
d = dict(zip(names, list(out)))
print(d)
wq = d["wave_queries"]
life = d["cyc_life_sum"]
print("stamp cost (two back to back): %.0f cycles" % (d["calib"] / d["searches"]))
print("waves %d, mean life %.1f Mcyc, max life %.1f Mcyc" % (d["waves"], life / d["waves"] / 1e6, d["cyc_life_max"] / 1e6))
for k in ("gen", "list", "search", "replay", "shade", "node_fetch", "node_test", "node_push", "face"):
    print("  %-10s %5.1f%% of wave time, %7.0f cycles per wave_query" % (k, 100.0 * d["cyc_" + k] / life, d["cyc_" + k] / wq))
print("  per node step: fetch %.0f test %.0f push %.0f cycles; per face step %.0f" % (
    d["cyc_node_fetch"] / d["node_steps"], d["cyc_node_test"] / d["node_steps"], d["cyc_node_push"] / d["node_steps"],
    d["cyc_face"] / max(1, d["face_steps"])))
import numpy as np
ws = np.zeros((16384, 16), dtype=np.uint64)
L = rtmi.lib()
if hasattr(L, "rtmi_debug_wave_stats"):
    L.rtmi_debug_wave_stats(ws.ctypes.data_as(C.c_void_p), C.c_size_t(ws.nbytes))
    ws = ws[ws[:, 0] > 0]
    setup = (ws[:, 15] >> np.uint64(32)).astype(np.float64); ctrl = (ws[:, 15] & np.uint64(0xffffffff)).astype(np.float64) * 256
    ws_raw = ws.copy()
    ws = ws.astype(np.float64)
    order = np.argsort(-ws[:, 0])
    if os.environ.get("RTMI_STATS_MARKS"):
        print("time (Mcyc) at 600, 1200, ... 5400 queries of the longest-lived waves, and cycles per query in between:")
        for w in order[:6].tolist():
            r = ws[w]
            raw = [int(ws_raw[w][1 + i]) for i in range(9) if ws_raw[w][1 + i] > 0]
            marks = [float(x & 0xffffffffff) for x in raw]
            steps = [x >> 40 for x in raw]
            print("    search steps per query in those spans: " + " ".join("%5.1f" % ((b2 - a2) / 600.0) for a2, b2 in zip([0] + steps[:-1], steps)))
            rates = [(b2 - a2) / 600.0 for a2, b2 in zip([0.0] + marks[:-1], marks)]
            print("  life %6.1f queries %5d | " % (r[0] / 1e6, r[10]) + " ".join("%5.1f" % (m / 1e6) for m in marks) + " | " + " ".join("%5.1fk" % (x / 1e3) for x in rates))
    lifes = np.sort(ws[:, 0])
    print("wave life percentiles (Mcyc): " + "  ".join("p%d %.0f" % (q, np.percentile(lifes, q) / 1e6) for q in (1, 10, 25, 50, 75, 90, 99, 100)))
    for lo, hi in ((0, 10), (10, 25), (25, 50), (50, 75), (75, 90), (90, 100)):
        a, b2 = np.percentile(lifes, lo), np.percentile(lifes, hi)
        sel = ws[(ws[:, 0] >= a) & (ws[:, 0] <= b2)]
        print("  life p%d..p%d: %d waves, mean queries %.0f, cycles/query %.0f, nodes/query %.0f"
              % (lo, hi, len(sel), sel[:, 10].mean(), (sel[:, 0] / np.maximum(sel[:, 10], 1)).mean(), (sel[:, 13] / np.maximum(sel[:, 10], 1)).mean()))
    cols = ["life", "gen", "list", "search", "replay", "shade", "nfetch", "ntest", "npush", "face", "queries", "nsteps", "fsteps", "npopped", "ins"]
    print("top waves by life (Mcyc; then per query cycles):")
    for w in order[:6].tolist() + order[len(order) // 2: len(order) // 2 + 2].tolist():
        r = ws[w]
        q = max(r[10], 1)
        print("  life %6.1f  queries %6d  per query: total %6.0f gen %5.0f list %5.0f search %6.0f replay %5.0f shade %5.0f | nsteps/q %5.2f fsteps/q %4.2f nodes/step %4.1f ins/q %4.2f | per nstep fetch %4.0f test %4.0f push %4.0f  per fstep %5.0f | setup/q %5.0f ctrl/q %5.0f"
              % (r[0] / 1e6, r[10], r[0] / q, r[1] / q, r[2] / q, r[3] / q, r[4] / q, r[5] / q, r[11] / q, r[12] / q,
                 r[13] / max(r[11], 1), r[14] / q, r[6] / max(r[11], 1), r[7] / max(r[11], 1), r[8] / max(r[11], 1), r[9] / max(r[12], 1), setup[w] / q, ctrl[w] / q))
print("rays/wave_query %.1f  node_steps/search %.2f  face_steps/search %.2f  nodes/node_step %.1f  blocks/face_step %.1f  insert_rounds/search %.3f"
      % (d["rays"] / wq, d["node_steps"] / d["searches"], d["face_steps"] / d["searches"],
         d["nodes_popped"] / max(1, d["node_steps"]), d["blocks_popped"] / max(1, d["face_steps"]), d["insert_rounds"] / d["searches"]))
PY
