#!/bin/bash
# Diagnostic (GPU box): wave-level step counts of the mesh search on the C3 frame.
# Builds the kernels with -DRTMI_STATS into ray-tracing-cuda_amd/lib/librtmi_stats.so (run the
# build part on the dev box: `tools/mesh_stats.sh build`), then `tools/mesh_stats.sh run [spp]`.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
if [ "${1:-run}" = build ]; then
  make -C $ROOT/ray-tracing-cuda_amd/csrc OUT=../lib/stats OBJ=../lib/stats/obj EXTRA=-DRTMI_STATS
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
R.render(); torch.cuda.synchronize()
out = (C.c_ulonglong * 32)()
rtmi.lib().rtmi_debug_counters(b.h, out, None)
names = ["queue", "rays", "abandoned", "-", "wave_queries", "searches", "node_steps", "face_steps", "nodes_popped",
         "blocks_popped", "insert_rounds", "hist<=1", "hist<=4", "hist<=8", "hist<=12", "hist<=20", "hist>20",
         "cyc_gen", "cyc_list", "cyc_search", "cyc_replay", "cyc_shade", "cyc_node_fetch", "cyc_node_test", "cyc_node_push",
         "cyc_face", "cyc_life_sum", "cyc_life_max", "waves"]
d = dict(zip(names, list(out)))
print(d)
wq = d["wave_queries"]
life = d["cyc_life_sum"]
print("waves %d, mean life %.1f Mcyc, max life %.1f Mcyc" % (d["waves"], life / d["waves"] / 1e6, d["cyc_life_max"] / 1e6))
for k in ("gen", "list", "search", "replay", "shade", "node_fetch", "node_test", "node_push", "face"):
    print("  %-10s %5.1f%% of wave time, %7.0f cycles per wave_query" % (k, 100.0 * d["cyc_" + k] / life, d["cyc_" + k] / wq))
print("  per node step: fetch %.0f test %.0f push %.0f cycles; per face step %.0f" % (
    d["cyc_node_fetch"] / d["node_steps"], d["cyc_node_test"] / d["node_steps"], d["cyc_node_push"] / d["node_steps"],
    d["cyc_face"] / max(1, d["face_steps"])))
print("rays/wave_query %.1f  node_steps/search %.2f  face_steps/search %.2f  nodes/node_step %.1f  blocks/face_step %.1f  insert_rounds/search %.3f"
      % (d["rays"] / wq, d["node_steps"] / d["searches"], d["face_steps"] / d["searches"],
         d["nodes_popped"] / max(1, d["node_steps"]), d["blocks_popped"] / max(1, d["face_steps"]), d["insert_rounds"] / d["searches"]))
PY
