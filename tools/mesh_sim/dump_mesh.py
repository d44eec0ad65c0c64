"""Write the C3 stand-in mesh as raw float32 (n*9) for tools/mesh_sim/mesh_sim.cc."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd", "rtmi"))
import scenes  # noqa: E402  (pure numpy; does not load librtmi)

n = int(sys.argv[2]) if len(sys.argv) > 2 else 76
scenes.procedural_bunny_mesh(n).astype("float32").tofile(sys.argv[1])
