#!/usr/bin/env python3
"""How far can FMA contraction move a frame?  (CPU only.)

The reference is built without -fmad=false (CMakeLists.txt:7), so nvcc fuses multiply-add pairs where it sees fit;
the oracle and the kernels are built with contraction OFF, and no reference output exists to say which is right.
This renders the same frames with the oracle as built (-ffp-contract=off) and with the same source built
`g++ -mfma -ffp-contract=fast` (oracle/Makefile: liboracle_fma.so) and reports the relative L2 distance of the
images and the fraction of pixels whose ray COUNT differs (a pixel whose stream desynchronised: some rejection loop
or hit decision flipped).  gcc's choice of which pairs to fuse is not nvcc's; this bounds the magnitude, not the bits.

usage: tools/contraction_study.py [--quick]   -> JSON on stdout (committed as profiles/r03_contraction_study.json)
"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CASES = [("c1_spheres_256x256x16_d8", "spheres", 256, 16, 8), ("cornell_256x256x1024_d50", "cornell_box", 256, 1024, 50)]
if "--quick" in sys.argv:
    CASES = [("c1_spheres_256x256x16_d8", "spheres", 256, 16, 8), ("cornell_256x256x64_d50", "cornell_box", 256, 64, 50)]

WORKER = r'''
import sys, os, numpy as np
sys.path.insert(0, os.path.join(%(root)r, "ray-tracing-cuda_amd")); sys.path.insert(0, os.path.join(%(root)r, "tests"))
import common
rgb, rays, states, total, _ = common.oracle_render(%(scene)r, %(side)d, %(side)d, %(spp)d, %(depth)d)
np.savez(%(out)r, rgb=rgb, rays=rays, total=total)
'''


def render(variant, scene, side, spp, depth, out):
    env = dict(os.environ)
    if variant == "fma":
        env["ORACLE_VARIANT"] = "fma"
    else:
        env.pop("ORACLE_VARIANT", None)
    subprocess.run([sys.executable, "-c", WORKER % dict(root=ROOT, scene=scene, side=side, spp=spp, depth=depth, out=out)],
                   check=True, env=env)


def main():
    import numpy as np
    res = {}
    with tempfile.TemporaryDirectory() as tmp:
        for tag, scene, side, spp, depth in CASES:
            paths = {}
            for variant in ("off", "fma"):
                paths[variant] = os.path.join(tmp, "%s_%s.npz" % (tag, variant))
                render(variant, scene, side, spp, depth, paths[variant])
            a, b = np.load(paths["off"]), np.load(paths["fma"])
            d = a["rgb"].astype(np.float64) - b["rgb"].astype(np.float64)
            res[tag] = {
                "rel_l2": float(np.sqrt((d ** 2).sum() / (a["rgb"].astype(np.float64) ** 2).sum())),
                "max_abs": float(np.abs(d).max()),
                "pixels_with_a_different_ray_count": float((a["rays"] != b["rays"]).mean()),
                "pixels_with_a_different_colour": float((a["rgb"] != b["rgb"]).any(axis=2).mean()),
                "ray_totals": [int(a["total"]), int(b["total"])],
            }
    print(json.dumps({"what": "oracle -ffp-contract=off vs the same source -mfma -ffp-contract=fast (g++)", "cases": res}, indent=1))


if __name__ == "__main__":
    main()
