#!/usr/bin/env python3
"""Writes stand-ins for the two assets the reference does not ship (its .gitignore excludes
resources/): resources/bunny.obj (scenes/bunny.cu:102) and resources/earthmap.jpg
(scenes/birthday.cu:80).  Both are procedural and deterministic (rtmi/scenes.py):
a 69,312-triangle closed blob filling the Stanford bunny's bounding box, and an
equirectangular colour map saved as a baseline JPEG.

usage: tools/make_assets.py [out_dir]   (default: current directory)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "ray-tracing-cuda_amd"))


def load_scenes():
    # rtmi/scenes.py has no dependency on the native library; import it without rtmi/__init__
    import importlib.util
    spec = importlib.util.spec_from_file_location("rt_scenes", os.path.join(ROOT, "ray-tracing-cuda_amd", "rtmi", "scenes.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def write_obj(path, faces):
    """faces (n,3,3) float32 -> OBJ with shared vertices; %.9g keeps every binary32 exactly."""
    verts, index = {}, []
    for tri in faces:
        ids = []
        for p in tri:
            key = (float(p[0]), float(p[1]), float(p[2]))
            if key not in verts:
                verts[key] = len(verts) + 1
            ids.append(verts[key])
        index.append(ids)
    with open(path, "w") as f:
        f.write("# procedural stand-in for the Stanford bunny (tools/make_assets.py)\n")
        for (x, y, z) in verts:
            f.write("v %.9g %.9g %.9g\n" % (x, y, z))
        for a, b, c in index:
            f.write("f %d %d %d\n" % (a, b, c))


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else "."
    res = os.path.join(out, "resources")
    os.makedirs(res, exist_ok=True)
    scenes = load_scenes()
    write_obj(os.path.join(res, "bunny.obj"), scenes.procedural_bunny_mesh())
    from PIL import Image
    img = scenes.procedural_earthmap(512, 1024)[..., :3]
    Image.fromarray(img).save(os.path.join(res, "earthmap.jpg"), quality=92, progressive=False, optimize=False, subsampling=0)
    print("wrote", os.path.join(res, "bunny.obj"), "and", os.path.join(res, "earthmap.jpg"))


if __name__ == "__main__":
    main()
