"""Host-side pieces of the scene-facing API layer (no GPU): the JPEG writer behind WriteImage,
the stbi_load stand-in (baseline JPEG + PPM) and the OBJ loader's vertex/face handling."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
API = os.path.join(ROOT, "ray-tracing-cuda_amd", "api")

PROG = r'''
#define STB_IMAGE_IMPLEMENTATION
#include <stb_image.h>
#include "rt_jpeg.h"
#include <cstdio>
#include <cstring>
#include <vector>
int main(int argc, char **argv) {
  if (!std::strcmp(argv[1], "decode")) {  // decode <in> <out.raw>
    int w, h, c;
    unsigned char *p = stbi_load(argv[2], &w, &h, &c, 4);
    if (!p) return 2;
    FILE *f = std::fopen(argv[3], "wb");
    std::fwrite(&w, 4, 1, f); std::fwrite(&h, 4, 1, f);
    std::fwrite(p, 1, (size_t)w * h * 4, f);
    std::fclose(f);
    stbi_image_free(p);
    return 0;
  }
  // encode <w> <h> <in.raw> <out.jpg>
  int w = std::atoi(argv[2]), h = std::atoi(argv[3]);
  std::vector<unsigned char> rgb((size_t)w * h * 3);
  FILE *f = std::fopen(argv[4], "rb");
  if (std::fread(rgb.data(), 1, rgb.size(), f) != rgb.size()) return 3;
  std::fclose(f);
  return rt_write_jpeg(argv[5], w, h, rgb.data()) ? 0 : 4;
}
'''


@pytest.fixture(scope="module")
def tool(tmp_path_factory):
    d = tmp_path_factory.mktemp("apitool")
    src = d / "tool.cc"
    src.write_text(PROG)
    exe = d / "tool"
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", os.path.join(API, "compat"), "-I", os.path.join(API, "src"),
                    str(src), "-o", str(exe)], check=True)
    return str(exe)


def _test_image(h, w):
    y, x = np.mgrid[0:h, 0:w]
    img = np.stack([(x * 255 // max(w - 1, 1)), (y * 255 // max(h - 1, 1)), ((x + y) * 7) % 256], -1).astype(np.uint8)
    img[h // 4:h // 2, w // 4:w // 2] = [250, 10, 30]
    return np.ascontiguousarray(img)


@pytest.mark.parametrize("h,w", [(16, 16), (37, 53), (64, 96)])
def test_jpeg_writer_is_decodable_and_close(tool, tmp_path, h, w):
    from PIL import Image
    img = _test_image(h, w)
    (tmp_path / "in.raw").write_bytes(img.tobytes())
    out = tmp_path / "out.jpg"
    subprocess.run([tool, "encode", str(w), str(h), str(tmp_path / "in.raw"), str(out)], check=True)
    dec = np.asarray(Image.open(out).convert("RGB"))
    assert dec.shape == img.shape
    assert np.abs(dec.astype(int) - img.astype(int)).max() <= 3  # quantiser step 1: near-lossless


@pytest.mark.parametrize("sub", [0, 2])  # 4:4:4 and 4:2:0
def test_stbi_load_reads_baseline_jpeg(tool, tmp_path, sub):
    from PIL import Image
    img = _test_image(40, 72)
    if sub:  # chroma-subsampled: use a smooth picture so the up-sampling filter does not dominate
        y, x = np.mgrid[0:40, 0:72]
        img = np.stack([x * 3, y * 6, 255 - x * 2 - y], -1).astype(np.uint8)
    p = tmp_path / "in.jpg"
    Image.fromarray(img).save(p, quality=95, progressive=False, subsampling=sub)
    subprocess.run([tool, "decode", str(p), str(tmp_path / "o.raw")], check=True)
    raw = (tmp_path / "o.raw").read_bytes()
    w, h = np.frombuffer(raw[:8], dtype=np.int32)
    got = np.frombuffer(raw[8:], dtype=np.uint8).reshape(h, w, 4)
    ref = np.asarray(Image.open(p).convert("RGB"))
    assert (w, h) == (72, 40) and (got[..., 3] == 255).all()
    err = np.abs(got[..., :3].astype(int) - ref.astype(int))
    if sub == 0:
        assert err.max() <= 3
    # 4:2:0: chroma is upsampled nearest-neighbour here, libjpeg smooths it, so sharp colour
    # edges differ locally; the picture as a whole must still agree
    assert err.mean() < 2.0 and np.percentile(err, 90) <= 4


def test_stbi_load_reads_ppm_and_own_jpeg(tool, tmp_path):
    img = _test_image(24, 40)
    ppm = tmp_path / "a.ppm"
    ppm.write_bytes(b"P6\n40 24\n255\n" + img.tobytes())
    subprocess.run([tool, "decode", str(ppm), str(tmp_path / "o.raw")], check=True)
    raw = (tmp_path / "o.raw").read_bytes()
    got = np.frombuffer(raw[8:], dtype=np.uint8).reshape(24, 40, 4)
    assert np.array_equal(got[..., :3], img)
    (tmp_path / "in.raw").write_bytes(img.tobytes())
    subprocess.run([tool, "encode", "40", "24", str(tmp_path / "in.raw"), str(tmp_path / "own.jpg")], check=True)
    subprocess.run([tool, "decode", str(tmp_path / "own.jpg"), str(tmp_path / "o2.raw")], check=True)
    got2 = np.frombuffer((tmp_path / "o2.raw").read_bytes()[8:], dtype=np.uint8).reshape(24, 40, 4)
    assert np.abs(got2[..., :3].astype(int) - img.astype(int)).max() <= 3


def test_make_assets_obj_round_trips_every_float(tmp_path):
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import make_assets
    scenes = make_assets.load_scenes()
    faces = scenes.procedural_bunny_mesh(5)
    p = tmp_path / "m.obj"
    make_assets.write_obj(str(p), faces)
    verts, tris = [], []
    for line in p.read_text().splitlines():
        if line.startswith("v "):
            verts.append([np.float32(t) for t in line.split()[1:4]])
        elif line.startswith("f "):
            tris.append([int(t) - 1 for t in line.split()[1:4]])
    back = np.array([[verts[i] for i in t] for t in tris], dtype=np.float32)
    assert np.array_equal(back, faces)
