"""Host-side logic of the product (no GPU): C-ABI surface, frame/tile arithmetic, scene
recorder bookkeeping, error behaviour, and the float-threshold facts the kernel relies on."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import common
import oraclelib
import rtmi
from rtmi import scenes
from rtmi.scenes import v3

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_declared_symbol_is_exported_and_bound():
    hdr = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(rtmi_[a-z0-9_]+)\s*\(", hdr)) - {"rtmi_transform_fn"}
    assert len(declared) >= 40
    L = rtmi.lib()
    bound = {n for n, _, _ in rtmi.SYMBOLS}
    for name in sorted(declared):
        assert hasattr(L, name), "librtmi.so does not export %s" % name
        assert name in bound, "%s is not bound in rtmi.SYMBOLS" % name
    assert L.rtmi_version() == 3


def test_float_thresholds_used_by_the_kernel():
    """The trace kernel (trace_helpers.h) replaces two double compares of the reference by float compares:
    '1e-3 <= t' (ray_tracing.cu:22 -> utils.cu:74) and 'fabs(det) < 1e-7' (utils.cu:60).
    Valid because the binary32 nearest to each constant lies ABOVE it and t/det are binary32."""
    f = np.float32
    assert float(f(1e-3)) > 1e-3 and float(np.nextafter(f(1e-3), f(0))) < 1e-3
    assert float(f(1e-7)) > 1e-7 and float(np.nextafter(f(1e-7), f(0))) < 1e-7
    assert float(f(1e9)) == 1e9  # Sky's t is exact in binary32 (sky.cu:20)


def test_rejection_threshold():
    """The trace kernel (trace_helpers.h) decides the rejection loop `sqrtf(s) > 1` (lambertian.cu:25-26) as
    `s > 1 + 2^-23`: check the equivalence on every binary32 in a window around 1 and on a
    random sample of the whole range [0, 3]."""
    f = np.float32
    thr = np.nextafter(f(1), f(2))
    xs = np.arange(0x3f7f0000, 0x3f810000, dtype=np.uint32).view(np.float32)
    assert ((np.sqrt(xs) > 1) == (xs > thr)).all()
    r = np.random.default_rng(0).uniform(0, 3, 1_000_000).astype(np.float32)
    assert ((np.sqrt(r) > 1) == (r > thr)).all()
    assert float(thr) == 1.00000011920928955078125


def test_jitter_in_binary32_for_power_of_two_frames():
    """render_body.h forms the pixel jitter of ray_tracing.cu:68-73 (binary64: (u + j) / W, 2x - 1, then (x + 1) / 2
    in Camera::RayAt, camera.cu:58-59, and a final float cast) as RN32(u + j) * (1 / W) when W is a power of two: check
    the two against each other on every shape of uniform variate (tiny, mid-range, 1.0) and every column."""
    f32, f64 = np.float32, np.float64
    rng = np.random.default_rng(5)
    draws = np.concatenate([rng.integers(0, 2 ** 32, 2_000_000, dtype=np.uint64), np.arange(0, 4096, dtype=np.uint64),
                            2 ** 32 - 1 - np.arange(0, 4096, dtype=np.uint64),
                            (np.uint64(1) << np.arange(0, 32, dtype=np.uint64))]).astype(np.uint32)
    # curand_uniform (utils.cuh:22-27 over XORWOW): x * 2^-32 + 2^-33 in binary32
    u = (draws.astype(f32) * f32(2.3283064e-10) + f32(1.16415322e-10)).astype(f32)
    assert u.min() > 0 and u.max() == 1.0
    for W in (1, 8, 256, 1024, 4096, 1 << 20):
        j = rng.integers(0, W + 1, u.size).astype(np.int64)  # columns 0..W-1, and H - i reaches H
        x = (u.astype(f64) + j.astype(f64)) / f64(W)
        x = 2 * x - 1
        x = (x + 1) / 2
        ref = x.astype(f32)
        got = ((u + j.astype(f32)).astype(f32) * f32(1.0 / W)).astype(f32)
        assert np.array_equal(ref.view(np.uint32), got.view(np.uint32)), W


@pytest.mark.parametrize("h,w", [(8, 8), (20, 30), (37, 53), (64, 64), (1, 1), (9, 200)])
@pytest.mark.parametrize("world", [1, 2, 3, 8])
def test_tile_shards_partition_the_frame(h, w, world):
    seen = np.zeros(h * w, dtype=np.int32)
    sizes = set()
    for r in range(world):
        f = rtmi.make_frame(h, w, 1, rank=r, world_size=world)
        pm = rtmi.pixel_map(f)
        sizes.add(pm.size)
        assert pm.size % 64 == 0
        ok = pm[pm >= 0]
        seen[ok] += 1
        # a tile's 64 work items are one 8x8 block, row-major
        for t in range(pm.size // 64):
            blk = pm[t * 64:(t + 1) * 64]
            if (blk >= 0).all():
                i, j = blk // w, blk % w
                assert (i - i[0] == np.repeat(np.arange(8), 8)).all() and (j - j[0] == np.tile(np.arange(8), 8)).all()
    assert (seen == 1).all()
    assert len(sizes) == 1  # every rank has the same number of work items (gather stride)


def test_frame_extent_limit():
    """A pixel's row and column share one 32-bit register of the trace kernel, so a frame side ends at RTMI_MAX_EXTENT =
    65535 (the reference has no such limit; include/rtmi.h says so next to rtmi_frame) -- and the refusal says why."""
    L = rtmi.lib()
    hdr = open(os.path.join(ROOT, "include", "rtmi.h")).read()
    assert "#define RTMI_MAX_EXTENT 65535" in hdr
    ok = rtmi.make_frame(65535, 8, 1)
    assert L.rtmi_frame_work_items(C.byref(ok)) == ((65535 + 7) // 8) * 64
    ok = rtmi.make_frame(8, 65535, 1)
    assert L.rtmi_frame_work_items(C.byref(ok)) == ((65535 + 7) // 8) * 64
    for h, w in ((65536, 8), (8, 65536), (1 << 20, 1 << 20)):
        bad = rtmi.make_frame(h, w, 1)
        assert L.rtmi_frame_work_items(C.byref(bad)) < 0
        assert b"65535" in L.rtmi_last_error() and b"RTMI_MAX_EXTENT" in L.rtmi_last_error()
    bad = rtmi.make_frame(0, 8, 1)  # any other bad frame keeps the generic words
    assert L.rtmi_frame_work_items(C.byref(bad)) < 0 and b"bad frame" in L.rtmi_last_error()


def test_frame_validation():
    L = rtmi.lib()
    bad = rtmi.make_frame(0, 8, 1)
    assert L.rtmi_frame_work_items(C.byref(bad)) < 0
    bad = rtmi.make_frame(8, 8, 1, rank=2, world_size=2)
    assert L.rtmi_frame_work_items(C.byref(bad)) < 0
    ok = rtmi.make_frame(16, 16, 1)
    assert L.rtmi_states_bytes(C.byref(ok)) == 4 * 64 * 6 * 4
    assert L.rtmi_tiles_bytes(C.byref(ok)) == 4 * 64 * 3 * 4
    assert L.rtmi_frame_pixel_of(C.byref(ok), 10 ** 9) == -1


def test_get_workload_matches_oracle():
    for spp in [1, 20, 100, 1024]:
        for world in [1, 3, 8]:
            for r in range(world):
                assert rtmi.get_workload(r, world, spp) == oraclelib.lib().orc_get_workload(r, world, spp)


def test_scene_stats_and_bytes_per_ray():
    b = common.build_scene(rtmi.SceneBuilder(1024), "cornell_box")
    st = b.stats()
    assert st["world"] == 9 and st["parallelograms"] == 18 and st["materials"] == 4
    assert b.bytes_per_ray() == 18 * 40 + 32  # 752 B/ray
    b = common.build_scene(rtmi.SceneBuilder(10086), "spheres")
    assert b.bytes_per_ray() == 488 * 28 + 32  # 13.7 KB/ray


def test_camera_matches_oracle_bitwise():
    for name in ["cornell_box", "spheres", "bunny", "mixed"]:
        p = common.build_scene(rtmi.SceneBuilder(5), name, 16 / 9)
        o = common.build_scene(oraclelib.OracleBuilder(5), name, 16 / 9)
        assert np.array_equal(p.camera_get(), o.camera_get())
    p, o = rtmi.SceneBuilder(1), oraclelib.OracleBuilder(1)
    for b in (p, o):
        b.camera_defocus(v3(1, 2, 3), v3(0, 0.5, 0), v3(0, 1, 0), 0.7, 1.5, 0.3, 4.0)
    assert np.array_equal(p.camera_get(), o.camera_get())


def test_scene_time_rng_draws_match_oracle():
    p = common.build_scene(rtmi.SceneBuilder(10086), "spheres")
    o = common.build_scene(oraclelib.OracleBuilder(10086), "spheres")
    assert np.array_equal(p.state0, o.state0) and not np.array_equal(p.state0, p.state0_fresh)


def test_error_behaviour():
    L = rtmi.lib()
    b = rtmi.SceneBuilder(1)
    white = b.lambertian(v3(1, 1, 1))
    with pytest.raises(rtmi.RtmiError):
        b.sphere(v3(0, 0, 0), 1.0, 99)  # unknown material handle
    with pytest.raises(rtmi.RtmiError):
        b.lambertian_tex(7)  # unknown texture handle
    with pytest.raises(rtmi.RtmiError):
        b.stats()  # no camera yet
    assert b"camera" in L.rtmi_last_error()
    b.camera_pinhole(v3(0, 0, 1), v3(0, 0, 0), v3(0, 1, 0), 1.0, 1.0)
    for _ in range(1024):
        b.sphere(v3(0, 0, 0), 1.0, white)
    with pytest.raises(rtmi.RtmiError) as e:  # HitableList::kMaxHitables (hitable_list.cuh:10)
        b.sphere(v3(0, 0, 0), 1.0, white)
    assert "(-4)" in str(e.value)


def test_non_finite_geometry_is_refused():
    """A NaN / infinite coordinate would poison the bounds of the search structures (the 4-wide tree's grid
    quantisation never terminates on an infinite extent): every hitable refuses it with RTMI_ERR_INVALID.  The
    reference would build such a mesh and never hit the face; a caller that wants that drops the face."""
    L = rtmi.lib()
    b = rtmi.SceneBuilder(1)
    m = b.lambertian(v3(1, 1, 1))
    inf, nan = float("inf"), float("nan")
    with pytest.raises(rtmi.RtmiError):
        b.sphere(v3(0, nan, 0), 1.0, m)
    with pytest.raises(rtmi.RtmiError):
        b.sphere(v3(0, 0, 0), inf, m)
    with pytest.raises(rtmi.RtmiError):
        b.triangle([v3(0, 0, 0), v3(1, inf, 0), v3(0, 1, 0)], m)
    with pytest.raises(rtmi.RtmiError):
        b.parallelogram([v3(0, 0, 0), v3(1, 0, 0), v3(0, nan, 0)], m)
    with pytest.raises(rtmi.RtmiError):
        b.parallelepiped([v3(0, 0, 0), v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, -inf)], m)
    with pytest.raises(rtmi.RtmiError):
        b.parallelepiped_lengths(v3(1, 1, 1), m, lambda p: p / 0.0 if p[0] else p * np.float32(inf))
    faces = np.zeros((5, 3, 3), dtype=np.float32)
    faces[3, 1, 2] = inf
    with pytest.raises(rtmi.RtmiError) as e:
        b.bvh(faces, m)
    assert "non-finite" in str(e.value) and b"non-finite" in L.rtmi_last_error()
    b.camera_pinhole(v3(0, 0, 1), v3(0, 0, 0), v3(0, 1, 0), 1.0, 1.0)
    assert b.stats()["world"] == 0  # nothing was appended


def test_exchange_argument_checks():
    """rtmi_gather / rtmi_reduce_sum validate their arguments before they look for RCCL."""
    L = rtmi.lib()
    f = rtmi.make_frame(16, 16, 1, rank=1, world_size=2)
    buf = (C.c_float * (2 * 64 * 3 * 2))()
    assert L.rtmi_gather(None, C.byref(f), buf, buf, 5, None) == -1  # root outside the world
    assert L.rtmi_gather(None, C.byref(f), None, buf, 0, None) == -1
    root = rtmi.make_frame(16, 16, 1, rank=0, world_size=2)
    assert L.rtmi_gather(None, C.byref(root), buf, None, 0, None) == -1 and b"d_all_tiles" in L.rtmi_last_error()
    assert L.rtmi_reduce_sum(None, C.byref(f), None, 0, None) == -1


def test_nested_list_bookkeeping():
    """A HitableList appended to a HitableList (hitable_list.cuh:8): one entry of its parent, up to
    kMaxHitables entries of its own, any nesting depth; brackets must balance."""
    L = rtmi.lib()
    b = rtmi.SceneBuilder(1)
    white = b.lambertian(v3(1, 1, 1))
    b.camera_pinhole(v3(0, 0, 1), v3(0, 0, 0), v3(0, 1, 0), 1.0, 1.0)
    b.sky()
    b.list_begin()
    for _ in range(1024):  # the nested list's own capacity
        b.sphere(v3(0, 0, 0), 1.0, white)
    with pytest.raises(rtmi.RtmiError) as e:
        b.sphere(v3(0, 0, 0), 1.0, white)
    assert "(-4)" in str(e.value)
    with pytest.raises(rtmi.RtmiError):
        b.list_begin()  # a nested list would be entry 1025 of the full list
    b2 = rtmi.SceneBuilder(1)
    w2 = b2.lambertian(v3(1, 1, 1))
    b2.camera_pinhole(v3(0, 0, 1), v3(0, 0, 0), v3(0, 1, 0), 1.0, 1.0)
    b2.list_begin()
    b2.sphere(v3(0, 0, 0), 1.0, w2)
    b2.list_begin()
    b2.parallelogram([v3(0, 0, 0), v3(1, 0, 0), v3(0, 1, 0)], w2)
    b2.parallelepiped([v3(0, 0, 0), v3(1, 0, 0), v3(0, 1, 0), v3(0, 0, 1)], w2)
    b2.list_end()
    with pytest.raises(rtmi.RtmiError):
        b2.commit()  # a list is still open (reported before the missing GPU is)
    assert b"still open" in L.rtmi_last_error()
    b2.list_end()
    with pytest.raises(rtmi.RtmiError):
        b2.list_end()  # unbalanced
    b2.sky()
    st = b2.stats()
    assert st["world"] == 2 and st["spheres"] == 1 and st["parallelograms"] == 7  # nested entries are inlined
    for _ in range(1022):
        b2.sky()
    with pytest.raises(rtmi.RtmiError):
        b2.sky()  # the world list itself is full at 1024 entries, the nested list having counted once


def test_thin_faces_are_counted_by_what_their_nodes_must_cover():
    """rtmi_scene_sliver_faces = mesh faces whose nodes carry a slack exponent (scene.hip: face_slack_exponent): smallest
    angle below asin(1/32) = 1.79 degrees, unless the face is too small for the triangle test's |det| >= 1e-7 ever to
    hold (|e1 x e2| < 1e-7: never accepted, nothing to widen for)."""
    def count(faces):
        b = rtmi.SceneBuilder(1)
        b.camera_pinhole(v3(0, 0, 1), v3(0, 0, 0), v3(0, 1, 0), 1.0, 1.0)
        b.bvh(np.asarray(faces, dtype=np.float32), b.lambertian(v3(1, 1, 1)), k_min=2)
        return b.sliver_faces()

    def tri(angle_deg, size):
        a = np.radians(angle_deg)
        return [[0, 0, 0], [size, 0, 0], [size * np.cos(a), size * np.sin(a), 0]]
    assert count([tri(60, 0.1), tri(2.0, 0.1), tri(1.8, 0.1)]) == 0
    assert count([tri(60, 0.1), tri(1.75, 0.1), tri(0.01, 0.1)]) == 2
    assert count([tri(178.5, 0.1)]) == 1          # obtuse: the two other corners are the thin ones
    assert count([tri(0.5, 1e-4)]) == 0            # 1e-8 x sin: below the test's determinant cut-off whatever the ray
    assert count([tri(0.5, 3e-3)]) == 0 and count([tri(0.5, 5e-3)]) == 1   # |e1 x e2| = 7.9e-8 / 2.2e-7


def test_trace_kernels_declare_no_static_lds():
    """render_body.h reads the id stack at LDS addresses formed from byte offsets of the DYNAMIC LDS array (lds_byte):
    that is only right while the array starts at LDS address 0, i.e. while no trace kernel declares static LDS
    (group_segment_fixed_size == 0 in the code object's metadata).  Checked on the product AND on the diagnostic
    margin-check build (a __shared__ added under one of its macros would make the fold read wrong ids without a fault);
    the launcher asks the same of whatever build is loaded (kernels.hip: hipFuncGetAttributes, once per variant)."""
    import subprocess
    import tempfile
    llvm = "/opt/rocm/lib/llvm/bin"
    if not os.path.exists(llvm + "/llvm-readelf"):
        pytest.skip("no llvm-readelf")
    libs = [rtmi.LIB_PATH, os.path.join(os.path.dirname(rtmi.LIB_PATH), "librtmi_check1.so")]
    assert os.path.exists(libs[1]), "librtmi_check1.so missing: __graft_entry__.build() builds it"
    for lib in libs:
        with tempfile.TemporaryDirectory() as tmp:
            fat, co = os.path.join(tmp, "fatbin"), os.path.join(tmp, "co.o")
            subprocess.check_call([llvm + "/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
            subprocess.check_call([llvm + "/clang-offload-bundler", "--unbundle", "--type=o", "--input=" + fat,
                                   "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], stderr=subprocess.DEVNULL)
            notes = subprocess.check_output([llvm + "/llvm-readelf", "--notes", co], text=True)
        seen = 0
        for blk in notes.split("- .agpr_count")[1:]:
            name = re.search(r"\.name:\s+(\S+)", blk).group(1)
            if "render_kernel" in name or "probe_kernel" in name:
                seen += 1
                assert re.search(r"\.group_segment_fixed_size:\s+0\b", blk), (lib, name)
        assert seen >= 18, lib


def test_render_opts_validation():
    """rtmi_render_ex checks its per-call options before anything else touches them."""
    import ctypes as C
    L = rtmi.lib()
    b = rtmi.SceneBuilder(1)
    fr = rtmi.make_frame(8, 8, 1)
    dummy = C.c_void_p(16)  # never dereferenced: argument checks come first
    bad_size = rtmi.render_opts()
    bad_size.size = 12
    assert L.rtmi_render_ex(b.h, C.byref(fr), C.byref(bad_size), dummy, dummy, None, None) == -1
    assert b"size" in L.rtmi_last_error()
    for kw in (dict(schedule=3), dict(threads_per_block=100), dict(threads_per_block=1024), dict(sparse_stride=12),
               dict(exclusive=2), dict(blocks_per_cu=-1), dict(probe_spp=65), dict(probe_spp=-1), dict(plan=3),
               dict(wave_priority=12), dict(wave_priority=8192), dict(lane_stride=3), dict(lane_stride=128), dict(cost_probe=2), dict(first_pass=5000)):
        o = rtmi.render_opts(**kw)
        assert L.rtmi_render_ex(b.h, C.byref(fr), C.byref(o), dummy, dummy, None, None) == -1, kw
        assert b"out of range" in L.rtmi_last_error()
    for pct in ((101, 0, 0), (0, -1, 0)):
        o = rtmi.render_opts(head_pct=pct)
        assert L.rtmi_render_ex(b.h, C.byref(fr), C.byref(o), dummy, dummy, None, None) == -1 and b"head_pct" in L.rtmi_last_error()
    o = rtmi.render_opts(head_pct=(40, 50, 30))  # the classes' thresholds must not increase
    assert L.rtmi_render_ex(b.h, C.byref(fr), C.byref(o), dummy, dummy, None, None) == -1 and b"must not increase" in L.rtmi_last_error()
    ok = rtmi.render_opts(schedule=2, sparse_stride=8, exclusive=0, probe_spp=8, head_pct=(90, 60, 20))
    assert L.rtmi_render_ex(b.h, C.byref(fr), C.byref(ok), dummy, dummy, None, None) == -1
    assert b"not committed" in L.rtmi_last_error()  # the options passed; the scene is what is missing
    # the call's counters + states copy + probe counts + tile costs and order + 32 words + the head list (16,384 entries)
    # + the probe's work counts + per quarter tile: cost, sorted list, order (4 each per tile) + 4 words + the chain
    # plan (per tile: what follows, its estimate, who renders it; per chain its first tile), rounded up to 256;
    # then the wave-priority table (2^14 SIMD rows of 16 words) and two kernel-argument blocks
    body = (40 * 8 + 64 * 6 * 4 + 64 * 4 + 1 * 4 * 2 + 128 + 16384 * 4 + 64 * 4 + 1 * 4 * 15 + 16 + 32768 * 4 + 255) & ~255
    total = L.rtmi_render_scratch_bytes(C.byref(fr))
    params = (total - body - (1 << 14) * 16 * 4) // 2
    assert total == body + (1 << 14) * 16 * 4 + 2 * params and params % 256 == 0 and 512 <= params <= 2048, (total, body, params)


def test_no_cpu_fallback():
    """Without a GPU every compute entry point must fail loudly, never render on the CPU."""
    L = rtmi.lib()
    if L.rtmi_device_count() > 0:
        pytest.skip("a GPU is present")
    b = common.build_scene(rtmi.SceneBuilder(1024), "cornell_box")
    with pytest.raises(rtmi.RtmiError) as e:
        b.commit()
    assert "(-2)" in str(e.value) and "no CPU fallback" in str(e.value)
    f = rtmi.make_frame(8, 8, 1)
    buf = (C.c_uint32 * (64 * 6))()
    assert L.rtmi_rng_init(C.c_uint64(1), C.byref(f), buf, None) == -2
    with pytest.raises(rtmi.RtmiError):
        rtmi.Renderer(b, 8, 8, 1)


def test_product_does_not_import_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may touch oracle/."""
    pkg = os.path.join(ROOT, "ray-tracing-cuda_amd")
    for dirpath, _, files in os.walk(pkg):
        for fn in files:
            if fn.endswith((".py", ".hip", ".h", ".cuh", ".cpp", ".hpp")):
                txt = open(os.path.join(dirpath, fn), errors="ignore").read()
                assert "oraclelib" not in txt and "liboracle" not in txt and "oracle/" not in txt, fn
