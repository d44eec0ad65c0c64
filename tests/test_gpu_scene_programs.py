"""The reference's scene programs, compiled UNCHANGED from /root/reference/scenes/*.cu against
ray-tracing-cuda_amd/api/ (build: __graft_entry__.build() -> make -C ray-tracing-cuda_amd/api),
run on the GPU through Main / DistributedMain, and compared with the ORACLE's render of the same
worlds (and, as a second check, with the Python binding).  The binaries are built in the development container (the
reference checkout does not exist on the GPU box) and travel with the repository snapshot.

Scene constants are computed on the device by the scene's own InitWorld kernel here
(glm::rotateY -> cosf/sinf, tan in the Camera constructor) and on the host in
rtmi/scenes.py, so the two paths may differ in the last bit of a few constants; frames are
therefore compared at the north-star tolerance, and exactly where no libm call is involved."""
import os
import subprocess
import sys

import numpy as np
import pytest

import common

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "ray-tracing-cuda_amd", "build", "scenes")


def run_scene(name, tmp_path, h, w, spp, depth=10, extra_env=None):
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.skip("%s not built (needs the reference checkout at build time)" % exe)
    env = dict(os.environ, RT_HEIGHT=str(h), RT_WIDTH=str(w), RT_SPP=str(spp), RT_MAX_DEPTH=str(depth),
               RT_DUMP=str(tmp_path / "frame.bin"))
    env.update(extra_env or {})
    r = subprocess.run([exe], cwd=str(tmp_path), env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Ray tracing finished in" in r.stderr
    img = np.fromfile(str(tmp_path / "frame.bin"), dtype=np.float32).reshape(h, w, 3)
    return img, r.stderr


def assets(tmp_path):
    subprocess.run([sys.executable, os.path.join(ROOT, "tools", "make_assets.py"), str(tmp_path)], check=True,
                   capture_output=True)


def test_cornell_box_program(tmp_path):
    """scenes/cornell_box.cu (cornell_box.cu:71-78) through Main, seed 1024, against the ORACLE's render of the same
    world.  The scene's constants come from the device's own libm here (glm::rotateY -> cosf / sinf, tan in the Camera
    constructor) and from the host's in the oracle, so a few constants may differ in their last bit: 1e-3 relative L2,
    and bit-equal on nearly all pixels."""
    h, w, spp = 64, 96, 8
    img, log = run_scene("cornell_box", tmp_path, h, w, spp)
    ref, _, _, _, _ = common.oracle_render("cornell_box", h, w, spp, 10)
    rel = common.rel_l2(img, ref)
    assert rel <= 1e-3, rel
    assert (img == ref).all(axis=2).mean() > 0.5, (img == ref).all(axis=2).mean()
    gpu, _, _, _, _ = common.gpu_render("cornell_box", h, w, spp, 10)
    assert np.array_equal(gpu, ref)  # the Python binding, same constants as the oracle: bit for bit
    from PIL import Image
    jpg = np.asarray(Image.open(str(tmp_path / "image.jpeg")).convert("RGB"))
    assert jpg.shape == (h, w, 3)
    want = (np.clip(img, 0, 1) * 255).astype(np.uint8)  # WriteImage truncates (utils.cu:92)
    assert np.abs(jpg.astype(int) - want.astype(int)).max() <= 4


def test_spheres_program_against_the_oracle(tmp_path):
    """scenes/spheres.cu (spheres.cu:100-110) through DistributedMain, seed 10086: the layout is drawn on the device
    from d_states[0] (spheres.cu:56-71,105; quirk g5), so pixel 0 starts further along its stream -- the oracle replays
    the same draws from its own copy of that state (rtmi/scenes.py: spheres, on an OracleBuilder)."""
    h, w, spp = 48, 64, 4
    img, _ = run_scene("spheres", tmp_path, h, w, spp)
    ref, _, _, _, _ = common.oracle_render("spheres", h, w, spp, 10)
    assert common.rel_l2(img, ref) <= 1e-3
    assert (img == ref).all(axis=2).mean() > 0.98
    assert np.array_equal(img[0, 0], ref[0, 0])  # pixel 0, whose stream the layout's draws advanced
    gpu, _, _, _, _ = common.gpu_render("spheres", h, w, spp, 10)
    assert np.array_equal(gpu, ref)


@pytest.fixture(scope="module")
def file_rccl(tmp_path_factory):
    """tests/stubs/file_rccl.c: a functional stand-in for RCCL that carries messages between processes through files."""
    out = str(tmp_path_factory.mktemp("file_rccl") / "libfile_rccl.so")
    subprocess.run(["gcc", "-shared", "-fPIC", "-O1", "-I/opt/rocm/include", "-o", out,
                    os.path.join(ROOT, "tests", "stubs", "file_rccl.c"), "-L/opt/rocm/lib", "-lamdhip64"], check=True)
    return out


def run_ranks(name, tmp_path, stub, n, h, w, spp, depth=10, extra_env=None, per_rank_setup=None):
    """`n` processes of scene program `name` on this one GPU, their RCCL calls answered by the file-backed stub."""
    exe = os.path.join(BIN, name)
    if not os.path.exists(exe):
        pytest.skip("%s not built" % exe)
    base = tmp_path / ("%s_x%d_%s" % (name, n, "_".join(sorted((extra_env or {}).values())) or "tiles"))
    (base / "msgs").mkdir(parents=True)
    (base / "rdv").mkdir()
    procs = []
    for r in range(n):
        d = base / ("r%d" % r)
        d.mkdir()
        if per_rank_setup:
            per_rank_setup(d)
        env = dict(os.environ, RT_HEIGHT=str(h), RT_WIDTH=str(w), RT_SPP=str(spp), RT_MAX_DEPTH=str(depth),
                   RT_DUMP=str(d / "frame.bin"), RANK=str(r), WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1", MASTER_PORT="29519",
                   RT_RUN_ID="pytest-file-%d-%s-%d" % (os.getpid(), name, n), RT_RENDEZVOUS_DIR=str(base / "rdv"),
                   LD_PRELOAD=stub, STUB_RCCL_DIR=str(base / "msgs"))
        env.update(extra_env or {})
        procs.append(subprocess.Popen([exe], cwd=str(d), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=600)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(l[-1500:] for l in logs)
    assert not [f for f in os.listdir(str(base / "msgs")) if f.startswith("msg_")], "messages nobody received"
    assert not os.path.exists(str(base / "r1" / "frame.bin"))  # only the root assembles and writes a frame
    return np.fromfile(str(base / "r0" / "frame.bin"), dtype=np.float32).reshape(h, w, 3), logs


def test_ranks_on_one_gpu_through_a_file_backed_rccl(tmp_path, file_rccl):
    """DistributedMain with 2 and 3 ranks END TO END on the one GPU of the box: pixel tiles sharded by rank, every rank
    renders its shard, `rtmi_gather` (grouped send / recv on the program's communicator, librtmi.so taking RCCL from
    the process) brings the tile buffers to rank 0, `rtmi_untile` assembles the frame -- with the RCCL entry points
    answered by tests/stubs/file_rccl.c, because RCCL itself refuses two ranks on one device.  The frame must be the
    single-process frame bit for bit (utils.cu:181-242 replaced; quirk g5: pixel 0's stream is advanced by the layout's
    draws on its owner only)."""
    h, w, spp = 40, 56, 4
    one, _ = run_scene("spheres", tmp_path, h, w, spp)
    for n in (2, 3):
        many, logs = run_ranks("spheres", tmp_path, file_rccl, n, h, w, spp)
        assert np.array_equal(many, one), n
        assert all("pixel-tile shard" in log for log in logs)
    # a mesh scene: every rank loads the model and builds the same trees
    h, w, spp = 32, 32, 2
    one, _ = (assets(tmp_path), run_scene("bunny", tmp_path, h, w, spp))[1]
    many, _ = run_ranks("bunny", tmp_path, file_rccl, 2, h, w, spp, per_rank_setup=assets)
    assert np.array_equal(many, one)


def test_reference_sample_split_on_two_ranks(tmp_path, file_rccl):
    """RT_DIST_MODE=spp: the reference's own decomposition (utils.cu:111-130,189,220,238) -- every rank renders the WHOLE
    frame with GetWorkload(rank, world, spp) samples from the SAME seed (quirk g11: rank 1's samples are a prefix of rank
    0's), no post-process; `rtmi_reduce_sum` adds the frames on rank 0, which divides by the total and post-processes.
    Expected: the oracle's two raw frames added and post-processed the same way."""
    import rtmi
    h, w, spp = 32, 40, 5
    got, logs = run_ranks("spheres", tmp_path, file_rccl, 2, h, w, spp, extra_env={"RT_DIST_MODE": "spp"})
    assert all("sample split" in log for log in logs)
    parts = [common.oracle_render("spheres", h, w, rtmi.get_workload(r, 2, spp), 10, post=False)[0] for r in range(2)]
    assert [rtmi.get_workload(r, 2, spp) for r in range(2)] == [3, 2]
    want = np.sqrt(np.clip((parts[0] + parts[1]) / np.float32(spp), np.float32(0), np.float32(1))).astype(np.float32)
    assert common.rel_l2(got, want) <= 1e-3
    assert (got == want).all(axis=2).mean() > 0.98


def write_ppm(path, rgb):
    """Binary PPM (P6): the stb_image stand-in (api/compat/stb_image.h) decodes it losslessly, whatever the file is
    called, so the texels the scene program samples are exactly the array the oracle is given."""
    rgb = np.ascontiguousarray(rgb[..., :3], dtype=np.uint8)
    with open(path, "wb") as f:
        f.write(b"P6\n%d %d\n255\n" % (rgb.shape[1], rgb.shape[0]))
        f.write(rgb.tobytes())


def test_bunny_program_with_obj_stand_in(tmp_path):
    """Model<false> (OBJ through api/model.h) -> BVH<Face<false>, AABB> -> DistributedMain, against the ORACLE's render
    of the same 69,312 faces (reference leaf size 2048): the loader leg and the renderer in one comparison."""
    from rtmi import scenes
    assets(tmp_path)
    h, w, spp = 64, 64, 2
    img, log = run_scene("bunny", tmp_path, h, w, spp)
    assert "69312 faces in mesh 0" in log
    mesh = scenes.procedural_bunny_mesh()
    ref, _, _, _, _ = common.oracle_render("bunny", h, w, spp, 10, faces=mesh)
    assert common.rel_l2(img, ref) <= 1e-3
    assert (img == ref).all(axis=2).mean() > 0.98  # (the device's tan() may move the camera frame by an ulp)
    gpu, _, _, _, _ = common.gpu_render("bunny", h, w, spp, 10, faces=mesh)
    assert np.array_equal(gpu, ref)  # the Python binding on the same mesh: bit for bit


def test_birthday_program_against_the_oracle(tmp_path):
    """scenes/birthday.cu through Main with a lossless stand-in for resources/earthmap.jpg, against the oracle's
    render with the same texels, at the tolerance the image-textured sphere is held to everywhere (acosf / atan2f of
    two libms pick the texel, sphere.cu:60-63): 1e-3 relative L2."""
    from rtmi import scenes
    (tmp_path / "resources").mkdir()
    earth = scenes.procedural_earthmap(128, 256)
    write_ppm(str(tmp_path / "resources" / "earthmap.jpg"), earth)
    h, w, spp = 64, 64, 8
    img, _ = run_scene("birthday", tmp_path, h, w, spp)
    ref, _, _, _, _ = common.oracle_render("birthday", h, w, spp, 10, earthmap=earth)
    rel = common.rel_l2(img, ref)
    assert rel <= 1e-3, rel
    centre = img[24:40, 24:40]  # the textured sphere: the map's colours, not a flat tone
    assert centre.std(axis=(0, 1)).max() > 0.02


def test_birthday_program_with_jpeg_stand_in(tmp_path):
    """The same program on a real baseline JPEG (tools/make_assets.py: 4:4:4, so that no chroma up-sampling filter
    is involved), decoded by the stand-in's own decoder and, for the oracle, by PIL: two IDCTs may differ by a level
    or two per texel, so this one is held to 1e-2."""
    assets(tmp_path)
    from PIL import Image
    h, w, spp = 64, 64, 8
    img, _ = run_scene("birthday", tmp_path, h, w, spp)
    rgb = np.asarray(Image.open(str(tmp_path / "resources" / "earthmap.jpg")).convert("RGB"))
    rgba = np.concatenate([rgb, np.full(rgb.shape[:2] + (1,), 255, np.uint8)], axis=2)
    ref, _, _, _, _ = common.oracle_render("birthday", h, w, spp, 10, earthmap=rgba)
    assert np.isfinite(img).all() and img.max() <= 1.0 and img.min() >= 0.0
    assert common.rel_l2(img, ref) <= 1e-2


def quilt_model(tmp_path):
    """A small OBJ with texture coordinates and two materials (the first with a map_Kd texture), its MTL and the
    texture as resources/textures/quilt.ppm; returns (faces0, uvs0, faces1, uvs1, rgba)."""
    res = tmp_path / "resources"
    (res / "models").mkdir(parents=True)
    (res / "textures").mkdir()
    rng = np.random.default_rng(21)
    n = 6
    g = np.linspace(-1.0, 1.0, n + 1)
    verts, uvs = [], []
    for i in range(n + 1):
        for j in range(n + 1):
            y = 0.35 + 0.25 * np.sin(2.1 * g[i]) * np.cos(1.7 * g[j]) + rng.uniform(-0.03, 0.03)
            verts.append(np.array([g[i], y, g[j] - 0.4], dtype=np.float32))
            uvs.append(np.array([i / n, j / n], dtype=np.float32))
    vid = lambda i, j: i * (n + 1) + j  # noqa: E731
    tris = []
    for i in range(n):
        for j in range(n):
            tris.append((vid(i, j), vid(i + 1, j), vid(i + 1, j + 1)))
            tris.append((vid(i, j), vid(i + 1, j + 1), vid(i, j + 1)))
    half = len(tris) // 2
    with open(str(res / "models" / "quilt.obj"), "w") as f:
        f.write("mtllib quilt.mtl\n")
        for p in verts:
            f.write("v %.9g %.9g %.9g\n" % (p[0], p[1], p[2]))
        for t in uvs:
            f.write("vt %.9g %.9g\n" % (t[0], t[1]))
        f.write("usemtl cloth\n")
        for a, b, c in tris[:half]:
            f.write("f %d/%d %d/%d %d/%d\n" % (a + 1, a + 1, b + 1, b + 1, c + 1, c + 1))
        f.write("usemtl brass\n")
        for a, b, c in tris[half:]:
            # negative (relative) indices and a quad-free polygon syntax the loader must accept too
            k = len(verts)
            f.write("f %d/%d %d/%d %d/%d\n" % (a - k, a - k, b - k, b - k, c - k, c - k))
    with open(str(res / "models" / "quilt.mtl"), "w") as f:
        f.write("newmtl cloth\nKd 1 1 1\nmap_Kd some/dir/quilt.ppm\nnewmtl brass\nKd 0.8 0.7 0.5\n")
    from rtmi import scenes
    rgba = scenes.procedural_earthmap(32, 48)
    write_ppm(str(res / "textures" / "quilt.ppm"), rgba)
    V, T = np.asarray(verts), np.asarray(uvs)
    out = []
    for part in (tris[:half], tris[half:]):
        idx = np.asarray(part)
        out += [V[idx].astype(np.float32), T[idx].astype(np.float32)]
    return out + [rgba]


def test_textured_model_program_against_the_oracle(tmp_path):
    """tests/scenes/textured_model.cu: Model<true> with `usemtl` / `mtllib` / `map_Kd` (model.h:76-91) feeding
    BVH<Face<true>, AABB> under Lambertian(ImageTexture) and under Metal, through Main -- against the oracle's render
    of the same faces, texture coordinates and texels.  No libm call decides a texel here (barycentric u, v:
    bvh.cuh:41-45), so the frames agree except where the device's tan() moved the camera frame by an ulp."""
    from rtmi.scenes import v3, PI_D
    import oraclelib
    f0, t0, f1, t1, rgba = quilt_model(tmp_path)
    h, w, spp = 40, 56, 4
    img, log = run_scene("textured_model", tmp_path, h, w, spp)
    assert "loading texture" in log and "quilt.ppm" in log
    o = oraclelib.OracleBuilder(1024)
    o.camera_pinhole(v3(0.2, 1.1, 2.6), v3(0, 0.5, 0), v3(0, 1, 0), PI_D / 3, w / h)
    o.parallelogram([v3(-4, 0, -4), v3(4, 0, -4), v3(-4, 0, 4)], o.lambertian(v3(0.55, 0.55, 0.5)))
    o.sky()
    o.bvh(f0, o.lambertian_tex(o.image_texture(rgba)), uvs=t0.reshape(-1, 6))
    o.bvh(f1, o.metal(v3(0.8, 0.75, 0.6), 0.1), uvs=t1.reshape(-1, 6))
    ref, _, _, _ = o.render(h, w, spp, 10)
    assert common.rel_l2(img, ref) <= 1e-3
    assert (img == ref).all(axis=2).mean() > 0.97


def test_run_time_overrides_and_determinism(tmp_path):
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    a, _ = run_scene("cornell_box", tmp_path / "a", 40, 40, 3, depth=5, extra_env={"RT_SEED": "7"})
    b, _ = run_scene("cornell_box", tmp_path / "b", 40, 40, 3, depth=5, extra_env={"RT_SEED": "7"})
    c, _ = run_scene("cornell_box", tmp_path / "b", 40, 40, 3, depth=5, extra_env={"RT_SEED": "8"})
    assert np.array_equal(a, b) and not np.array_equal(a, c)


def test_reference_sample_split_mode_single_rank(tmp_path):
    """RT_DIST_MODE=spp with one rank: raw sums + root-side sqrt(clamp(sum/spp)) must equal the
    kernel's own post-process (utils.cu:126-129 vs ray_tracing.cu:78-83)."""
    (tmp_path / "a").mkdir()
    (tmp_path / "b").mkdir()
    a, _ = run_scene("spheres", tmp_path / "a", 40, 48, 3)
    b, log = run_scene("spheres", tmp_path / "b", 40, 48, 3, extra_env={"RT_DIST_MODE": "spp"})
    assert "sample split" in log
    assert np.array_equal(a, b)


def test_nested_lists_program(tmp_path):
    """tests/scenes/nested_lists.cu (a scene program of this repository, same API as the reference's
    four): HitableLists appended to HitableLists, walked by the flatten kernel and replayed through
    rtmi_list_begin/end.  The frame equals the same world recorded through the Python binding."""
    import torch
    import rtmi
    from rtmi.scenes import v3, PI_D
    h, w, spp = 40, 52, 4
    img, _ = run_scene("nested_lists", tmp_path, h, w, spp)
    b = rtmi.SceneBuilder(1024)  # Main() seeds with 1024
    b.camera_pinhole(v3(0, 0.8, 2.2), v3(0, 0.6, -1), v3(0, 1, 0), PI_D / 3, w / h)
    m0, m1 = b.lambertian(v3(0.2, 0.6, 0.8)), b.lambertian(v3(0.8, 0.3, 0.2))
    m2, m3 = b.metal(v3(0.7, 0.7, 0.6), 0.25), b.dielectric(v3(1, 1, 1), 1.5)
    m4 = b.diffuse_light(b.constant_texture(v3(3, 3, 3)))

    def wall(z, m, dx):
        b.parallelogram([v3(-1.5 + dx, -0.2, z), v3(1.5 + dx, -0.2, z), v3(-1.5 + dx, 1.8, z)], m)

    b.sky()
    b.list_begin()
    wall(-2.0, m0, 0.0)
    b.list_begin()
    wall(-2.0, m1, np.float32(0.7))
    b.sphere(v3(-0.6, 0.5, -1.0), 0.45, m2)
    b.list_begin()
    b.parallelepiped([v3(0.3, 0.0, -1.4), v3(0.9, 0.0, -1.4), v3(0.3, 0.7, -1.4), v3(0.3, 0.0, -0.8)], m3)
    b.parallelogram([v3(0.3, 0.0, -0.8), v3(0.9, 0.0, -0.8), v3(0.3, 0.7, -0.8)], m1)
    b.list_end()
    b.list_end()
    b.triangle([v3(-1.4, 1.2, -1.9), v3(-0.4, 1.2, -1.9), v3(-0.9, 1.9, -1.9)], m4)
    b.list_end()
    b.list_begin()
    b.list_end()
    b.list_begin()
    b.sphere(v3(0, -100.2, -1), 100.0, m0)
    b.list_end()
    wall(-2.0, m2, np.float32(-0.9))
    assert b.stats()["world"] == 5
    b.commit()
    R = rtmi.Renderer(b, h, w, spp, 10).init_rng()
    R.render()
    ref, _ = R.untile()
    torch.cuda.synchronize()
    ref = ref.cpu().numpy()
    assert common.rel_l2(img, ref) <= 1e-3
    assert (img == ref).all(axis=2).mean() > 0.98


def test_two_ranks_rendezvous_through_the_id_file(tmp_path):
    """DistributedMain with WORLD_SIZE=2: both ranks must obtain rank 0's ncclUniqueId through the
    token-named file and enter ncclCommInitRank together, although a stale id file of another launch
    lies in the rendezvous directory.  On a box with one GPU RCCL then refuses the communicator
    ("Duplicate GPU detected") -- which both ranks can only report if the hand-shake worked; with two
    GPUs the job completes and rank 0's frame equals the single-rank frame."""
    import torch
    exe = os.path.join(BIN, "spheres")
    if not os.path.exists(exe):
        pytest.skip("%s not built" % exe)
    h, w, spp = 32, 48, 2
    rdv = tmp_path / "rdv"
    rdv.mkdir()
    (rdv / "rtmi_rccl_id_run-other-127.0.0.1-29999").write_bytes(b"RTMIID1\0" + b"x" * 248)  # someone else's
    procs = []
    for r in range(2):
        d = tmp_path / ("r%d" % r)
        d.mkdir()
        env = dict(os.environ, RT_HEIGHT=str(h), RT_WIDTH=str(w), RT_SPP=str(spp), RT_DUMP=str(d / "frame.bin"),
                   RANK=str(r), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RT_RUN_ID="pytest-%d" % os.getpid(),
                   RT_RENDEZVOUS_DIR=str(rdv), NCCL_DEBUG="WARN", HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([exe], cwd=str(d), env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = [p.communicate(timeout=300)[0] for p in procs]
    if torch.cuda.device_count() >= 2:
        assert all(p.returncode == 0 for p in procs), logs
        one, _ = run_scene("spheres", tmp_path, h, w, spp)
        two = np.fromfile(str(tmp_path / "r0" / "frame.bin"), dtype=np.float32).reshape(h, w, 3)
        assert np.array_equal(one, two)
    else:
        for log in logs:
            assert "timed out waiting" not in log
            assert "Duplicate GPU detected" in log, log[-2000:]
    assert not [f for f in os.listdir(str(rdv)) if "pytest" in f and not f.endswith(".tmp")] or torch.cuda.device_count() < 2
