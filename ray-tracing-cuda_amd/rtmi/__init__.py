"""rtmi — Python binding of librtmi.so (the MI355X path-tracing hot path).

This package is plumbing over the C ABI declared in include/rtmi.h: ctypes calls
for the scene recorder / RNG / render entry points and PyTorch tensors for device
memory, streams and ``torch.distributed``.  There is no CPU rendering path: every
compute call raises ``RtmiError`` when librtmi.so or a GPU is missing.
"""
import ctypes as C
import os

import numpy as np

_PKG_DIR = os.path.dirname(os.path.abspath(__file__))
_ROOT = os.path.dirname(_PKG_DIR)  # ray-tracing-cuda_amd/
# RTMI_LIB_PATH: diagnostic builds of the same library (tools/mesh_stats.sh); never a different implementation
LIB_PATH = os.environ.get("RTMI_LIB_PATH") or os.path.join(_ROOT, "lib", "librtmi.so")

TILE = 8
STATE_WORDS = 6
MAX_DEPTH = 64


class RtmiError(RuntimeError):
    pass


class Frame(C.Structure):
    """rtmi_frame of include/rtmi.h."""
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("spp", C.c_int32), ("max_depth", C.c_int32),
                ("post_process", C.c_int32), ("rank", C.c_int32), ("world_size", C.c_int32)]


class RenderOpts(C.Structure):
    """rtmi_render_opts of include/rtmi.h (per-call scheduling options)."""
    _fields_ = [("size", C.c_int32), ("schedule", C.c_int32), ("blocks_per_cu", C.c_int32),
                ("threads_per_block", C.c_int32), ("sparse_stride", C.c_int32), ("exclusive", C.c_int32),
                ("outlier_x10", C.c_int32), ("probe_spp", C.c_int32), ("head_pct", C.c_int32 * 3),
                ("plan", C.c_int32), ("wave_priority", C.c_int32), ("lane_stride", C.c_int32),
                ("promote_after", C.c_int32), ("cost_probe", C.c_int32), ("first_pass", C.c_int32), ("reserved", C.c_int32),
                ("d_scratch", C.c_void_p), ("scratch_bytes", C.c_size_t)]


def render_opts(schedule=-1, blocks_per_cu=0, threads_per_block=0, sparse_stride=0, exclusive=-1, outlier_x10=0,
                probe_spp=0, head_pct=(0, 0, 0), scratch=None, plan=-1, wave_priority=-1, lane_stride=0, promote_after=-1,
                cost_probe=-1, first_pass=-1):
    """``scratch``: a torch uint8/int32 CUDA tensor of at least ``scratch_bytes(frame)`` bytes that holds ALL
    per-call state of the render (keep it alive until the render has finished)."""
    o = RenderOpts(C.sizeof(RenderOpts), schedule, blocks_per_cu, threads_per_block, sparse_stride, exclusive,
                   outlier_x10, probe_spp, (C.c_int32 * 3)(*head_pct), plan, wave_priority, lane_stride, promote_after,
                   cost_probe, first_pass, 0, None, 0)
    if scratch is not None:
        o.d_scratch = scratch.data_ptr()
        o.scratch_bytes = scratch.numel() * scratch.element_size()
    return o


TRANSFORM_FN = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)

_lib = None

# every symbol include/rtmi.h declares: (name, restype, argtypes)
_fp = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_frp = C.POINTER(Frame)
SYMBOLS = [
    ("rtmi_last_error", C.c_char_p, []),
    ("rtmi_version", C.c_int, []),
    ("rtmi_device_count", C.c_int, []),
    ("rtmi_scene_create", C.c_void_p, []),
    ("rtmi_scene_destroy", None, [C.c_void_p]),
    ("rtmi_constant_texture", C.c_int, [C.c_void_p, _fp]),
    ("rtmi_image_texture", C.c_int, [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int, C.c_size_t]),
    ("rtmi_lambertian", C.c_int, [C.c_void_p, _fp]),
    ("rtmi_lambertian_tex", C.c_int, [C.c_void_p, C.c_int]),
    ("rtmi_metal", C.c_int, [C.c_void_p, _fp, C.c_float]),
    ("rtmi_dielectric", C.c_int, [C.c_void_p, _fp, C.c_double]),
    ("rtmi_diffuse_light", C.c_int, [C.c_void_p, C.c_int]),
    ("rtmi_add_sphere", C.c_int, [C.c_void_p, _fp, C.c_double, C.c_int]),
    ("rtmi_add_triangle", C.c_int, [C.c_void_p, _fp, C.c_int]),
    ("rtmi_add_parallelogram", C.c_int, [C.c_void_p, _fp, C.c_int]),
    ("rtmi_add_parallelepiped", C.c_int, [C.c_void_p, _fp, C.c_int]),
    ("rtmi_add_parallelepiped_lengths", C.c_int, [C.c_void_p, _fp, C.c_int, TRANSFORM_FN, C.c_void_p]),
    ("rtmi_add_parallelepiped_faces", C.c_int, [C.c_void_p, _fp, C.c_int]),
    ("rtmi_add_sky", C.c_int, [C.c_void_p]),
    ("rtmi_list_begin", C.c_int, [C.c_void_p]),
    ("rtmi_list_end", C.c_int, [C.c_void_p]),
    ("rtmi_add_bvh", C.c_int, [C.c_void_p, _fp, _fp, C.c_int, C.c_int, C.c_int]),
    ("rtmi_camera_pinhole", C.c_int, [C.c_void_p, _fp, _fp, _fp, C.c_double, C.c_double]),
    ("rtmi_camera_defocus", C.c_int, [C.c_void_p, _fp, _fp, _fp, C.c_double, C.c_double, C.c_double, C.c_double]),
    ("rtmi_camera_raw", C.c_int, [C.c_void_p, _fp, _fp, _fp, _fp]),
    ("rtmi_camera_get", C.c_int, [C.c_void_p, _fp]),
    ("rtmi_camera_set", C.c_int, [C.c_void_p, _fp, C.c_int, C.c_double]),
    ("rtmi_scene_commit", C.c_int, [C.c_void_p]),
    ("rtmi_scene_stats", C.c_int, [C.c_void_p, C.POINTER(C.c_int64)]),
    ("rtmi_scene_bytes_per_ray", C.c_int64, [C.c_void_p]),
    ("rtmi_scene_sliver_faces", C.c_int64, [C.c_void_p]),
    ("rtmi_frame_work_items", C.c_int64, [_frp]),
    ("rtmi_frame_pixel_of", C.c_int64, [_frp, C.c_int64]),
    ("rtmi_frame_pixel_map", C.c_int, [_frp, C.POINTER(C.c_int64)]),
    ("rtmi_states_bytes", C.c_size_t, [_frp]),
    ("rtmi_tiles_bytes", C.c_size_t, [_frp]),
    ("rtmi_rng_init", C.c_int, [C.c_uint64, _frp, C.c_void_p, C.c_void_p]),
    ("rtmi_rng_host_state", C.c_int, [C.c_uint64, C.c_uint64, _u32p]),
    ("rtmi_rng_host_random_float", C.c_float, [C.c_float, C.c_float, _u32p]),
    ("rtmi_rng_set_state", C.c_int, [_frp, C.c_void_p, C.c_int64, _u32p, C.c_void_p]),
    ("rtmi_rng_get_state", C.c_int, [_frp, C.c_void_p, C.c_int64, _u32p, C.c_void_p]),
    ("rtmi_render", C.c_int, [C.c_void_p, _frp, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rtmi_render_ex", C.c_int, [C.c_void_p, _frp, C.POINTER(RenderOpts), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rtmi_render_scratch_bytes", C.c_size_t, [_frp]),
    ("rtmi_render_launch_shape", C.c_int, [C.c_void_p, _frp, C.POINTER(RenderOpts), C.POINTER(C.c_int32)]),
    ("rtmi_render_mode", C.c_int, [C.c_void_p, _frp, C.POINTER(RenderOpts), C.POINTER(C.c_int32)]),
    ("rtmi_render_status", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    ("rtmi_last_ray_total", C.c_int, [C.c_void_p, C.POINTER(C.c_uint64), C.c_void_p]),
    ("rtmi_debug_counters", C.c_int, [C.c_void_p, C.POINTER(C.c_ulonglong), C.c_void_p]),
    ("rtmi_debug_counters_ex", C.c_int, [C.c_void_p, C.c_void_p, C.POINTER(C.c_ulonglong), C.c_void_p]),
    ("rtmi_gather", C.c_int, [C.c_void_p, _frp, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]),
    ("rtmi_reduce_sum", C.c_int, [C.c_void_p, _frp, C.c_void_p, C.c_int, C.c_void_p]),
    ("rtmi_untile", C.c_int, [_frp, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rtmi_untile_u32", C.c_int, [_frp, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("rtmi_post_process", C.c_int, [C.c_void_p, C.c_int64, C.c_int, C.c_void_p]),
    ("rtmi_get_workload", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("rtmi_selftest_arithmetic", C.c_int, [C.POINTER(C.c_ulonglong)]),
    ("rtmi_set_launch", C.c_int, [C.c_int, C.c_int]),
    ("rtmi_set_schedule", C.c_int, [C.c_int]),
]


def lib():
    """Load librtmi.so; raises RtmiError when it has not been built (no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RtmiError("%s is missing: build it with __graft_entry__.build() "
                            "(make -C ray-tracing-cuda_amd/csrc); there is no CPU fallback" % LIB_PATH)
        # PyTorch bundles its own HIP runtime (torch/lib/libamdhip64.so).  Import torch first so
        # librtmi.so binds to that already-loaded runtime; loading /opt/rocm's copy first would
        # put two HIP runtimes in one process and torch would then see no device.
        import torch  # noqa: F401
        L = C.CDLL(LIB_PATH)
        for name, res, args in SYMBOLS:
            fn = getattr(L, name)
            fn.restype = res
            fn.argtypes = args
        _lib = L
    return _lib


def _check(rc, what):
    if rc < 0:
        raise RtmiError("%s failed (%d): %s" % (what, rc, lib().rtmi_last_error().decode()))
    return rc


def _f(a):
    a = np.ascontiguousarray(np.asarray(a, dtype=np.float32).reshape(-1))
    return a, a.ctypes.data_as(_fp)


def make_frame(height, width, spp, max_depth=10, post=True, rank=0, world_size=1):
    return Frame(height, width, spp, max_depth, 1 if post else 0, rank, world_size)


def work_items(frame):
    return _check(lib().rtmi_frame_work_items(C.byref(frame)), "rtmi_frame_work_items")


def pixel_map(frame):
    """int64 array: global pixel index of each work item of this shard (-1 = padding)."""
    out = np.empty(work_items(frame), dtype=np.int64)
    _check(lib().rtmi_frame_pixel_map(C.byref(frame), out.ctypes.data_as(C.POINTER(C.c_int64))),
           "rtmi_frame_pixel_map")
    return out


def get_workload(rank, world_size, spp):
    return lib().rtmi_get_workload(rank, world_size, spp)


class SceneBuilder:
    """Builder protocol of rtmi/scenes.py over the C ABI's scene recorder.

    ``seed`` seeds the host copy of pixel 0's RNG stream that scene programs may draw
    from (scenes/spheres.cu:105); ``Renderer`` stores the advanced state back into the
    device state of pixel 0 so rendering continues that stream (quirk g5)."""

    def __init__(self, seed=0):
        self.L = lib()
        self.h = C.c_void_p(self.L.rtmi_scene_create())
        self.seed = seed
        self._keep = []
        self.state0 = np.zeros(STATE_WORDS, dtype=np.uint32)
        _check(self.L.rtmi_rng_host_state(C.c_uint64(seed), C.c_uint64(0), self.state0.ctypes.data_as(_u32p)),
               "rtmi_rng_host_state")
        self.state0_fresh = self.state0.copy()

    def __del__(self):
        try:
            self.L.rtmi_scene_destroy(self.h)
        except Exception:
            pass

    def constant_texture(self, rgb):
        return _check(self.L.rtmi_constant_texture(self.h, _f(rgb)[1]), "rtmi_constant_texture")

    def image_texture(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        return _check(self.L.rtmi_image_texture(self.h, rgba.ctypes.data_as(C.POINTER(C.c_uint8)), rgba.shape[0],
                                                rgba.shape[1], rgba.shape[1] * 4), "rtmi_image_texture")

    def lambertian(self, rgb):
        return _check(self.L.rtmi_lambertian(self.h, _f(rgb)[1]), "rtmi_lambertian")

    def lambertian_tex(self, tex):
        return _check(self.L.rtmi_lambertian_tex(self.h, tex), "rtmi_lambertian_tex")

    def metal(self, rgb, fuzz):
        return _check(self.L.rtmi_metal(self.h, _f(rgb)[1], C.c_float(float(fuzz))), "rtmi_metal")

    def dielectric(self, rgb, index):
        return _check(self.L.rtmi_dielectric(self.h, _f(rgb)[1], float(index)), "rtmi_dielectric")

    def diffuse_light(self, tex):
        return _check(self.L.rtmi_diffuse_light(self.h, tex), "rtmi_diffuse_light")

    def sphere(self, c, r, mat):
        _check(self.L.rtmi_add_sphere(self.h, _f(c)[1], float(r), mat), "rtmi_add_sphere")

    def triangle(self, p, mat):
        _check(self.L.rtmi_add_triangle(self.h, _f(p)[1], mat), "rtmi_add_triangle")

    def parallelogram(self, p, mat):
        _check(self.L.rtmi_add_parallelogram(self.h, _f(p)[1], mat), "rtmi_add_parallelogram")

    def parallelepiped(self, p, mat):
        _check(self.L.rtmi_add_parallelepiped(self.h, _f(p)[1], mat), "rtmi_add_parallelepiped")

    def parallelepiped_lengths(self, lengths, mat, transform):
        def cb(pin, pout, _user):
            o = transform(np.array([pin[0], pin[1], pin[2]], dtype=np.float32))
            pout[0], pout[1], pout[2] = float(o[0]), float(o[1]), float(o[2])

        cfn = TRANSFORM_FN(cb)
        self._keep.append(cfn)
        _check(self.L.rtmi_add_parallelepiped_lengths(self.h, _f(lengths)[1], mat, cfn, None),
               "rtmi_add_parallelepiped_lengths")

    def sky(self):
        _check(self.L.rtmi_add_sky(self.h), "rtmi_add_sky")

    def list_begin(self):
        """``l = new HitableList()`` appended to the list under construction; closed by list_end()."""
        _check(self.L.rtmi_list_begin(self.h), "rtmi_list_begin")

    def list_end(self):
        _check(self.L.rtmi_list_end(self.h), "rtmi_list_end")

    def bvh(self, faces, mat, uvs=None, k_min=2048):
        faces = np.ascontiguousarray(faces, dtype=np.float32).reshape(-1, 9)
        uvp = None
        if uvs is not None:
            uvs = np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 6)
            uvp = uvs.ctypes.data_as(_fp)
        _check(self.L.rtmi_add_bvh(self.h, faces.ctypes.data_as(_fp), uvp, faces.shape[0],
                                   -1 if mat is None else mat, k_min), "rtmi_add_bvh")

    def camera_pinhole(self, pos, look_at, up, fov, aspect):
        _check(self.L.rtmi_camera_pinhole(self.h, _f(pos)[1], _f(look_at)[1], _f(up)[1], float(fov), float(aspect)),
               "rtmi_camera_pinhole")

    def camera_defocus(self, pos, look_at, up, fov, aspect, aperture, focus):
        _check(self.L.rtmi_camera_defocus(self.h, _f(pos)[1], _f(look_at)[1], _f(up)[1], float(fov), float(aspect),
                                          float(aperture), float(focus)), "rtmi_camera_defocus")

    def camera_raw(self, pos, llc, horiz, vert):
        _check(self.L.rtmi_camera_raw(self.h, _f(pos)[1], _f(llc)[1], _f(horiz)[1], _f(vert)[1]), "rtmi_camera_raw")

    def camera_get(self):
        out = np.zeros(21, dtype=np.float32)
        _check(self.L.rtmi_camera_get(self.h, out.ctypes.data_as(_fp)), "rtmi_camera_get")
        return out.reshape(7, 3)

    def random_float(self, mn, mx):
        return np.float32(self.L.rtmi_rng_host_random_float(C.c_float(float(np.float32(mn))),
                                                            C.c_float(float(np.float32(mx))),
                                                            self.state0.ctypes.data_as(_u32p)))

    def stats(self):
        out = (C.c_int64 * 8)()
        _check(self.L.rtmi_scene_stats(self.h, out), "rtmi_scene_stats")
        keys = ["world", "spheres", "parallelograms", "triangles", "bvh_faces", "bvh_nodes", "materials", "textures"]
        return dict(zip(keys, list(out)))

    def sliver_faces(self):
        return _check(self.L.rtmi_scene_sliver_faces(self.h), "rtmi_scene_sliver_faces")

    def bytes_per_ray(self):
        return _check(self.L.rtmi_scene_bytes_per_ray(self.h), "rtmi_scene_bytes_per_ray")

    def commit(self):
        _check(self.L.rtmi_scene_commit(self.h), "rtmi_scene_commit")
        return self


class Renderer:
    """One rank's share of a frame: device buffers (torch), RNG init, render, untile.

    Mirrors what the reference's ``Main``/``DistributedMain`` do around the kernel
    (utils.cu:132-242) — allocation, CudaRandomInit, PathTracing launch — but keeps
    results on the device; the caller decides when to copy or gather."""

    def __init__(self, scene, height, width, spp, max_depth=10, post=True, rank=0, world_size=1, device=None):
        import torch
        if not torch.cuda.is_available():
            raise RtmiError("no GPU visible to torch: the render path has no CPU fallback")
        self.torch = torch
        self.L = lib()
        self.scene = scene
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.frame = make_frame(height, width, spp, max_depth, post, rank, world_size)
        self.items = work_items(self.frame)
        with torch.cuda.device(self.device):
            self.states = torch.empty((STATE_WORDS, self.items), dtype=torch.int32, device=self.device)
            self.tiles = torch.empty((self.items, 3), dtype=torch.float32, device=self.device)
            self.ray_counts = torch.empty((self.items,), dtype=torch.int32, device=self.device)

    def _stream(self):
        return C.c_void_p(self.torch.cuda.current_stream(self.device).cuda_stream)

    def init_rng(self, seed=None):
        seed = self.scene.seed if seed is None else seed
        with self.torch.cuda.device(self.device):
            _check(self.L.rtmi_rng_init(C.c_uint64(seed), C.byref(self.frame), C.c_void_p(self.states.data_ptr()),
                                        self._stream()), "rtmi_rng_init")
            # pixel 0 (work item 0 of rank 0) continues the stream the scene program drew from
            if self.frame.rank == 0 and seed == self.scene.seed and \
                    not np.array_equal(self.scene.state0, self.scene.state0_fresh):
                _check(self.L.rtmi_rng_set_state(C.byref(self.frame), C.c_void_p(self.states.data_ptr()), 0,
                                                 self.scene.state0.ctypes.data_as(_u32p), self._stream()),
                       "rtmi_rng_set_state")
        return self

    def render(self, count_rays=True, opts=None):
        """Enqueue the trace kernel on torch's current stream (asynchronous).  ``opts``: a
        ``render_opts(...)`` structure with per-call scheduling options (None = the defaults)."""
        self._last_scratch = opts.d_scratch if opts is not None else None
        with self.torch.cuda.device(self.device):
            rc = self.L.rtmi_render_ex(self.scene.h, C.byref(self.frame), C.byref(opts) if opts is not None else None,
                                       C.c_void_p(self.states.data_ptr()), C.c_void_p(self.tiles.data_ptr()),
                                       C.c_void_p(self.ray_counts.data_ptr()) if count_rays else None, self._stream())
        _check(rc, "rtmi_render_ex")
        return self

    def check(self):
        """Wait for the last render and raise if it reported an incomplete frame (rtmi_render_status)."""
        with self.torch.cuda.device(self.device):
            _check(self.L.rtmi_render_status(self.scene.h, C.c_void_p(getattr(self, "_last_scratch", None)), None,
                                             self._stream()), "rtmi_render_status")
        return self

    def scratch_bytes(self):
        return int(self.L.rtmi_render_scratch_bytes(C.byref(self.frame)))

    def new_scratch(self):
        """Device memory for the per-call state of one render (``render_opts(scratch=...)``)."""
        return self.torch.zeros((self.scratch_bytes() + 7) // 8, dtype=self.torch.int64, device=self.device)

    def launch_shape(self, opts=None):
        """{workgroups, lanes per workgroup, workgroups per CU, CUs} of a render of this frame."""
        out = (C.c_int32 * 4)()
        with self.torch.cuda.device(self.device):
            _check(self.L.rtmi_render_launch_shape(self.scene.h, C.byref(self.frame),
                                                   C.byref(opts) if opts is not None else None, out),
                   "rtmi_render_launch_shape")
        return dict(zip(("blocks", "threads", "blocks_per_cu", "compute_units"), list(out)))

    def mode(self, opts=None):
        """How a render of this frame would be scheduled (rtmi_render_mode)."""
        out = (C.c_int32 * 8)()
        with self.torch.cuda.device(self.device):
            _check(self.L.rtmi_render_mode(self.scene.h, C.byref(self.frame), C.byref(opts) if opts is not None else None, out),
                   "rtmi_render_mode")
        return dict(zip(("scheduled", "first_pass_samples", "first_pass_resumed", "planned_chains", "wave_priority_every",
                         "lane_stride", "waves", "tiles"), list(out)))

    def total_rays(self, scratch=None):
        """Closest-hit queries of the last render (of the one that used ``scratch``, if given); raises when that
        render reported an incomplete frame."""
        out = C.c_uint64(0)
        with self.torch.cuda.device(self.device):
            _check(self.L.rtmi_render_status(self.scene.h, C.c_void_p(scratch.data_ptr()) if scratch is not None else None,
                                             C.byref(out), self._stream()), "rtmi_render_status")
        return out.value

    def untile(self, all_tiles=None, all_counts=None):
        """Row-major (H,W,3) image [and (H,W) ray counts] from tile-major buffers of all ranks."""
        torch = self.torch
        f = self.frame
        with torch.cuda.device(self.device):
            if all_tiles is None:  # this rank's own render: an incomplete frame must not be handed on
                self.check()
            tiles = self.tiles if all_tiles is None else all_tiles
            assert tiles.numel() == self.items * 3 * f.world_size, "expected the buffers of all ranks back to back"
            img = torch.zeros((f.height, f.width, 3), dtype=torch.float32, device=self.device)
            _check(self.L.rtmi_untile(C.byref(f), C.c_void_p(tiles.data_ptr()), C.c_void_p(img.data_ptr()),
                                      self._stream()), "rtmi_untile")
            cnt = None
            counts = self.ray_counts if (all_counts is None and f.world_size == 1) else all_counts
            if counts is not None:
                cnt = torch.zeros((f.height, f.width), dtype=torch.int32, device=self.device)
                _check(self.L.rtmi_untile_u32(C.byref(f), C.c_void_p(counts.data_ptr()), C.c_void_p(cnt.data_ptr()),
                                              self._stream()), "rtmi_untile_u32")
        return img, cnt
