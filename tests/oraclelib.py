"""ctypes binding of the CPU oracle (oracle/_build/liboracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py — never by the product package.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ORACLE_DIR = os.path.join(_ROOT, "oracle")
# ORACLE_VARIANT=fma: the contraction-on build of oracle/Makefile, for tools/contraction_study.py ONLY (it checks nothing)
_LIB_NAME = "liboracle_fma.so" if os.environ.get("ORACLE_VARIANT") == "fma" else "liboracle.so"
_LIB_PATH = os.path.join(_ORACLE_DIR, "_build", _LIB_NAME)

_TRANSFORM = C.CFUNCTYPE(None, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_void_p)


def build(force=False):
    """Compile the oracle with g++ if the shared object is missing or stale."""
    srcs = [os.path.join(_ORACLE_DIR, f) for f in ("oracle.cc", "vecmath.hpp", "xorwow.hpp", "Makefile")]
    stale = force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in srcs if os.path.exists(s))
    if stale:
        subprocess.run(["make", "-C", _ORACLE_DIR, "_build/" + _LIB_NAME], check=True, capture_output=True)
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        fp = C.POINTER(C.c_float)
        u32p = C.POINTER(C.c_uint32)
        L.orc_scene_new.restype = C.c_void_p
        L.orc_scene_free.argtypes = [C.c_void_p]
        L.orc_constant_texture.argtypes = [C.c_void_p, fp]
        L.orc_image_texture.argtypes = [C.c_void_p, C.POINTER(C.c_uint8), C.c_int, C.c_int]
        L.orc_lambertian.argtypes = [C.c_void_p, fp]
        L.orc_lambertian_tex.argtypes = [C.c_void_p, C.c_int]
        L.orc_metal.argtypes = [C.c_void_p, fp, C.c_float]
        L.orc_dielectric.argtypes = [C.c_void_p, fp, C.c_double]
        L.orc_diffuse_light.argtypes = [C.c_void_p, C.c_int]
        L.orc_add_sphere.argtypes = [C.c_void_p, fp, C.c_double, C.c_int]
        L.orc_add_triangle.argtypes = [C.c_void_p, fp, C.c_int]
        L.orc_add_parallelogram.argtypes = [C.c_void_p, fp, C.c_int]
        L.orc_add_parallelepiped.argtypes = [C.c_void_p, fp, C.c_int]
        L.orc_add_parallelepiped_lengths.argtypes = [C.c_void_p, fp, C.c_int, _TRANSFORM, C.c_void_p]
        L.orc_add_sky.argtypes = [C.c_void_p]
        L.orc_list_begin.argtypes = [C.c_void_p]
        L.orc_list_end.argtypes = [C.c_void_p]
        L.orc_add_bvh.argtypes = [C.c_void_p, fp, fp, C.c_int, C.c_int, C.c_int]
        L.orc_camera_pinhole.argtypes = [C.c_void_p, fp, fp, fp, C.c_double, C.c_double]
        L.orc_camera_defocus.argtypes = [C.c_void_p, fp, fp, fp, C.c_double, C.c_double, C.c_double, C.c_double]
        L.orc_camera_raw.argtypes = [C.c_void_p, fp, fp, fp, fp]
        L.orc_camera_get.argtypes = [C.c_void_p, fp]
        L.orc_rng_init.argtypes = [C.c_uint64, u32p, C.c_int64, C.c_int64]
        L.orc_rng_next.argtypes = [u32p]
        L.orc_rng_next.restype = C.c_uint32
        L.orc_rng_uniform.argtypes = [u32p]
        L.orc_rng_uniform.restype = C.c_float
        L.orc_random_float.argtypes = [C.c_float, C.c_float, u32p]
        L.orc_random_float.restype = C.c_float
        L.orc_rng_jump_pow2.argtypes = [u32p, C.c_int]
        L.orc_rng_step_v.argtypes = [u32p]
        L.orc_get_workload.argtypes = [C.c_int, C.c_int, C.c_int]
        L.orc_probe_hit.argtypes = [C.c_void_p, fp, fp, C.c_double, C.c_double, C.POINTER(C.c_double),
                                    C.POINTER(C.c_int)]
        L.orc_probe_scatter.argtypes = [C.c_void_p, C.c_int, fp, fp, C.c_double, fp, u32p, fp]
        L.orc_probe_camera_ray.argtypes = [C.c_void_p, C.c_double, C.c_double, u32p, fp]
        L.orc_render.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, u32p, fp, u32p,
                                 C.POINTER(C.c_int32), C.c_int64, C.c_int]
        L.orc_render.restype = C.c_uint64
        L.orc_last_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
        L.orc_post_process.argtypes = [fp, C.c_int64, C.c_int]
        _lib = L
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float))


def _f3(v):
    return np.ascontiguousarray(np.asarray(v, dtype=np.float32).reshape(-1))


def rng_init(seed, n, first=0):
    """(n, 6) uint32 array of compact XORWOW states {d, v0..v4} for pixels first..first+n."""
    st = np.zeros((n, 6), dtype=np.uint32)
    lib().orc_rng_init(C.c_uint64(seed), st.ctypes.data_as(C.POINTER(C.c_uint32)), first, n)
    return st


class OracleBuilder:
    """Builder protocol (see rtmi/scenes.py) over the oracle's constructors."""

    def __init__(self, seed=0):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_scene_new())
        self._keep = []
        self.seed = seed
        self.state0 = rng_init(seed, 1)[0].copy()  # pixel 0's RNG stream (spheres.cu:105)

    def __del__(self):
        try:
            self.L.orc_scene_free(self.h)
        except Exception:
            pass

    # textures / materials
    def constant_texture(self, rgb):
        return self.L.orc_constant_texture(self.h, _fp(_f3(rgb)))

    def image_texture(self, rgba):
        rgba = np.ascontiguousarray(rgba, dtype=np.uint8)
        return self.L.orc_image_texture(self.h, rgba.ctypes.data_as(C.POINTER(C.c_uint8)), rgba.shape[0],
                                        rgba.shape[1])

    def lambertian(self, rgb):
        return self.L.orc_lambertian(self.h, _fp(_f3(rgb)))

    def lambertian_tex(self, tex):
        return self.L.orc_lambertian_tex(self.h, tex)

    def metal(self, rgb, fuzz):
        return self.L.orc_metal(self.h, _fp(_f3(rgb)), C.c_float(float(fuzz)))

    def dielectric(self, rgb, index):
        return self.L.orc_dielectric(self.h, _fp(_f3(rgb)), float(index))

    def diffuse_light(self, tex):
        return self.L.orc_diffuse_light(self.h, tex)

    # hitables
    def sphere(self, c, r, mat):
        assert self.L.orc_add_sphere(self.h, _fp(_f3(c)), float(r), mat) == 0

    def triangle(self, p, mat):
        assert self.L.orc_add_triangle(self.h, _fp(_f3(p)), mat) == 0

    def parallelogram(self, p, mat):
        assert self.L.orc_add_parallelogram(self.h, _fp(_f3(p)), mat) == 0

    def parallelepiped(self, p, mat):
        assert self.L.orc_add_parallelepiped(self.h, _fp(_f3(p)), mat) == 0

    def parallelepiped_lengths(self, lengths, mat, transform):
        def cb(pin, pout, _user):
            o = transform(np.array([pin[0], pin[1], pin[2]], dtype=np.float32))
            pout[0], pout[1], pout[2] = float(o[0]), float(o[1]), float(o[2])

        cfn = _TRANSFORM(cb)
        self._keep.append(cfn)
        assert self.L.orc_add_parallelepiped_lengths(self.h, _fp(_f3(lengths)), mat, cfn, None) == 0

    def sky(self):
        assert self.L.orc_add_sky(self.h) == 0

    def list_begin(self):
        """A real nested HitableList object (its Hit() is the nested call of hitable_list.cu:7-25)."""
        assert self.L.orc_list_begin(self.h) == 0

    def list_end(self):
        assert self.L.orc_list_end(self.h) == 0

    def bvh(self, faces, mat, uvs=None, k_min=2048):
        faces = np.ascontiguousarray(faces, dtype=np.float32).reshape(-1, 9)
        uvp = None
        if uvs is not None:
            uvs = np.ascontiguousarray(uvs, dtype=np.float32).reshape(-1, 6)
            uvp = _fp(uvs)
        assert self.L.orc_add_bvh(self.h, _fp(faces), uvp, faces.shape[0], -1 if mat is None else mat, k_min) == 0

    # camera
    def camera_pinhole(self, pos, look_at, up, fov, aspect):
        self.L.orc_camera_pinhole(self.h, _fp(_f3(pos)), _fp(_f3(look_at)), _fp(_f3(up)), float(fov), float(aspect))

    def camera_defocus(self, pos, look_at, up, fov, aspect, aperture, focus):
        self.L.orc_camera_defocus(self.h, _fp(_f3(pos)), _fp(_f3(look_at)), _fp(_f3(up)), float(fov), float(aspect),
                                  float(aperture), float(focus))

    def camera_raw(self, pos, llc, horiz, vert):
        self.L.orc_camera_raw(self.h, _fp(_f3(pos)), _fp(_f3(llc)), _fp(_f3(horiz)), _fp(_f3(vert)))

    def camera_get(self):
        out = np.zeros(21, dtype=np.float32)
        self.L.orc_camera_get(self.h, _fp(out))
        return out.reshape(7, 3)

    # scene-time RNG draws from pixel 0's stream
    def random_float(self, mn, mx):
        return np.float32(self.L.orc_random_float(C.c_float(float(np.float32(mn))), C.c_float(float(np.float32(mx))),
                                                  self.state0.ctypes.data_as(C.POINTER(C.c_uint32))))

    # probes
    def probe_hit(self, o, d, t_from=1e-3, t_to=float("inf")):
        out = (C.c_double * 6)()
        mat = C.c_int(-1)
        hit = self.L.orc_probe_hit(self.h, _fp(_f3(o)), _fp(_f3(d)), t_from, t_to, out, C.byref(mat))
        return bool(hit), np.array(list(out)), mat.value

    def probe_scatter(self, mat, o, d, t, n, state):
        out = np.zeros(9, dtype=np.float32)
        sc = self.L.orc_probe_scatter(self.h, mat, _fp(_f3(o)), _fp(_f3(d)), float(t), _fp(_f3(n)),
                                      state.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(out))
        return bool(sc), out

    def probe_camera_ray(self, x, y, state=None):
        if state is None:
            state = np.zeros(6, dtype=np.uint32)
        out = np.zeros(6, dtype=np.float32)
        self.L.orc_probe_camera_ray(self.h, float(x), float(y), state.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(out))
        return out

    def render(self, height, width, spp, max_depth, post=True, pixel_ids=None, threads=None, states=None):
        """Returns (rgb (H,W,3) float32, rays (H,W) uint32, states (H*W,6) uint32, total_rays).

        RNG states are curand_init(self.seed, idx, 0) per pixel; pixel 0 starts from the
        builder's ``state0``, which scene construction may have advanced (quirk g5)."""
        n = height * width
        if states is None:
            if pixel_ids is None:
                states = rng_init(self.seed, n)
            else:  # seed only the pixels that will be rendered (a full 4096^2 init takes minutes)
                states = np.zeros((n, 6), dtype=np.uint32)
                for idx in np.asarray(pixel_ids).reshape(-1):
                    states[int(idx)] = rng_init(self.seed, 1, first=int(idx))[0]
            states[0] = self.state0
        rgb = np.zeros((n, 3), dtype=np.float32)
        rays = np.zeros(n, dtype=np.uint32)
        ids_p, n_ids = None, 0
        if pixel_ids is not None:
            pixel_ids = np.ascontiguousarray(pixel_ids, dtype=np.int32)
            ids_p, n_ids = pixel_ids.ctypes.data_as(C.POINTER(C.c_int32)), pixel_ids.size
        if threads is None:
            threads = os.cpu_count() or 1
        total = self.L.orc_render(self.h, height, width, spp, max_depth, 1 if post else 0,
                                  states.ctypes.data_as(C.POINTER(C.c_uint32)), _fp(rgb),
                                  rays.ctypes.data_as(C.POINTER(C.c_uint32)), ids_p, n_ids, threads)
        return rgb.reshape(height, width, 3), rays.reshape(height, width), states, int(total)

    def counters(self):
        out = (C.c_uint64 * 3)()
        self.L.orc_last_counters(self.h, out)
        return {"rays": out[0], "bvh_boxes": out[1], "bvh_faces": out[2]}
