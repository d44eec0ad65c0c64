"""Scene programs: the workloads of /root/reference/scenes/*.cu restated against a
builder protocol.

A *builder* is any object exposing the reference's constructors as methods
(``lambertian(rgb)``, ``sphere(center, radius, mat)``, ``parallelogram(p, mat)``,
``parallelepiped_lengths(lengths, mat, transform)``, ``sky()``, ``bvh(faces, mat)``,
``camera_pinhole(...)``, ``random_float(mn, mx)`` ...).  The product's builder
(``rtmi.SceneBuilder``) forwards to the C-ABI of librtmi.so; the test suite supplies a
second builder over its CPU checker.  Scene programs are *inputs*: they contain no
rendering arithmetic, only the constants of the reference's ``InitWorld`` kernels.

All vector constants are rounded to binary32 exactly as ``glm::vec3(double literals)``
would be; ``rotate_y`` restates ``glm::rotateY`` (glm/gtx/rotate_vector.inl) in
binary32 arithmetic.
"""
import math

import numpy as np

f32 = np.float32
PI_F = f32(3.14159265358979323846264338327950288)  # glm::pi<float>()
PI_D = 3.14159265358979323846264338327950288  # glm::pi<double>()


def v3(x, y, z):
    return np.array([x, y, z], dtype=np.float32)


def rotate_y(p, angle):
    """glm::rotateY(vec3, float): Result.x = v.x*Cos + v.z*Sin; Result.z = -v.x*Sin + v.z*Cos.

    cosf/sinf are taken as the correctly rounded binary32 values of the binary32
    angle (libm double result narrowed once)."""
    angle = f32(angle)
    c = f32(math.cos(float(angle)))
    s = f32(math.sin(float(angle)))
    p = np.asarray(p, dtype=np.float32)
    x = f32(f32(p[0] * c) + f32(p[2] * s))
    z = f32(f32(f32(-p[0]) * s) + f32(p[2] * c))
    return v3(x, p[1], z)


# --------------------------------------------------------------------------- cornell_box
def cornell_box(b, aspect):
    """scenes/cornell_box.cu:35-69.  World order: Sky, 6 parallelograms, 2 boxes."""
    b.camera_pinhole(v3(278, 278, -800), v3(278, 278, 0), v3(0, 1, 0), PI_D * 2 / 9, aspect)
    red = b.lambertian(v3(0.65, 0.05, 0.05))
    white = b.lambertian(v3(0.73, 0.73, 0.73))
    green = b.lambertian(v3(0.12, 0.45, 0.15))
    light = b.diffuse_light(b.constant_texture(v3(1, 1, 1)))
    b.sky()
    P = [
        v3(0, 0, 0), v3(0, 555, 0), v3(0, 0, 555),
        v3(555, 0, 0), v3(555, 555, 0), v3(555, 0, 555),
        v3(213, 554, 332), v3(213, 554, 227), v3(343, 554, 332),
        v3(0, 0, 0), v3(555, 0, 0), v3(0, 0, 555),
        v3(0, 0, 555), v3(0, 555, 555), v3(555, 0, 555),
        v3(0, 555, 0), v3(555, 555, 0), v3(0, 555, 555),
    ]
    b.parallelogram(P[0:3], red)
    b.parallelogram(P[3:6], green)
    b.parallelogram(P[6:9], light)
    b.parallelogram(P[9:12], white)
    b.parallelogram(P[12:15], white)
    b.parallelogram(P[15:18], white)
    a1 = f32(f32(-PI_F) * f32(0.1))  # -glm::pi<float>() * 0.1f
    a2 = f32(PI_F / f32(12.0))  # glm::pi<float>() / 12.f
    b.parallelepiped_lengths(v3(165, 165, 165), white, lambda p: rotate_y(p, a1) + v3(130, 0, 165))
    b.parallelepiped_lengths(v3(165, 330, 165), white, lambda p: rotate_y(p, a2) + v3(265, 0, 295))


# --------------------------------------------------------------------------- spheres
def spheres(b, aspect):
    """scenes/spheres.cu:34-81.  The layout is drawn from pixel 0's RNG state
    (scenes/spheres.cu:105: ``d_states`` == &states[0]) through ``b.random_float``;
    constructor arguments are drawn left to right (quirk g6).  ``(center -
    vec3(4,0.2,0)).length()`` is GLM's component count (3), so every small sphere
    is kept (quirk g4): 4 big spheres, Sky, then 22*22 small ones = 489 entries."""
    b.camera_pinhole(v3(13, 2, 3), v3(0, 0, 0), v3(0, 1, 0), PI_D * 1 / 9, aspect)
    ground = b.lambertian(v3(0.5, 0.5, 0.5))
    b.sphere(v3(0, -1000, 0), 1000.0, ground)
    m1 = b.dielectric(v3(1, 1, 1), 1.5)
    b.sphere(v3(0, 1, 0), 1.0, m1)
    m2 = b.lambertian(v3(0.4, 0.2, 0.1))
    b.sphere(v3(-4, 1, 0), 1.0, m2)
    m3 = b.metal(v3(0.7, 0.6, 0.5), 0.0)
    b.sphere(v3(4, 1, 0), 1.0, m3)
    b.sky()
    for a in range(-11, 11):
        for bb in range(-11, 11):
            choose_mat = b.random_float(0, 1)
            # vec3(a + CudaRandomFloat(0, 0.9), 0.2, b + CudaRandomFloat(0, 0.9)):
            # int + float in binary32; CudaRandomFloat(0, 0.9, s): max is a float
            # parameter, so 0.9 -> 0.9f.
            cx = f32(f32(a) + b.random_float(0, 0.9))
            cz = f32(f32(bb) + b.random_float(0, 0.9))
            center = v3(cx, 0.2, cz)
            r = b.random_float(0, 1)
            g = b.random_float(0, 1)
            bl = b.random_float(0, 1)
            albedo = v3(r, g, bl)
            # `choose_mat < 0.8`: float vs double literal compares in double
            if float(choose_mat) < 0.8:
                mat = b.lambertian(albedo)
            elif float(choose_mat) < 0.95:
                fuzz = b.random_float(0, 0.5)
                mat = b.metal(albedo, fuzz)
            else:
                mat = b.dielectric(v3(1, 1, 1), 1.5)
            b.sphere(center, 0.2, mat)


# --------------------------------------------------------------------------- bunny
def bunny(b, aspect, faces, k_min=2048):
    """scenes/bunny.cu:44-62.  World order: Parallelogram, Sky, BVH (mesh appended by
    InitModel after InitWorld).  ``faces`` is an (n,3,3) float32 array standing in for
    resources/bunny.obj, which the reference does not ship (.gitignore:2)."""
    b.camera_pinhole(v3(-0.025, 0.1, -0.5), v3(-0.025, 0.1, 0), v3(0, 1, 0), PI_D * 2 / 9, aspect)
    # vec3(-0.025 - 0.5, 0.1 - 0.5, 2): double arithmetic, then narrowed
    P = [v3(-0.025 - 0.5, 0.1 - 0.5, 2), v3(-0.025 + 0.5, 0.1 - 0.5, 2), v3(-0.025 - 0.5, 0.1 + 0.5, 2)]
    green = b.lambertian(v3(0.12, 0.45, 0.15))
    b.parallelogram(P, green)
    b.sky()
    white = b.lambertian(v3(1, 1, 1))
    b.bvh(faces, white, k_min=k_min)


def procedural_bunny_mesh(n=76, seed=7):
    """Deterministic stand-in for the absent Stanford bunny: a bumpy closed blob of 12*n*n
    triangles (default 69,312; the Stanford mesh has 69,451) filling the bunny's bounding box
    [-0.095,0.061]x[0.033,0.187]x[-0.062,0.059] so the camera of scenes/bunny.cu:47 frames it.
    Cube-sphere tessellation (six n x n grids pushed onto the unit sphere): no polar triangle
    fans, all faces of similar size, like a scanned mesh.  float64 table arithmetic narrowed
    once to float32; a fixed LCG instead of an RNG library, so it is reproducible anywhere."""
    cx, cy, cz = -0.017, 0.110, -0.0015
    rx, ry, rz = 0.070, 0.069, 0.054
    state = seed & 0xFFFFFFFF
    waves = []
    for _ in range(10):
        vals = []
        for _ in range(7):
            state = (1664525 * state + 1013904223) & 0xFFFFFFFF
            vals.append(state / 4294967296.0)
        waves.append(vals)
    g = (np.arange(n + 1, dtype=np.float64) / n) * 2.0 - 1.0
    A, B = np.meshgrid(g, g, indexing="ij")  # (n+1, n+1)
    one = np.ones_like(A)
    cube_faces = [(one, A, B), (-one, B, A), (B, one, A), (A, -one, B), (A, B, one), (B, A, -one)]
    tris = []
    for (X, Y, Z) in cube_faces:
        L = np.sqrt(X * X + Y * Y + Z * Z)
        dx, dy, dz = X / L, Y / L, Z / L
        rad = np.ones_like(dx)
        for (a, bb, c, d, e, f, h) in waves:
            k = 1.0 + np.floor(a * 4.0)
            rad += 0.03 * np.sin(k * (3.0 * bb * dx + 3.0 * c * dy + 3.0 * d * dz) + 6.28 * e) * (0.5 + 0.5 * f) * (0.6 + 0.4 * h)
        P = np.stack([cx + rx * rad * dx, cy + ry * rad * dy, cz + rz * rad * dz], axis=-1).astype(np.float32)
        p00, p10, p01, p11 = P[:-1, :-1], P[1:, :-1], P[:-1, 1:], P[1:, 1:]
        tris.append(np.stack([p00, p10, p11], axis=2).reshape(-1, 3, 3))
        tris.append(np.stack([p00, p11, p01], axis=2).reshape(-1, 3, 3))
    return np.ascontiguousarray(np.concatenate(tris, axis=0).astype(np.float32))


# --------------------------------------------------------------------------- birthday
def birthday(b, aspect, earthmap_rgba):
    """scenes/birthday.cu:42-74.  ``earthmap_rgba`` (h,w,4) uint8 stands in for
    resources/earthmap.jpg, which the reference does not ship."""
    b.camera_pinhole(v3(278, 278, -800), v3(278, 278, 0), v3(0, 1, 0), PI_D * 2 / 9, aspect)
    red = b.lambertian(v3(0.65, 0.05, 0.05))
    white = b.lambertian(v3(0.73, 0.73, 0.73))
    green = b.lambertian(v3(0.12, 0.45, 0.15))
    light = b.diffuse_light(b.constant_texture(v3(1, 1, 1)))
    earth = b.lambertian_tex(b.image_texture(earthmap_rgba))
    b.sky()
    P = [
        v3(0, 0, 0), v3(0, 555, 0), v3(0, 0, 555),
        v3(555, 0, 0), v3(555, 555, 0), v3(555, 0, 555),
        v3(213, 554, 332), v3(213, 554, 227), v3(343, 554, 332),
        v3(0, 0, 0), v3(555, 0, 0), v3(0, 0, 555),
        v3(555, 555, 555), v3(0, 555, 555), v3(555, 0, 555),
        v3(0, 555, 0), v3(555, 555, 0), v3(0, 555, 555),
    ]
    b.parallelogram(P[0:3], red)
    b.parallelogram(P[3:6], green)
    b.parallelogram(P[6:9], light)
    b.parallelogram(P[9:12], white)
    b.parallelogram(P[12:15], white)
    b.parallelogram(P[15:18], white)
    b.sphere(v3(278, 278, 0), 100.0, earth)


def procedural_earthmap(h=256, w=512):
    """Deterministic equirectangular RGBA8 stand-in for earthmap.jpg (integer math only)."""
    y, x = np.mgrid[0:h, 0:w]
    land = (((x * 7 + y * 13) // 37 + (x // 29) * (y // 23)) % 5) < 2
    r = np.where(land, 40 + (x * 3 + y) % 90, 10 + (y % 30))
    g = np.where(land, 110 + (x + y * 2) % 100, 40 + (x % 50))
    bch = np.where(land, 30 + (x * 5) % 40, 150 + (x + y) % 100)
    img = np.stack([r, g, bch, np.full_like(r, 255)], axis=-1).astype(np.uint8)
    return np.ascontiguousarray(img)


# --------------------------------------------------------------------------- small synthetic scenes for tests
def sky_only(b, aspect):
    """Known-answer k1: a world holding only Sky."""
    b.camera_pinhole(v3(0, 0, 0), v3(0, 0, -1), v3(0, 1, 0), PI_D / 2, aspect)
    b.sky()


def furnace(b, aspect, rho=0.5):
    """Known-answer k5: camera inside a closed box of emitters seen through nothing —
    a Lambertian sphere (albedo rho) surrounded by a uniformly emitting enclosure."""
    b.camera_pinhole(v3(0, 0, 4), v3(0, 0, 0), v3(0, 1, 0), PI_D / 4, aspect)
    light = b.diffuse_light(b.constant_texture(v3(1, 1, 1)))
    grey = b.lambertian(v3(rho, rho, rho))
    b.sphere(v3(0, 0, 0), 1.0, grey)
    L = 10.0
    b.parallelepiped([v3(-L, -L, -L), v3(L, -L, -L), v3(-L, L, -L), v3(-L, -L, L)], light)


def mixed(b, aspect, seed=1):
    """Every primitive and material kind in one small world (test coverage of the
    list tie rules, nested boxes, triangle, metal fuzz, dielectric, defocus-less)."""
    b.camera_pinhole(v3(0, 1.5, 6), v3(0, 0.8, 0), v3(0, 1, 0), PI_D / 4, aspect)
    ground = b.lambertian(v3(0.5, 0.5, 0.5))
    b.sphere(v3(0, -100, 0), 100.0, ground)
    b.sky()
    b.sphere(v3(-1.2, 0.6, 0.5), 0.6, b.dielectric(v3(1, 1, 1), 1.5))
    b.sphere(v3(1.2, 0.6, 0.5), 0.6, b.metal(v3(0.8, 0.7, 0.6), 0.3))
    b.sphere(v3(0.0, 0.4, 1.6), 0.4, b.metal(v3(0.9, 0.9, 0.9), 0.0))
    b.parallelepiped([v3(-0.4, 0, -1.4), v3(0.4, 0, -1.4), v3(-0.4, 1.2, -1.4), v3(-0.4, 0, -0.6)],
                     b.lambertian(v3(0.2, 0.3, 0.8)))
    b.triangle([v3(-2.5, 0, -2), v3(2.5, 0, -2), v3(0, 3.0, -2.5)], b.lambertian(v3(0.7, 0.2, 0.2)))
    b.parallelogram([v3(-1, 3.0, -1), v3(1, 3.0, -1), v3(-1, 3.0, 1)],
                    b.diffuse_light(b.constant_texture(v3(4, 4, 4))))


SCENE_SEEDS = {"cornell_box": 1024, "birthday": 1024, "spheres": 10086, "bunny": 10086}
"""Main() seeds with 1024 (utils.cu:146), DistributedMain() with 10086 (utils.cu:202)."""
