// C ABI of librtmi.so (declared in include/rtmi.h).  Thin glue: argument
// checking, the host-side scene recorder, uploads, and kernel launches.  There is
// deliberately no CPU rendering path in this library.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types and enumerators only: the entry points are looked up at run time (rtmi_gather)
#include <string.h>

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include <string>
#include <unordered_map>

#include "../../include/rtmi.h"
#include "kernels.h"
#include "scene.h"
#include "xorwow.h"

using namespace rtmi;

static thread_local std::string g_err;
static int fail(int code, const std::string &msg) {
  g_err = msg;
  return code;
}
static int hip_fail(hipError_t e, const char *what) {
  return fail(RTMI_ERR_HIP, std::string(what) + ": " + hipGetErrorString(e));
}
#define HIP_TRY(expr)                                    \
  do {                                                   \
    hipError_t e__ = (expr);                             \
    if (e__ != hipSuccess) return hip_fail(e__, #expr);  \
  } while (0)

// Process-wide DEFAULTS of the scheduling parameters; every rtmi_render call works on its own copy
// (rtmi_render_ex overrides fields per call).  The RTMI_* environment variables are tuning overrides
// of the built-in defaults and are read once, when the library is first used.
static std::mutex g_tune_mu;
static RenderTuning g_tune;
static std::once_flag g_tune_once;
static int env_int(const char *name, int dflt) {
  const char *e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}
static bool valid_stride(int v) { return v >= 1 && v <= 64 && (v & (v - 1)) == 0; }
static RenderTuning default_tuning() {
  std::call_once(g_tune_once, [] {
    g_tune.schedule = 1;
    g_tune.blocks_per_cu = 0;
    g_tune.threads = 0;
    g_tune.sparse_stride = env_int("RTMI_SPARSE_STRIDE", kSparseStride);
    if (!valid_stride(g_tune.sparse_stride)) g_tune.sparse_stride = kSparseStride;
    g_tune.exclusive = env_int("RTMI_EXCLUSIVE", 1) ? 1 : 0;
    g_tune.outlier_x10 = env_int("RTMI_OUTLIER_X10", 20);
    if (g_tune.outlier_x10 < 1) g_tune.outlier_x10 = 20;
    g_tune.head_pct[0] = 80, g_tune.head_pct[1] = 55, g_tune.head_pct[2] = 30;
    g_tune.promote = env_int("RTMI_PROMOTE", 16);  // samples after which a pixel's own ray count may promote it (0: never)
    if (g_tune.promote < 0) g_tune.promote = 0;
    g_tune.probe_spp = env_int("RTMI_PROBE_SPP", 0);
    if (g_tune.probe_spp < 0 || g_tune.probe_spp > 64) g_tune.probe_spp = 0;
    g_tune.cost_probe = env_int("RTMI_COST_PROBE", 1) != 0;
    g_tune.first_pass = env_int("RTMI_FIRST_PASS", 1);  // 0: a discarded probe; 1: the frame's first spp / 16 samples; N > 1: spp / N
    if (g_tune.first_pass < 0) g_tune.first_pass = 1;
    g_tune.lane_stride = env_int("RTMI_LANE_STRIDE", 0);
    if (g_tune.lane_stride < 0 || g_tune.lane_stride > 64 || (g_tune.lane_stride & (g_tune.lane_stride - 1)) != 0) g_tune.lane_stride = 0;
    g_tune.plan = env_int("RTMI_PLAN", 1);  // list frames: planned chains instead of the queue (0 never, 1 when waves have few tiles, 2 always)
    if (g_tune.plan < 0 || g_tune.plan > 2) g_tune.plan = 1;
    g_tune.prio_every = env_int("RTMI_PRIO", 16);  // wave priorities: update interval in iterations (0: off)
    if (g_tune.prio_every < 0 || (g_tune.prio_every & (g_tune.prio_every - 1)) != 0) g_tune.prio_every = 16;
  });
  std::lock_guard<std::mutex> lk(g_tune_mu);
  return g_tune;
}

static Scene *S(rtmi_scene *s) { return reinterpret_cast<Scene *>(s); }
static const Scene *S(const rtmi_scene *s) { return reinterpret_cast<const Scene *>(s); }
static V3 v3(const float *p) { return mk(p[0], p[1], p[2]); }
static bool all_finite(const float *p, size_t n) {
  for (size_t i = 0; i < n; i++)
    if (!std::isfinite(p[i])) return false;
  return true;
}

// (why the calling thread's last make_frame refused its frame, when the reason deserves its own words)
static thread_local const char *t_frame_why = nullptr;
static const char *frame_why(const char *otherwise) { return t_frame_why ? t_frame_why : otherwise; }
static bool make_frame(const rtmi_frame *f, FrameDev *out) {
  t_frame_why = nullptr;
  if (f && (f->height > RTMI_MAX_EXTENT || f->width > RTMI_MAX_EXTENT)) {  // (a pixel's row and column share a word)
    t_frame_why = "frame larger than 65535 x 65535 (RTMI_MAX_EXTENT): a pixel's row and column share a 32-bit word";
    return false;
  }
  if (!f || f->height <= 0 || f->width <= 0 || f->spp < 0 || f->world_size <= 0 || f->rank < 0 ||
      f->rank >= f->world_size)
    return false;
  FrameDev d;
  d.height = f->height, d.width = f->width, d.spp = f->spp, d.max_depth = f->max_depth, d.post = f->post_process;
  d.k_begin = 0, d.k_end = f->spp;
  d.rank = f->rank, d.world = f->world_size;
  d.tiles_x = (f->width + RTMI_TILE - 1) / RTMI_TILE;
  d.tiles_y = (f->height + RTMI_TILE - 1) / RTMI_TILE;
  d.n_tiles = d.tiles_x * d.tiles_y;
  // every rank gets the same number of work items (rank 0's share); ranks that own
  // one tile fewer carry one inert padding tile, so gathered buffers have one stride
  d.local_tiles = (d.n_tiles + d.world - 1) / d.world;
  d.items = (int64_t)d.local_tiles * 64;
  *out = d;
  return true;
}

// jump matrices, uploaded once per device
static std::mutex g_mu;
static std::unordered_map<int, uint32_t *> g_jump;
static std::unordered_map<int, int> g_cus;
static int device_jump(uint32_t **out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_mu);
  auto it = g_jump.find(dev);
  if (it == g_jump.end()) {
    const HostJump &J = host_jump_tables();
    uint32_t *d = nullptr;
    HIP_TRY(hipMalloc(&d, J.m.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(d, J.m.data(), J.m.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    it = g_jump.emplace(dev, d).first;
  }
  *out = it->second;
  return RTMI_OK;
}

template <typename R>
static int upload(Scene *s, const std::vector<R> &v, const R **out) {
  *out = nullptr;
  if (v.empty()) return RTMI_OK;
  void *d = nullptr;
  HIP_TRY(hipMalloc(&d, v.size() * sizeof(R)));
  s->dev_allocs.push_back(d);
  HIP_TRY(hipMemcpy(d, v.data(), v.size() * sizeof(R), hipMemcpyHostToDevice));
  *out = reinterpret_cast<const R *>(d);
  return RTMI_OK;
}

extern "C" {

const char *rtmi_last_error(void) { return g_err.c_str(); }
int rtmi_version(void) { return RTMI_VERSION; }
int rtmi_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

// ------------------------------------------------------------------ scene
rtmi_scene *rtmi_scene_create(void) { return reinterpret_cast<rtmi_scene *>(new Scene()); }

static void free_device(Scene *s) {
  if (s->d_sched) (void)hipFree(s->d_sched);
  s->d_sched = nullptr;
  s->sched_bytes = 0;
  for (void *p : s->dev_allocs) (void)hipFree(p);
  s->dev_allocs.clear();
  s->d_counters = nullptr;
  s->committed = false;
}
void rtmi_scene_destroy(rtmi_scene *s) {
  if (!s) return;
  free_device(S(s));
  delete S(s);
}

int rtmi_constant_texture(rtmi_scene *s, const float rgb[3]) {
  if (!s || !rgb) return fail(RTMI_ERR_INVALID, "null argument");
  HostTex t;
  t.rgb = v3(rgb);
  S(s)->texs.push_back(t);
  return (int)S(s)->texs.size() - 1;
}
int rtmi_image_texture(rtmi_scene *s, const uint8_t *rgba, int height, int width, size_t pitch) {
  if (!s || !rgba || height <= 0 || width <= 0) return fail(RTMI_ERR_INVALID, "bad image texture");
  if (pitch == 0) pitch = (size_t)width * 4;
  if (pitch < (size_t)width * 4) return fail(RTMI_ERR_INVALID, "pitch smaller than a row");
  HostTex t;
  t.image = true;
  t.h = height, t.w = width;
  t.rgba.resize((size_t)height * width * 4);
  for (int y = 0; y < height; y++) memcpy(&t.rgba[(size_t)y * width * 4], rgba + (size_t)y * pitch, (size_t)width * 4);
  S(s)->texs.push_back(std::move(t));
  return (int)S(s)->texs.size() - 1;
}
static int add_mat(rtmi_scene *s, int kind, V3 rgb, float param, int tex) {
  HostMat m;
  m.kind = kind, m.rgb = rgb, m.param = param, m.tex = tex;
  S(s)->mats.push_back(m);
  return (int)S(s)->mats.size() - 1;
}
static bool tex_ok(rtmi_scene *s, int t) { return t >= 0 && t < (int)S(s)->texs.size(); }
int rtmi_lambertian(rtmi_scene *s, const float rgb[3]) {
  if (!s || !rgb) return fail(RTMI_ERR_INVALID, "null argument");
  return add_mat(s, MAT_LAMBERTIAN, v3(rgb), 0.f, -1);
}
int rtmi_lambertian_tex(rtmi_scene *s, int texture) {
  if (!s || !tex_ok(s, texture)) return fail(RTMI_ERR_INVALID, "unknown texture handle");
  return add_mat(s, MAT_LAMBERTIAN, splat(0.f), 0.f, texture);
}
int rtmi_metal(rtmi_scene *s, const float rgb[3], float fuzz) {
  if (!s || !rgb) return fail(RTMI_ERR_INVALID, "null argument");
  return add_mat(s, MAT_METAL, v3(rgb), fuzz < 1 ? fuzz : 1, -1);  // metal.cu:10
}
int rtmi_dielectric(rtmi_scene *s, const float rgb[3], double refractive_index) {
  if (!s || !rgb) return fail(RTMI_ERR_INVALID, "null argument");
  return add_mat(s, MAT_DIELECTRIC, v3(rgb), (float)refractive_index, -1);
}
int rtmi_diffuse_light(rtmi_scene *s, int texture) {
  if (!s || !tex_ok(s, texture)) return fail(RTMI_ERR_INVALID, "unknown texture handle");
  return add_mat(s, MAT_LIGHT, splat(0.f), 0.f, texture);
}

// Every HitableList -- the world and each nested one -- holds at most kMaxHitables entries
// (hitable_list.cuh:10,20); a nested list counts as ONE entry of its parent.  Nested lists are
// recorded inlined at their position, which gives the same closest hit (DESIGN.md "List flattening").
static int count_entry(rtmi_scene *s) {
  if (S(s)->list_counts.back() >= RTMI_MAX_HITABLES)
    return fail(RTMI_ERR_CAPACITY, "HitableList::kMaxHitables (1024) exceeded");
  S(s)->list_counts.back()++;
  return RTMI_OK;
}
static int append(rtmi_scene *s, const HostObj &o) {
  int rc = count_entry(s);
  if (rc) return rc;
  S(s)->world.push_back(o);
  S(s)->committed = false;
  return RTMI_OK;
}
static bool mat_ok(rtmi_scene *s, int m) { return m >= 0 && m < (int)S(s)->mats.size(); }

int rtmi_add_sphere(rtmi_scene *s, const float c[3], double radius, int material) {
  if (!s || !c || !mat_ok(s, material)) return fail(RTMI_ERR_INVALID, "bad sphere arguments");
  if (!all_finite(c, 3) || !std::isfinite(radius)) return fail(RTMI_ERR_INVALID, "non-finite sphere");
  HostObj o{};
  o.kind = OBJ_SPHERE, o.mat = material, o.p[0] = v3(c), o.radius = radius;
  return append(s, o);
}
int rtmi_add_triangle(rtmi_scene *s, const float p[9], int material) {
  if (!s || !p || !mat_ok(s, material)) return fail(RTMI_ERR_INVALID, "bad triangle arguments");
  if (!all_finite(p, 9)) return fail(RTMI_ERR_INVALID, "non-finite triangle corner");
  HostObj o{};
  o.kind = OBJ_TRI, o.mat = material;
  for (int i = 0; i < 3; i++) o.p[i] = v3(p + 3 * i);
  return append(s, o);
}
int rtmi_add_parallelogram(rtmi_scene *s, const float p[9], int material) {
  if (!s || !p || !mat_ok(s, material)) return fail(RTMI_ERR_INVALID, "bad parallelogram arguments");
  if (!all_finite(p, 9)) return fail(RTMI_ERR_INVALID, "non-finite parallelogram corner");
  HostObj o{};
  o.kind = OBJ_PGRAM, o.mat = material;
  for (int i = 0; i < 3; i++) o.p[i] = v3(p + 3 * i);
  return append(s, o);
}
int rtmi_add_parallelepiped(rtmi_scene *s, const float p[12], int material) {
  if (!s || !p || !mat_ok(s, material)) return fail(RTMI_ERR_INVALID, "bad parallelepiped arguments");
  if (!all_finite(p, 12)) return fail(RTMI_ERR_INVALID, "non-finite parallelepiped corner");
  HostObj o{};
  o.kind = OBJ_BOX, o.mat = material;
  V3 c[4], corners[8];
  for (int i = 0; i < 4; i++) c[i] = v3(p + 3 * i);
  box_from_points(c, corners);
  box_faces(corners, o.p);
  return append(s, o);
}
int rtmi_add_parallelepiped_lengths(rtmi_scene *s, const float lengths[3], int material, rtmi_transform_fn transform,
                                    void *user) {
  if (!s || !lengths || !transform || !mat_ok(s, material))
    return fail(RTMI_ERR_INVALID, "bad parallelepiped arguments");
  // parallelepiped.cu:37-52: axis corners from the lengths, then the user's transform
  float p[4][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}, {0, 0, 0}}, q[4][3];
  for (int i = 1; i <= 3; i++) p[i][i - 1] = lengths[i - 1];
  for (int i = 0; i < 4; i++)
    for (int k = 0; k < 3; k++) q[i][k] = lengths[k];
  for (int i = 1; i <= 3; i++) q[i][i - 1] = 0;
  HostObj o{};
  o.kind = OBJ_BOX, o.mat = material;
  V3 corners[8];
  for (int i = 0; i < 4; i++) {
    float t[3];
    transform(p[i], t, user);
    corners[i] = v3(t);
    transform(q[i], t, user);
    corners[4 + i] = v3(t);
  }
  for (int i = 0; i < 8; i++)
    if (!std::isfinite(corners[i].x) || !std::isfinite(corners[i].y) || !std::isfinite(corners[i].z))
      return fail(RTMI_ERR_INVALID, "the transform produced a non-finite parallelepiped corner");
  box_faces(corners, o.p);
  return append(s, o);
}
int rtmi_add_parallelepiped_faces(rtmi_scene *s, const float faces[54], int material) {
  if (!s || !faces || !mat_ok(s, material)) return fail(RTMI_ERR_INVALID, "bad parallelepiped arguments");
  if (!all_finite(faces, 54)) return fail(RTMI_ERR_INVALID, "non-finite parallelepiped corner");
  HostObj o{};
  o.kind = OBJ_BOX, o.mat = material;
  for (int i = 0; i < 18; i++) o.p[i] = v3(faces + 3 * i);
  return append(s, o);
}
int rtmi_list_begin(rtmi_scene *s) {
  if (!s) return fail(RTMI_ERR_INVALID, "null scene");
  int rc = count_entry(s);  // the nested list is one entry of the list it is appended to
  if (rc) return rc;
  S(s)->list_counts.push_back(0);
  S(s)->committed = false;
  return RTMI_OK;
}
int rtmi_list_end(rtmi_scene *s) {
  if (!s) return fail(RTMI_ERR_INVALID, "null scene");
  if (S(s)->list_counts.size() < 2) return fail(RTMI_ERR_INVALID, "rtmi_list_end without rtmi_list_begin");
  S(s)->list_counts.pop_back();
  return RTMI_OK;
}
int rtmi_add_sky(rtmi_scene *s) {
  if (!s) return fail(RTMI_ERR_INVALID, "null scene");
  HostObj o{};
  o.kind = OBJ_SKY;
  return append(s, o);
}
int rtmi_add_bvh(rtmi_scene *s, const float *faces, const float *uvs, int n, int material, int leaf_max) {
  if (!s || n < 0 || (n > 0 && !faces)) return fail(RTMI_ERR_INVALID, "bad bvh arguments");
  if (material >= (int)S(s)->mats.size()) return fail(RTMI_ERR_INVALID, "unknown material handle");
  // The reference would build such a mesh and simply never hit the face; here a non-finite coordinate would
  // poison the bounds of the search tree, so the mesh is refused (the caller drops the face).
  if (n > 0 && !all_finite(faces, (size_t)n * 9)) return fail(RTMI_ERR_INVALID, "non-finite face coordinate in the mesh");
  HostBvh b;
  b.n = n, b.mat = material, b.leaf_max = leaf_max > 0 ? leaf_max : 2048;
  b.faces.assign(faces, faces + (size_t)n * 9);
  if (uvs) b.uvs.assign(uvs, uvs + (size_t)n * 6);
  S(s)->bvhs.push_back(std::move(b));
  HostObj o{};
  o.kind = OBJ_BVH, o.bvh = (int)S(s)->bvhs.size() - 1;
  return append(s, o);
}

int rtmi_camera_pinhole(rtmi_scene *s, const float pos[3], const float look_at[3], const float up[3], double fov,
                        double aspect) {
  if (!s || !pos || !look_at || !up) return fail(RTMI_ERR_INVALID, "null argument");
  camera_pinhole(*S(s), v3(pos), v3(look_at), v3(up), fov, aspect);
  S(s)->committed = false;
  return RTMI_OK;
}
int rtmi_camera_defocus(rtmi_scene *s, const float pos[3], const float look_at[3], const float up[3], double fov,
                        double aspect, double aperture, double focus_distance) {
  if (!s || !pos || !look_at || !up) return fail(RTMI_ERR_INVALID, "null argument");
  camera_defocus(*S(s), v3(pos), v3(look_at), v3(up), fov, aspect, aperture, focus_distance);
  S(s)->committed = false;
  return RTMI_OK;
}
int rtmi_camera_raw(rtmi_scene *s, const float pos[3], const float llc[3], const float horiz[3], const float vert[3]) {
  if (!s || !pos || !llc || !horiz || !vert) return fail(RTMI_ERR_INVALID, "null argument");
  camera_raw(*S(s), v3(pos), v3(llc), v3(horiz), v3(vert));
  S(s)->committed = false;
  return RTMI_OK;
}
int rtmi_camera_set(rtmi_scene *s, const float f[21], int is_defocus, double lens_radius) {
  if (!s || !f) return fail(RTMI_ERR_INVALID, "null argument");
  CameraDev &c = S(s)->cam;
  c.position = v3(f), c.llc = v3(f + 3), c.horizontal = v3(f + 6), c.vertical = v3(f + 9);
  c.u = v3(f + 12), c.v = v3(f + 15);
  S(s)->cam_w = v3(f + 18);
  c.defocus = is_defocus ? 1 : 0;
  c.lens_radius = (float)lens_radius;  // DiskRand(float radius), camera.cu:64,74
  S(s)->has_camera = true;
  S(s)->committed = false;
  return RTMI_OK;
}
int rtmi_camera_get(const rtmi_scene *s, float out[21]) {
  if (!s || !out || !S(s)->has_camera) return fail(RTMI_ERR_INVALID, "scene has no camera");
  const CameraDev &c = S(s)->cam;
  const V3 vs[7] = {c.position, c.llc, c.horizontal, c.vertical, c.u, c.v, S(s)->cam_w};
  for (int i = 0; i < 7; i++) out[i * 3] = vs[i].x, out[i * 3 + 1] = vs[i].y, out[i * 3 + 2] = vs[i].z;
  return RTMI_OK;
}

int rtmi_scene_commit(rtmi_scene *sp) {
  if (!sp) return fail(RTMI_ERR_INVALID, "null scene");
  Scene *s = S(sp);
  if (s->list_counts.size() != 1) return fail(RTMI_ERR_INVALID, "a nested list is still open (rtmi_list_end missing)");
  if (rtmi_device_count() <= 0) return fail(RTMI_ERR_NO_DEVICE, "no HIP device: librtmi has no CPU fallback");
  free_device(s);
  std::string err = s->flatten();
  if (!err.empty()) return fail(RTMI_ERR_INVALID, err);
  HIP_TRY(hipGetDevice(&s->device));
  // image textures
  s->tex_recs.clear();
  for (const HostTex &t : s->texs) {
    if (!t.image) continue;
    void *d = nullptr;
    size_t pitch = 0;
    HIP_TRY(hipMallocPitch(&d, &pitch, (size_t)t.w * 4, (size_t)t.h));
    s->dev_allocs.push_back(d);
    HIP_TRY(hipMemcpy2D(d, pitch, t.rgba.data(), (size_t)t.w * 4, (size_t)t.w * 4, (size_t)t.h,
                        hipMemcpyHostToDevice));
    TexRec r{};
    r.rgba = reinterpret_cast<const uint8_t *>(d);
    r.height = t.h, r.width = t.w, r.pitch = (int64_t)pitch;
    s->tex_recs.push_back(r);
  }
  SceneDev d{};
  int rc;
  if ((rc = upload(s, s->runs, &d.runs))) return rc;
  if ((rc = upload(s, s->spheres, &d.spheres))) return rc;
  if ((rc = upload(s, s->tris, &d.tris))) return rc;
  if ((rc = upload(s, s->pair_boxes, &d.pair_boxes))) return rc;
  if ((rc = upload(s, s->pair_pts, &d.pair_pts))) return rc;
  if ((rc = upload(s, s->tri_nrm, &d.tri_nrm))) return rc;
  if ((rc = upload(s, s->sph_groups, &d.sph_groups))) return rc;
  if ((rc = upload(s, s->sph_members, &d.sph_members))) return rc;
  d.sph_mag = s->sph_mag;
  d.n_sph_groups = (int)s->sph_groups.size();
  if ((rc = upload(s, s->bvh_recs, &d.bvhs))) return rc;
  if ((rc = upload(s, s->nodes, &d.nodes))) return rc;
  if ((rc = upload(s, s->qnodes, &d.qnodes))) return rc;
  if ((rc = upload(s, s->leaf_paths, &d.leaf_paths))) return rc;
  if ((rc = upload(s, s->tops, &d.tops))) return rc;
  if ((rc = upload(s, s->faces, &d.faces))) return rc;
  if ((rc = upload(s, s->face_uv, &d.face_uv))) return rc;
#ifdef RTMI_CHECK_MARGINS
  if ((rc = upload(s, s->face_of_orig, &d.face_of_orig))) return rc;
#endif
  if ((rc = upload(s, s->mat_recs, &d.mats))) return rc;
  if ((rc = upload(s, s->tex_recs, &d.texs))) return rc;
  d.n_runs = (int)s->runs.size() - 4;  // without the padding records
  d.n_pairs = (int)s->pair_pts.size();
  d.list_mag = s->list_mag;
  d.n_mats = (int)s->mat_recs.size();
  d.n_nodes = (int)s->nodes.size();
  d.n_leaf_paths = (int)s->leaf_paths.size();
  d.sub_reserve = s->sub_depth > 0 ? 3 * s->sub_depth + 3 + kMeshFaceSlack : 0;
  d.det_safe = 1;
  for (const HotTri &t : s->tris) {
    const double a = std::sqrt((double)t.e1[0] * t.e1[0] + (double)t.e1[1] * t.e1[1] + (double)t.e1[2] * t.e1[2]);
    const double b = std::sqrt((double)t.e2[0] * t.e2[0] + (double)t.e2[1] * t.e2[1] + (double)t.e2[2] * t.e2[2]);
    if (!(a * b <= 0x1p120)) d.det_safe = 0;
  }
  d.unsigned_colours = 1;
  for (const MatRec &m : s->mat_recs) {
    const float c[3] = {m.r, m.g, m.b};
    for (float x : c) {
      uint32_t bits;
      memcpy(&bits, &x, sizeof(bits));
      if (bits >> 31) d.unsigned_colours = 0;
    }
  }
  d.cam = s->cam;
  s->dev = d;
  void *c = nullptr;
  // (behind the counters: the two kernel-argument blocks of a render without caller-owned scratch -- probe pass, real pass)
  HIP_TRY(hipMalloc(&c, RTMI_COUNTER_WORDS * sizeof(unsigned long long) + 2 * render_params_bytes()));
  s->dev_allocs.push_back(c);
  HIP_TRY(hipMemset(c, 0, RTMI_COUNTER_WORDS * sizeof(unsigned long long)));
  s->d_counters = reinterpret_cast<unsigned long long *>(c);
  HIP_TRY(hipDeviceSynchronize());
  s->committed = true;
  return RTMI_OK;
}

int rtmi_scene_stats(const rtmi_scene *sp, int64_t out[8]) {
  if (!sp || !out) return fail(RTMI_ERR_INVALID, "null argument");
  Scene tmp = *S(sp);  // flatten a copy so an uncommitted scene can be inspected
  tmp.dev_allocs.clear();
  std::string err = tmp.flatten();
  if (!err.empty()) return fail(RTMI_ERR_INVALID, err);
  out[0] = (int64_t)tmp.list_counts[0];
  out[1] = (int64_t)tmp.n_spheres;
  out[2] = (int64_t)tmp.n_pgrams;
  out[3] = (int64_t)tmp.n_triangles;
  out[4] = (int64_t)tmp.faces.size();
  out[5] = (int64_t)tmp.nodes.size();
  out[6] = (int64_t)tmp.mat_recs.size();
  out[7] = (int64_t)tmp.texs.size();
  return RTMI_OK;
}

int64_t rtmi_scene_sliver_faces(const rtmi_scene *sp) {
  if (!sp) return fail(RTMI_ERR_INVALID, "null scene");
  Scene tmp = *S(sp);
  tmp.dev_allocs.clear();
  std::string err = tmp.flatten();
  if (!err.empty()) return fail(RTMI_ERR_INVALID, err);
  return tmp.sliver_faces;
}

int64_t rtmi_scene_bytes_per_ray(const rtmi_scene *sp) {
  if (!sp) return fail(RTMI_ERR_INVALID, "null scene");
  Scene tmp = *S(sp);
  tmp.dev_allocs.clear();
  std::string err = tmp.flatten();
  if (!err.empty()) return fail(RTMI_ERR_INVALID, err);
  return tmp.bytes_per_ray;
}

// ------------------------------------------------------------------ frame
int64_t rtmi_frame_work_items(const rtmi_frame *f) {
  FrameDev d;
  if (!make_frame(f, &d)) return fail(RTMI_ERR_INVALID, frame_why("bad frame"));
  return d.items;
}
int64_t rtmi_frame_pixel_of(const rtmi_frame *f, int64_t q) {
  FrameDev d;
  if (!make_frame(f, &d) || q < 0 || q >= d.items) return -1;
  return frame_pixel_of(d, d.rank, q);
}
int rtmi_frame_pixel_map(const rtmi_frame *f, int64_t *out) {
  FrameDev d;
  if (!make_frame(f, &d) || !out) return fail(RTMI_ERR_INVALID, frame_why("bad frame"));
  for (int64_t q = 0; q < d.items; q++) out[q] = frame_pixel_of(d, d.rank, q);
  return RTMI_OK;
}
size_t rtmi_states_bytes(const rtmi_frame *f) {
  FrameDev d;
  if (!make_frame(f, &d)) return 0;
  return (size_t)d.items * RTMI_STATE_WORDS * sizeof(uint32_t);
}
size_t rtmi_tiles_bytes(const rtmi_frame *f) {
  FrameDev d;
  if (!make_frame(f, &d)) return 0;
  return (size_t)d.items * 3 * sizeof(float);
}

// ------------------------------------------------------------------ RNG
int rtmi_rng_init(uint64_t seed, const rtmi_frame *f, void *d_states, void *stream) {
  FrameDev d;
  if (!make_frame(f, &d) || !d_states) return fail(RTMI_ERR_INVALID, frame_why("bad rng_init arguments"));
  if (rtmi_device_count() <= 0) return fail(RTMI_ERR_NO_DEVICE, "no HIP device: librtmi has no CPU fallback");
  uint32_t *jump = nullptr;
  int rc = device_jump(&jump);
  if (rc) return rc;
  HIP_TRY(launch_rng_init(seed, d, jump, reinterpret_cast<uint32_t *>(d_states), (hipStream_t)stream));
  return RTMI_OK;
}
int rtmi_rng_host_state(uint64_t seed, uint64_t subsequence, uint32_t state[RTMI_STATE_WORDS]) {
  if (!state) return fail(RTMI_ERR_INVALID, "null state");
  Rng r = host_rng_init(seed, subsequence);
  state[0] = r.d, state[1] = r.v0, state[2] = r.v1, state[3] = r.v2, state[4] = r.v3, state[5] = r.v4;
  return RTMI_OK;
}
float rtmi_rng_host_random_float(float mn, float mx, uint32_t state[RTMI_STATE_WORDS]) {
  Rng r{state[0], state[1], state[2], state[3], state[4], state[5]};
  float x = rng_range(mn, mx, r);
  state[0] = r.d, state[1] = r.v0, state[2] = r.v1, state[3] = r.v2, state[4] = r.v3, state[5] = r.v4;
  return x;
}
int rtmi_rng_set_state(const rtmi_frame *f, void *d_states, int64_t q, const uint32_t state[RTMI_STATE_WORDS],
                       void *stream) {
  FrameDev d;
  if (!make_frame(f, &d) || !d_states || !state || q < 0 || q >= d.items)
    return fail(RTMI_ERR_INVALID, "bad rng_set_state arguments");
  uint32_t *base = reinterpret_cast<uint32_t *>(d_states);
  for (int w = 0; w < RTMI_STATE_WORDS; w++)
    HIP_TRY(hipMemcpyAsync(base + (size_t)w * d.items + q, &state[w], sizeof(uint32_t), hipMemcpyHostToDevice,
                           (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return RTMI_OK;
}
int rtmi_rng_get_state(const rtmi_frame *f, const void *d_states, int64_t q, uint32_t state[RTMI_STATE_WORDS],
                       void *stream) {
  FrameDev d;
  if (!make_frame(f, &d) || !d_states || !state || q < 0 || q >= d.items)
    return fail(RTMI_ERR_INVALID, "bad rng_get_state arguments");
  const uint32_t *base = reinterpret_cast<const uint32_t *>(d_states);
  for (int w = 0; w < RTMI_STATE_WORDS; w++)
    HIP_TRY(hipMemcpyAsync(&state[w], base + (size_t)w * d.items + q, sizeof(uint32_t), hipMemcpyDeviceToHost,
                           (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return RTMI_OK;
}

// ------------------------------------------------------------------ render
// Per-call scratch: [ counters: RTMI_COUNTER_WORDS x 8 B ][ probe RNG states ][ probe ray counts ][ tile costs ]
// [ tile order ][ 32 words of scheduler meta ][ head list: kHeadCap words ][ probe work counts ][ quarter costs ]
// [ quarters sorted ][ quarter order ][ 4 words ][ chain plan: per tile estimate of what follows, next tile, claim; per
// chain its first tile (kMaxChains words) ] rounded up to 256 bytes, then [ wave-priority table: kPrioTabBytes ]
// [ two kernel-argument blocks: probe pass, real pass ].  The counters come first so that rtmi_render_status can find
// them from the scratch pointer alone.
static constexpr size_t kCounterBytes = RTMI_COUNTER_WORDS * sizeof(unsigned long long);
static constexpr int kMaxChains = 1 << 15;  // planned chains: one per wave of the grid (8 waves x 4 SIMDs x 1024 CUs)
static size_t scratch_body_bytes(const FrameDev &d);
static size_t scratch_bytes_of(const FrameDev &d) {
  return scratch_body_bytes(d) + kPrioTabBytes + 2 * render_params_bytes();
}
static size_t scratch_body_bytes(const FrameDev &d) {
  const size_t n = (size_t)d.items, nt = (size_t)d.local_tiles;
  return (kCounterBytes + n * RTMI_STATE_WORDS * 4 + n * 4 + nt * 4 * 2 + 128 + (size_t)kHeadCap * 4 +
         n * 4 + nt * 4 * 15 + 16 + (size_t)kMaxChains * 4 + 255) & ~(size_t)255;  // + the probe's work counts, the quarter-tile costs, their sorted
                                                     // list, the order, the chain plan; then the wave-priority table and
                                                     // the two kernel-argument blocks (scratch_bytes_of)
}
size_t rtmi_render_scratch_bytes(const rtmi_frame *f) {
  FrameDev d;
  if (!make_frame(f, &d)) return 0;
  return scratch_bytes_of(d);
}

int rtmi_render(const rtmi_scene *sp, const rtmi_frame *f, void *d_states, float *d_tiles, uint32_t *d_ray_counts,
                void *stream) {
  return rtmi_render_ex(sp, f, nullptr, d_states, d_tiles, d_ray_counts, stream);
}

// rtmi_render_opts over the process defaults -> the tuning of ONE call.
static int resolve_opts(const rtmi_render_opts *opts, RenderTuning *tune, void **scratch, size_t *scratch_bytes) {
  *tune = default_tuning();
  *scratch = nullptr, *scratch_bytes = 0;
  if (!opts) return RTMI_OK;
  if (opts->size != (int32_t)sizeof(rtmi_render_opts)) return fail(RTMI_ERR_INVALID, "rtmi_render_opts.size does not match this library");
  if (opts->schedule > 2 || opts->blocks_per_cu < 0 || opts->threads_per_block < 0 || (opts->threads_per_block % 64) != 0 ||
      opts->threads_per_block > 512 || (opts->sparse_stride != 0 && !valid_stride(opts->sparse_stride)) || opts->exclusive > 1 ||
      opts->outlier_x10 < 0 || opts->probe_spp < 0 || opts->probe_spp > 64 || opts->plan > 2 || opts->wave_priority > 4096 ||
      (opts->wave_priority > 0 && (opts->wave_priority & (opts->wave_priority - 1)) != 0) || opts->lane_stride < 0 ||
      opts->lane_stride > 64 || (opts->lane_stride & (opts->lane_stride - 1)) != 0 || opts->cost_probe > 1 || opts->first_pass > 4096)
    return fail(RTMI_ERR_INVALID, "rtmi_render_opts field out of range");
  for (int i = 0; i < 3; i++)
    if (opts->head_pct[i] < 0 || opts->head_pct[i] > 100) return fail(RTMI_ERR_INVALID, "rtmi_render_opts.head_pct outside [0, 100]");
  if (opts->schedule >= 0) tune->schedule = opts->schedule;
  if (opts->blocks_per_cu > 0) tune->blocks_per_cu = opts->blocks_per_cu;
  if (opts->threads_per_block > 0) tune->threads = opts->threads_per_block;
  if (opts->sparse_stride > 0) tune->sparse_stride = opts->sparse_stride;
  if (opts->exclusive >= 0) tune->exclusive = opts->exclusive;
  if (opts->outlier_x10 > 0) tune->outlier_x10 = opts->outlier_x10;
  if (opts->probe_spp > 0) tune->probe_spp = opts->probe_spp;
  if (opts->plan >= 0) tune->plan = opts->plan;
  if (opts->wave_priority >= 0) tune->prio_every = opts->wave_priority;
  if (opts->lane_stride > 0) tune->lane_stride = opts->lane_stride;
  if (opts->promote_after >= 0) tune->promote = opts->promote_after;
  if (opts->cost_probe >= 0) tune->cost_probe = opts->cost_probe;
  if (opts->first_pass >= 0) tune->first_pass = opts->first_pass;
  for (int i = 0; i < 3; i++)
    if (opts->head_pct[i] > 0) tune->head_pct[i] = opts->head_pct[i];
  if (!(tune->head_pct[0] >= tune->head_pct[1] && tune->head_pct[1] >= tune->head_pct[2]))
    return fail(RTMI_ERR_INVALID, "rtmi_render_opts.head_pct must not increase from the heaviest class to the lightest");
  *scratch = opts->d_scratch, *scratch_bytes = opts->scratch_bytes;
  return RTMI_OK;
}

// Kernel variant and launch shape of a frame on the current device.
struct LaunchShape {
  uint32_t variant;
  int threads, per_cu, blocks, n_cu, lane_stride;
};
static constexpr int kMaxLaneStride = 16;
static int launch_shape(const Scene *s, const FrameDev &d, const RenderTuning &tune, LaunchShape *out) {
  int dev = 0;
  HIP_TRY(hipGetDevice(&dev));
  if (dev != s->device) return fail(RTMI_ERR_INVALID, "scene was committed on another device");
  int n_cu = 0;
  {  // compute-unit count, cached per device (hipGetDeviceProperties is slow)
    std::lock_guard<std::mutex> lk(g_mu);
    auto it = g_cus.find(dev);
    if (it == g_cus.end()) {
      hipDeviceProp_t prop;
      HIP_TRY(hipGetDeviceProperties(&prop, dev));
      it = g_cus.emplace(dev, prop.multiProcessorCount).first;
    }
    n_cu = it->second;
  }
  const uint32_t variant = pick_variant(s->features);
  int threads = tune.threads > 0 ? tune.threads : 256;
  if (threads > 256 && !(variant & F_BVH)) threads = 256;  // only the mesh kernels are built for larger workgroups
  if (tune.threads <= 0 && (variant & F_BVH)) {
    // mesh variants keep ~310 B of LDS per lane plus per-workgroup tables: when a deep id stack leaves
    // room for one 256-lane workgroup only, smaller workgroups keep more lanes resident
    int best = 0;
    for (int t = 256; t >= 64; t /= 2) {
      const int lanes = t * render_occupancy(variant, s->dev, d, t);
      if (lanes > best) best = lanes, threads = t;
    }
  }
  int per_cu = tune.blocks_per_cu > 0 ? tune.blocks_per_cu : render_occupancy(variant, s->dev, d, threads);
  if (per_cu <= 0) per_cu = 1;
  int64_t want = (d.items + threads - 1) / threads;
  int64_t cap = (int64_t)n_cu * per_cu;
  // A list frame smaller than the grid is spread thin (render_body.h: lane_stride): one pixel per 2 / 4 / ... lanes, as
  // far as the grid has room, when the closest-hit query shares its candidate tests between the lanes of a wave (the
  // culled list scan, the grouped sphere scan: a wave with a quarter of the rays then runs shorter iterations)
  int stride = 1;
  const bool shared_tests = ((variant & F_TRIS) && s->dev.n_pairs >= kCullMinPairs) || ((variant & F_SGROUP) && s->dev.n_sph_groups > 0);
  if (!(variant & F_BVH) && shared_tests) {
    if (tune.lane_stride > 0) stride = tune.lane_stride;
    else
      while (stride < kMaxLaneStride && want * stride * 2 <= cap) stride *= 2;
  }
  want *= stride;
  int blocks = (int)(want < cap ? want : cap);
  if (blocks < 1) blocks = 1;
  out->variant = variant, out->threads = threads, out->per_cu = per_cu, out->blocks = blocks, out->n_cu = n_cu;
  out->lane_stride = stride;
  return RTMI_OK;
}

// How a frame will be rendered: everything rtmi_render_ex decides before it launches anything (rtmi_render_mode reports
// it).  `tune` is the call's tuning, already carrying the launch shape's lane stride; prio_every may be lowered here.
struct RenderMode {
  bool scheduled, resume, may_plan, prio;
  int probe_spp;
};
static RenderMode decide_mode(const FrameDev &d, uint32_t variant, const LaunchShape &ls, RenderTuning &tune) {
  const int blocks = ls.blocks, threads = ls.threads;
  const int64_t resident = (int64_t)blocks * threads;
  // When is a list frame planned?  The plan wins where the queue cannot even things out (few tiles per wave) AND its
  // estimates are good enough (long pixels: many samples).  Measured, planned against queued, cornell depth 50: 1.33
  // tiles per wave (a 2048^2 frame over eight GPUs): 128 spp 20.9 / 21.0 ms, 512 spp 66 / 77, 4096 spp 480 / 590; 2.67
  // tiles per wave (1024^2): 128 spp 38.3 / 35.1, 256 spp 66.8 / 64.3, 512 spp 122.4 / 123.5, 1024 spp 237 / 241; 5.3
  // (half a 2048^2 x 4096 frame) 1763 / 1787; 6.4 (a C5 shard, 8192 spp) 2853 / 2917; 10.7: the same.  Spheres 1024^2 x
  // 64 spp, 3.2 tiles per wave: 21.3-22.6 / 20.4.
  const int64_t plan_tiles = d.local_tiles, plan_waves = (int64_t)blocks * (threads / 64);
  const bool plan_pays = (2 * plan_tiles <= 3 * plan_waves && d.spp >= 128) || (plan_tiles <= 3 * plan_waves && d.spp >= 512) ||
                         (plan_tiles <= 8 * plan_waves && d.spp >= 2048);
  const bool may_plan = tune.plan && tune.prio_every > 0 && d.spp >= 64 && !(variant & F_BVH) && ls.lane_stride == 1 &&
                        (tune.plan == 2 || plan_pays);
  // Samples of the scheduler's first look at the frame.  As a DISCARDED probe (first_pass = 0): two for the queue (its
  // order only has to be roughly longest-first: 2 / 4 / 8 / 16 spp gave 606 / 620 / 609 / 614 ms on a round-3 C4 shard),
  // 1 / 256 of the frame's samples, at most 16, for a plan, which is only as balanced as its estimates (C4 shard 2 / 8 /
  // 16 / 32 / 64: 495 / 486 / 484 / 482 / 484 ms, probe included).  rtmi_render_opts.probe_spp overrides either way.
  int probe_spp = tune.probe_spp > 0 ? tune.probe_spp : 2;
  if (tune.probe_spp <= 0 && may_plan) probe_spp = d.spp / 256 < 2 ? 2 : d.spp / 256 > 16 ? 16 : d.spp / 256;
  const bool many_tiles = (int64_t)d.local_tiles * 64 > resident;
  // The probe is the frame's own first samples (tune.first_pass): samples [0, s1) of every pixel are rendered into the
  // caller's buffers from the queue in image order, their ray counts order / plan the rest, and the second launch
  // resumes every pixel at sample s1 -- nothing is rendered twice, and s1 can be a sixteenth of the frame where a
  // discarded probe had to stay at a few samples (a 64-spp frame planned on 2 discarded samples: 21.9 ms; on 8: 20.5,
  // their cost included).  first_pass = 0 keeps the discarded probe on a scratch copy of the RNG states.
  const bool two_pass = tune.first_pass != 0;
  if (two_pass && tune.probe_spp <= 0) {
    // a frame that will be planned spends a sixteenth of its samples (at most 64) on the first pass: the plan is as good
    // as its estimates (C2: 238.9 -> 236.9 ms); everything else two -- the first pass runs from the plain queue, which
    // is the slower way to render a mesh frame (C3 with 32 first samples: 83 ms against 74) or a short one (spheres
    // 1024^2 x 64 spp: 2 / 4 / 8 first samples 20.3 / 20.7 / 21.5 ms; the discarded 2-spp probe: 20.9)
    const int div = tune.first_pass > 1 ? tune.first_pass : 16;
    probe_spp = !may_plan ? 2 : d.spp / div < 2 ? 2 : d.spp / div > 64 ? 64 : d.spp / div;
  }
  // (a first pass costs nothing but a launch, so short frames are scheduled too where it matters most: a mesh frame's
  // outlier pixels -- the reference's own bunny program, 1280 x 720 x 20 spp: 7.3 -> 4.9 ms -- from 8 samples per pixel
  // on; list frames from 32: at the reference's defaults, 100-200 spp, scheduled and unscheduled differ by +-4 %)
  const int min_spp = two_pass ? ((variant & F_BVH) ? 8 : 32) : 32 * probe_spp;
  const bool scheduled = tune.schedule == 2 || (tune.schedule == 1 && many_tiles && d.spp >= min_spp && d.spp >= 2 * probe_spp);
  // wave priorities (render_body.h: wave_priority_update) pay for themselves when a wave lives for many updates
  // ... from eight samples per pixel on; a short frame's waves live for tens of iterations, so they look every four
  // (C1, spheres 256^2 x 16 spp: 2.07 -> 1.86 ms; every 16: 1.89, every 2: 1.94, every iteration: 2.07)
  const bool prio = tune.prio_every > 0 && d.spp >= 8;
  if (prio && d.spp < 64 && tune.prio_every > 4) tune.prio_every = 4;
  RenderMode m;
  m.scheduled = scheduled, m.may_plan = may_plan, m.prio = prio, m.probe_spp = probe_spp;
  m.resume = scheduled && two_pass && probe_spp < d.spp;
  return m;
}

int rtmi_render_launch_shape(const rtmi_scene *sp, const rtmi_frame *f, const rtmi_render_opts *opts, int32_t out[4]) {
  if (!sp || !out) return fail(RTMI_ERR_INVALID, "null argument");
  const Scene *s = S(sp);
  if (!s->committed) return fail(RTMI_ERR_INVALID, "scene not committed");
  FrameDev d;
  if (!make_frame(f, &d)) return fail(RTMI_ERR_INVALID, frame_why("bad frame"));
  RenderTuning tune;
  void *scratch;
  size_t scratch_bytes;
  int rc = resolve_opts(opts, &tune, &scratch, &scratch_bytes);
  if (rc) return rc;
  LaunchShape ls;
  if ((rc = launch_shape(s, d, tune, &ls))) return rc;
  out[0] = ls.blocks, out[1] = ls.threads, out[2] = ls.per_cu, out[3] = ls.n_cu;
  return RTMI_OK;
}

int rtmi_render_mode(const rtmi_scene *sp, const rtmi_frame *f, const rtmi_render_opts *opts, int32_t out[8]) {
  if (!sp || !out) return fail(RTMI_ERR_INVALID, "null argument");
  const Scene *s = S(sp);
  if (!s->committed) return fail(RTMI_ERR_INVALID, "scene not committed");
  FrameDev d;
  if (!make_frame(f, &d)) return fail(RTMI_ERR_INVALID, frame_why("bad frame"));
  RenderTuning tune;
  void *scratch;
  size_t scratch_bytes;
  int rc = resolve_opts(opts, &tune, &scratch, &scratch_bytes);
  if (rc) return rc;
  LaunchShape ls;
  if ((rc = launch_shape(s, d, tune, &ls))) return rc;
  tune.lane_stride = ls.lane_stride;
  const RenderMode m = decide_mode(d, ls.variant, ls, tune);
  const int waves = ls.blocks * (ls.threads / 64);
  out[0] = m.scheduled ? 1 : 0;
  out[1] = m.scheduled ? m.probe_spp : 0;
  out[2] = m.resume ? 1 : 0;
  const int simds = ls.n_cu * 4 < waves ? ls.n_cu * 4 : waves, rounds = simds > 0 ? (waves + simds - 1) / simds : 0;
  out[3] = m.scheduled && m.may_plan && m.prio && simds * rounds <= kMaxChains ? 1 : 0;
  out[4] = m.prio ? tune.prio_every : 0;
  out[5] = ls.lane_stride;
  out[6] = waves;
  out[7] = d.local_tiles;
  return RTMI_OK;
}

int rtmi_render_ex(const rtmi_scene *sp, const rtmi_frame *f, const rtmi_render_opts *opts, void *d_states,
                   float *d_tiles, uint32_t *d_ray_counts, void *stream) {
  if (!sp || !d_states || !d_tiles) return fail(RTMI_ERR_INVALID, "null argument");
  RenderTuning tune;
  void *user_scratch = nullptr;
  size_t user_scratch_bytes = 0;
  int rc = resolve_opts(opts, &tune, &user_scratch, &user_scratch_bytes);
  if (rc) return rc;
  const Scene *s = S(sp);
  if (!s->committed) return fail(RTMI_ERR_INVALID, "scene not committed");
  FrameDev d;
  if (!make_frame(f, &d)) return fail(RTMI_ERR_INVALID, frame_why("bad frame"));
  if (d.max_depth < 0 || d.max_depth > RTMI_MAX_DEPTH) return fail(RTMI_ERR_DEPTH, "max_depth outside [0, 64]");
  // a pixel's closest-hit queries (at most max_depth + 1 per sample, ray_tracing.cu:22) are counted in 31 bits of its
  // ray_counts word (bit 31: the scheduler's mark) and of the trace kernel's register
  if ((int64_t)d.spp * (d.max_depth + 1) > (int64_t)RTMI_MAX_PIXEL_QUERIES)
    return fail(RTMI_ERR_INVALID, "spp x (max_depth + 1) above 2^31 - 1 (RTMI_MAX_PIXEL_QUERIES): a pixel's closest-hit queries are counted in 31 bits");
  LaunchShape ls;
  if ((rc = launch_shape(s, d, tune, &ls))) return rc;
  const uint32_t variant = ls.variant;
  const int threads = ls.threads, blocks = ls.blocks;
  tune.lane_stride = ls.lane_stride;
  hipStream_t st = (hipStream_t)stream;
  const size_t need = scratch_bytes_of(d);
  if (user_scratch && user_scratch_bytes < need)
    return fail(RTMI_ERR_INVALID, "rtmi_render_opts.scratch_bytes < rtmi_render_scratch_bytes(frame)");
  // Every piece of device state of this call -- queue cursors, ray total, abandoned-search flag, the scheduler's
  // buffers -- lives in the caller's scratch when one is given: renders of one scene on several streams (or as N
  // shards on one device) then share nothing but the read-only scene.  Without one the scene's own (a cache, not
  // scene state) is used, which ties renders of this scene to one at a time.
  unsigned long long *counters = user_scratch ? reinterpret_cast<unsigned long long *>(user_scratch) : s->d_counters;
  // the kernels' argument blocks (kernels.hip: RenderParams): probe pass, real pass
  char *params = user_scratch ? reinterpret_cast<char *>(user_scratch) + scratch_body_bytes(d) + kPrioTabBytes
                              : reinterpret_cast<char *>(s->d_counters) + kCounterBytes;
  // Longest-first tile order, first pass, plan, priorities: decide_mode
  SchedPlan plan;
  const RenderMode mode = decide_mode(d, variant, ls, tune);
  const bool scheduled = mode.scheduled, may_plan = mode.may_plan, prio = mode.prio;
  const int probe_spp = mode.probe_spp;
  void *scratch = user_scratch;
  if (!scratch && (scheduled || prio)) {
    Scene *ms = const_cast<Scene *>(s);
    std::lock_guard<std::mutex> lk(g_mu);  // (re)allocation only
    if (ms->sched_bytes < need) {
      if (ms->d_sched) (void)hipFree(ms->d_sched);
      ms->d_sched = nullptr, ms->sched_bytes = 0;
      HIP_TRY(hipMalloc(&ms->d_sched, need));
      ms->sched_bytes = need;
    }
    scratch = ms->d_sched;
  }
  FrameDev first = d;               // the launch that finishes the frame: all of it, or what a first pass left
  uint32_t *ray_buf = d_ray_counts;  // (a resumed pixel reads its count back: scratch when the caller wants none)
  if (prio) {
    plan.prio_tab = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(scratch) + scratch_body_bytes(d));
    HIP_TRY(hipMemsetAsync(plan.prio_tab, 0, kPrioTabBytes, st));
  }
  if (scheduled) {
    const size_t n = (size_t)d.items, nt = (size_t)d.local_tiles;
    uint32_t *p_states = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(scratch) + kCounterBytes);
    uint32_t *p_rays = p_states + n * RTMI_STATE_WORDS;
    uint32_t *p_cost = p_rays + n;
    uint32_t *p_order = p_cost + nt;
    uint32_t *p_meta = p_order + nt;  // 16 + 16 words (launch_tile_order); 450 * nt words in: 8-byte aligned
    uint32_t *p_work = p_meta + 32 + kHeadCap;
    uint32_t *p_qcost = p_work + n, *p_qsorted = p_qcost + 4 * nt, *p_qmap = p_qsorted + 4 * nt, *p_qmax = p_qmap + 4 * nt;
    uint32_t *p_fut = p_qmax + 4;
    int32_t *p_next = reinterpret_cast<int32_t *>(p_fut + nt);
    uint32_t *p_claims = reinterpret_cast<uint32_t *>(p_next + nt);
    int32_t *p_first = reinterpret_cast<int32_t *>(p_claims + nt);  // kMaxChains words
    // mesh frames (binary32 t): the probe also books the lane-steps of its mesh searches on the pixels they serve
    const bool by_cost = tune.cost_probe && (variant & F_BVH) && !(variant & F_SPHERE);
    // first pass: the frame's own samples [0, probe_spp) into the caller's buffers, or a discarded probe on copies
    const bool resume = mode.resume;
    uint32_t *first_states = resume ? reinterpret_cast<uint32_t *>(d_states) : p_states;
    uint32_t *first_rays = resume && d_ray_counts ? d_ray_counts : p_rays;
    if (!resume) HIP_TRY(hipMemcpyAsync(p_states, d_states, n * RTMI_STATE_WORDS * 4, hipMemcpyDeviceToDevice, st));
    FrameDev probe = d;
    if (resume) {
      probe.k_end = probe_spp;
      first.k_begin = probe_spp, ray_buf = first_rays;
    } else {
      probe.spp = probe_spp, probe.k_end = probe_spp;
    }
    HIP_TRY(hipMemsetAsync(counters, 0, kCounterBytes, st));
    // (a discarded probe writes its radiance into d_tiles, which the real pass overwrites)
    SchedPlan probe_plan;
    if (by_cost) {
      HIP_TRY(hipMemsetAsync(p_work, 0, n * 4, st));
      probe_plan.visit_counts = p_work;
    }
    // (a first pass of the frame's own samples is a frame of s1 samples per pixel: from 32 on it has wave priorities too)
    const bool first_prio = prio && resume && probe_spp >= 32 && env_int("RTMI_FIRST_PRIO", 1) != 0;
    if (first_prio) probe_plan.prio_tab = plan.prio_tab;
    HIP_TRY(launch_render(variant, s->dev, probe, first_states, d_tiles, first_rays, counters, probe_plan, true, blocks,
                          threads, tune, params, st));
    if (first_prio) HIP_TRY(hipMemsetAsync(plan.prio_tab, 0, kPrioTabBytes, st));
    if (resume) {
      // The scheduler's kernels read the first pass's ray counts and MARK the head's pixels in them (bit 31), and the
      // marks must outlive the pixels' final counts, which the second launch writes into the same words as it goes: they
      // work on a copy (in the region a discarded probe's RNG states would have used).
      HIP_TRY(hipMemcpyAsync(p_states, first_rays, n * 4, hipMemcpyDeviceToDevice, st));
      p_rays = p_states;
    }
    const uint32_t sparse_cap = (uint32_t)(((int64_t)blocks * threads / tune.sparse_stride) / 64 * 64);
    // the head of a mesh frame's queue: pixels in weight classes (the default), or -- when the call names a
    // sparse stride, or RTMI_HEAD_CLASSES=0 -- the outlier tiles at one pixel per that many lanes
    static const bool head_classes = env_int("RTMI_HEAD_CLASSES", 1) != 0;
    const bool by_pixels = head_classes && !(opts && opts->sparse_stride > 0);
    uint32_t *p_head = (variant & F_BVH) && by_pixels ? p_meta + 32 : nullptr;
    HIP_TRY(launch_tile_order(p_rays, d.local_tiles, p_cost, p_meta, p_order, p_head, sparse_cap, blocks * (threads / 64),
                              tune.outlier_x10, tune.head_pct, st));
    // (the head's marks in p_rays are bit 31: quarter_cost_kernel masks them off)
    HIP_TRY(launch_quarter_order(p_order, by_cost && by_pixels ? p_work : nullptr, p_rays, d.local_tiles, p_qcost, p_qsorted, p_qmax,
                                 p_qmap, st));
    plan.tile_order = p_qmap;
    plan.sparse_items = p_meta + 1;
    plan.head_list = p_head;
    plan.probe_marks = p_head ? p_rays : nullptr;
    plan.probe_spp = probe_spp;
    plan.tile_cost = p_cost;
    // list frames: planned chains instead of the queue (kernels.h: launch_chain_plan)
    const int grid_waves = blocks * (threads / 64);
    if (may_plan && prio) {
      const int simds = ls.n_cu * 4 < grid_waves ? ls.n_cu * 4 : grid_waves;  // (four SIMDs per compute unit)
      const int rounds = (grid_waves + simds - 1) / simds;
      if (simds * rounds <= kMaxChains) {
        HIP_TRY(launch_chain_plan(p_order, p_cost, d.local_tiles, simds, rounds, d.spp, probe_spp, p_first, p_next, p_fut, st));
        HIP_TRY(hipMemsetAsync(p_claims, 0, nt * 4, st));
        plan.chain_next = p_next, plan.chain_fut = p_fut, plan.chain_first = p_first, plan.claims = p_claims;
        plan.plan_simds = simds, plan.plan_rounds = rounds;
        plan.tile_order = p_order;  // (per tile in this mode: the take-over's order)
      }
    }
  }
  HIP_TRY(hipMemsetAsync(counters, 0, kCounterBytes, st));
  HIP_TRY(launch_render(variant, s->dev, first, reinterpret_cast<uint32_t *>(d_states), d_tiles, ray_buf, counters,
                        plan, false, blocks, threads, tune, params + render_params_bytes(), st));
  return RTMI_OK;
}

int rtmi_render_status(const rtmi_scene *sp, const void *d_scratch, uint64_t *out_rays, void *stream) {
  if (!sp) return fail(RTMI_ERR_INVALID, "null argument");
  const Scene *s = S(sp);
  if (!s->committed) return fail(RTMI_ERR_INVALID, "scene not committed");
  const unsigned long long *counters = d_scratch ? reinterpret_cast<const unsigned long long *>(d_scratch) : s->d_counters;
  unsigned long long v[2] = {0, 0};  // [0] rays, [1] abandoned mesh searches (must be 0)
  HIP_TRY(hipMemcpyAsync(v, counters + 1, sizeof(v), hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  if (out_rays) *out_rays = v[0];
  if (v[1] != 0) return fail(RTMI_ERR_INTERNAL, "mesh search stack overflow: the frame is incomplete");
  return RTMI_OK;
}

int rtmi_last_ray_total(const rtmi_scene *sp, uint64_t *out_rays, void *stream) {
  if (!out_rays) return fail(RTMI_ERR_INVALID, "null argument");
  return rtmi_render_status(sp, nullptr, out_rays, stream);
}

// Diagnostic: the raw counter words of the most recent render (RTMI_STATS builds fill words 4..32).
int rtmi_debug_counters(const rtmi_scene *sp, unsigned long long out[RTMI_COUNTER_WORDS], void *stream) {
  return rtmi_debug_counters_ex(sp, nullptr, out, stream);
}
int rtmi_debug_counters_ex(const rtmi_scene *sp, const void *d_scratch, unsigned long long out[RTMI_COUNTER_WORDS],
                           void *stream) {
  if (!sp || !out) return fail(RTMI_ERR_INVALID, "null argument");
  const Scene *s = S(sp);
  if (!s->committed) return fail(RTMI_ERR_INVALID, "scene not committed");
  const unsigned long long *counters = d_scratch ? reinterpret_cast<const unsigned long long *>(d_scratch) : s->d_counters;
  HIP_TRY(hipMemcpyAsync(out, counters, kCounterBytes, hipMemcpyDeviceToHost, (hipStream_t)stream));
  HIP_TRY(hipStreamSynchronize((hipStream_t)stream));
  return RTMI_OK;
}

#ifdef RTMI_STATS
// diagnostic builds only (not declared in rtmi.h): per-wave cycle records of the last render
int rtmi_debug_wave_stats(unsigned long long *out, size_t bytes) {
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(rtmi::copy_wave_stats(out, bytes));
  return RTMI_OK;
}
#endif

// ------------------------------------------------------------------ exchange
// RCCL is not a link-time dependency of this library: the caller owns the communicator, so its RCCL is already
// in the process (a second copy must not be pulled in next to e.g. the one PyTorch bundles).  The four entry points
// are taken from the process image, or from librccl.so.1 when nothing has loaded it yet.
extern "C++" {
namespace {
struct Rccl {
  decltype(&ncclGroupStart) group_start = nullptr;
  decltype(&ncclGroupEnd) group_end = nullptr;
  decltype(&ncclSend) send = nullptr;
  decltype(&ncclRecv) recv = nullptr;
  decltype(&ncclReduce) reduce = nullptr;
  decltype(&ncclGetErrorString) err = nullptr;
  decltype(&ncclCommCount) count = nullptr;
  bool ok = false;
};
const Rccl &rccl() {
  static const Rccl r = [] {
    Rccl x;
    // "already in the process" and "the handle to look symbols up in" are two things: RTLD_DEFAULT is a null handle
    // on glibc, so a found symbol must not be mistaken for a failed dlopen (which once pulled a SECOND copy of RCCL
    // in next to the one the caller's communicator came from)
    const bool in_process = dlsym(RTLD_DEFAULT, "ncclSend") != nullptr;
    void *h = RTLD_DEFAULT;
    if (!in_process) {
      h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
      if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
      if (!h) return x;
    }
    x.group_start = reinterpret_cast<decltype(x.group_start)>(dlsym(h, "ncclGroupStart"));
    x.group_end = reinterpret_cast<decltype(x.group_end)>(dlsym(h, "ncclGroupEnd"));
    x.send = reinterpret_cast<decltype(x.send)>(dlsym(h, "ncclSend"));
    x.recv = reinterpret_cast<decltype(x.recv)>(dlsym(h, "ncclRecv"));
    x.reduce = reinterpret_cast<decltype(x.reduce)>(dlsym(h, "ncclReduce"));
    x.err = reinterpret_cast<decltype(x.err)>(dlsym(h, "ncclGetErrorString"));
    x.count = reinterpret_cast<decltype(x.count)>(dlsym(h, "ncclCommCount"));
    x.ok = x.group_start && x.group_end && x.send && x.recv && x.reduce;
    return x;
  }();
  return r;
}
int rccl_fail(const Rccl &R, ncclResult_t e, const char *what) {
  return fail(RTMI_ERR_HIP, std::string(what) + ": " + (R.err ? R.err(e) : "RCCL error"));
}
}  // namespace
}  // extern "C++"
#define RCCL_TRY(expr)                                        \
  do {                                                        \
    ncclResult_t e__ = (expr);                                \
    if (e__ != ncclSuccess) return rccl_fail(R, e__, #expr);  \
  } while (0)

int rtmi_gather(void *nccl_comm, const rtmi_frame *f, const float *d_tiles, float *d_all_tiles, int root, void *stream) {
  FrameDev d;
  if (!make_frame(f, &d) || !d_tiles || root < 0 || root >= d.world) return fail(RTMI_ERR_INVALID, frame_why("bad gather arguments"));
  if (d.rank == root && !d_all_tiles) return fail(RTMI_ERR_INVALID, "the root needs d_all_tiles");
  const size_t count = (size_t)d.items * 3;
  hipStream_t st = (hipStream_t)stream;
  if (d.rank == root && d_all_tiles + (size_t)root * count != d_tiles)
    HIP_TRY(hipMemcpyAsync(d_all_tiles + (size_t)root * count, d_tiles, count * sizeof(float), hipMemcpyDeviceToDevice, st));
  if (d.world == 1) return RTMI_OK;
  if (!nccl_comm) return fail(RTMI_ERR_INVALID, "world_size > 1 needs an RCCL communicator");
  const Rccl &R = rccl();
  if (!R.ok) return fail(RTMI_ERR_NO_DEVICE, "RCCL (librccl.so.1) is not available in this process");
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
  // xGMI is point to point and every peer has a direct link to the root: one grouped round of sends, no ring.
  // A group that was opened is always closed, also when a send / recv inside it fails: the communicator must not be
  // left mid-group for the caller's next collective.
  RCCL_TRY(R.group_start());
  ncclResult_t first = ncclSuccess;
  const char *what = "";
  if (d.rank == root) {
    for (int r = 0; r < d.world && first == ncclSuccess; r++)
      if (r != root) first = R.recv(d_all_tiles + (size_t)r * count, count, ncclFloat, r, comm, st), what = "ncclRecv";
  } else {
    first = R.send(d_tiles, count, ncclFloat, root, comm, st), what = "ncclSend";
  }
  const ncclResult_t closed = R.group_end();
  if (first != ncclSuccess) return rccl_fail(R, first, what);
  if (closed != ncclSuccess) return rccl_fail(R, closed, "ncclGroupEnd");
  return RTMI_OK;
}

int rtmi_reduce_sum(void *nccl_comm, const rtmi_frame *f, float *d_tiles, int root, void *stream) {
  FrameDev d;
  if (!make_frame(f, &d) || !d_tiles || root < 0) return fail(RTMI_ERR_INVALID, frame_why("bad reduce arguments"));
  // In the reference's sample split every rank renders the WHOLE frame (utils.cu:189,216-221), so the frame says nothing
  // about how many ranks there are: the communicator does.  NULL = this rank is the only one, its sum is the sum.
  if (!nccl_comm) {
    if (root != 0) return fail(RTMI_ERR_INVALID, "reduce without a communicator is a single rank: root must be 0");
    return RTMI_OK;
  }
  const Rccl &R = rccl();
  if (!R.ok) return fail(RTMI_ERR_NO_DEVICE, "RCCL (librccl.so.1) is not available in this process");
  ncclComm_t comm = reinterpret_cast<ncclComm_t>(nccl_comm);
  if (R.count) {
    int n = 0;
    RCCL_TRY(R.count(comm, &n));
    if (root >= n) return fail(RTMI_ERR_INVALID, "reduce root is not a rank of the communicator");
  }
  RCCL_TRY(R.reduce(d_tiles, d_tiles, (size_t)d.items * 3, ncclFloat, ncclSum, root, comm, (hipStream_t)stream));
  return RTMI_OK;
}

int rtmi_untile(const rtmi_frame *f, const float *d_all_tiles, float *d_image, void *stream) {
  FrameDev d;
  if (!make_frame(f, &d) || !d_all_tiles || !d_image) return fail(RTMI_ERR_INVALID, frame_why("bad untile arguments"));
  HIP_TRY(launch_untile(d, d_all_tiles, d_image, (hipStream_t)stream));
  return RTMI_OK;
}
int rtmi_untile_u32(const rtmi_frame *f, const uint32_t *d_all, uint32_t *d_image, void *stream) {
  FrameDev d;
  if (!make_frame(f, &d) || !d_all || !d_image) return fail(RTMI_ERR_INVALID, frame_why("bad untile arguments"));
  HIP_TRY(launch_untile_u32(d, d_all, d_image, (hipStream_t)stream));
  return RTMI_OK;
}
int rtmi_post_process(float *d_image, int64_t n_pixels, int spp, void *stream) {
  if (!d_image || n_pixels < 0 || spp <= 0) return fail(RTMI_ERR_INVALID, "bad post_process arguments");
  HIP_TRY(launch_post(d_image, n_pixels * 3, spp, (hipStream_t)stream));
  return RTMI_OK;
}
int rtmi_get_workload(int rank, int world_size, int spp) {
  return spp / world_size + (int)(rank < (spp % world_size));  // utils.cu:111-113
}
int rtmi_selftest_arithmetic(unsigned long long *mismatches) {
  if (!mismatches) return fail(RTMI_ERR_INVALID, "mismatches == NULL");
  if (rtmi_device_count() <= 0) return fail(RTMI_ERR_NO_DEVICE, "no HIP device: librtmi has no CPU fallback");
  unsigned long long *d_bad = nullptr;
  HIP_TRY(hipMalloc(&d_bad, 8 * sizeof(unsigned long long)));
  hipError_t e = hipMemset(d_bad, 0, 8 * sizeof(unsigned long long));
  if (e == hipSuccess) e = launch_arithmetic_selftest(d_bad, nullptr);
  if (e == hipSuccess) e = hipMemcpy(mismatches, d_bad, 8 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  (void)hipFree(d_bad);
  HIP_TRY(e);
  return RTMI_OK;
}
int rtmi_set_schedule(int mode) {
  if (mode < 0 || mode > 2) return fail(RTMI_ERR_INVALID, "schedule mode must be 0 (image order), 1 (auto) or 2 (always longest-first)");
  (void)default_tuning();
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_tune.schedule = mode;
  return RTMI_OK;
}
int rtmi_set_launch(int blocks_per_cu, int threads_per_block) {
  if (blocks_per_cu < 0 || threads_per_block < 0 || (threads_per_block % 64) != 0 || threads_per_block > 512)
    return fail(RTMI_ERR_INVALID, "threads_per_block must be a multiple of 64, at most 512 (256 for scenes without meshes)");
  (void)default_tuning();
  std::lock_guard<std::mutex> lk(g_tune_mu);
  g_tune.blocks_per_cu = blocks_per_cu;
  g_tune.threads = threads_per_block;
  return RTMI_OK;
}

}  // extern "C"
