// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// CPU restatement (plain C++17, no dependencies) of the reference's per-pixel
// trace loop, op for op, with the reference's float/double mixing and quirks.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
// load this library, and only as the checker.
//
// PARITY STATUS: the reference is CUDA-only device code and cannot be built or
// run in this image (no nvcc / NVIDIA GPU / cuRAND / glm / glog / stb), and it
// ships no tests, golden vectors or fixtures.  This
// oracle is therefore pinned by (a) the analytic known-answers the reference
// source itself documents (sphere.cu:56-58 UV table, parallelogram.cu:19-21 UV
// diagram, utils.cu:111-113 workload split, sky.cu:9-14 gradient) and (b)
// rocRAND's independent XORWOW for the RNG recurrence + 2^67 jump.  cuRAND's
// seed salts, GLM, thrust::sort tie order and nvcc's FMA contraction choices
// remain "parity unpinned" (see DESIGN.md).
//
// Build: g++ -O2 -ffp-contract=off (every float op is one IEEE rounding).
//
// Every function cites the reference file:line it follows; paths are relative
// to /root/reference/ray-tracing-cuda/.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "vecmath.hpp"
#include "xorwow.hpp"

namespace orc {

// ---------------------------------------------------------------- counters
struct Counters {
  uint64_t rays = 0;         // world->Hit queries issued by Trace (ray_tracing.cu:22)
  uint64_t bvh_boxes = 0;    // AABB::Hit evaluations (bvh.cu:19)
  uint64_t bvh_faces = 0;    // Face::Hit evaluations inside BVH leaves (bvh.cuh:129)
};
static thread_local Counters *tl_counters = nullptr;

// ---------------------------------------------------------------- ray.cu:6-15
struct Ray {
  vec3 position_, direction_;
  Ray() {}
  Ray(vec3 p, vec3 d) : position_(p) { direction_ = normalize(d); }
  vec3 position() const { return position_; }
  vec3 direction() const { return direction_; }
};

struct Material;

// ---------------------------------------------------------------- hitable.cuh:13-23
struct HitRecord {
  double t = 0, u = 0, v = 0;
  vec3 normal;
  Material *material_ptr = nullptr;
};

struct Hitable {
  virtual ~Hitable() {}
  virtual bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) = 0;
};

// ---------------------------------------------------------------- textures/*.cu
struct Texture {
  virtual ~Texture() {}
  virtual vec3 Value(double u, double v, const vec3 &p) const = 0;
};

// textures/constant_texture.cu:11-14
struct ConstantTexture : Texture {
  vec3 color_;
  ConstantTexture() {}
  explicit ConstantTexture(vec3 c) : color_(c) {}
  vec3 Value(double, double, const vec3 &) const override { return color_; }
};

// textures/image_texture.cu:9-38.  tex2D<float4> with the all-zero
// cudaTextureDesc: point filter, wrap addressing, normalised coordinates,
// RGBA8 -> float by /255.  The texture unit's fixed-point coordinate
// arithmetic is not reproducible off NVIDIA hardware: "parity unpinned";
// restated as texel = floor(frac(coord) * extent).
struct ImageTexture : Texture {
  std::vector<uint8_t> rgba_;
  int h_ = 0, w_ = 0;
  vec3 Value(double u, double v, const vec3 &) const override {
    v = 1.0 - v;  // image_texture.cu:11
    float fu = (float)u, fv = (float)v;
    fu = fu - std::floor(fu);
    fv = fv - std::floor(fv);
    int ix = (int)std::floor(fu * (float)w_);
    int iy = (int)std::floor(fv * (float)h_);
    if (ix > w_ - 1) ix = w_ - 1;
    if (iy > h_ - 1) iy = h_ - 1;
    if (ix < 0) ix = 0;
    if (iy < 0) iy = 0;
    const uint8_t *px = &rgba_[((size_t)iy * w_ + ix) * 4];
    return vec3((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
  }
};

// ---------------------------------------------------------------- material.cuh:13-20, material.cu:3-6
struct Material {
  virtual ~Material() {}
  virtual bool Scatter(const Ray &ray, const HitRecord &record, Xorwow *state, vec3 *out_albedo,
                       Ray *out_ray) = 0;
  virtual vec3 Emit(double, double, const vec3 &) const { return vec3(0.0f); }
};

// lambertian.cu:9-43
struct Lambertian : Material {
  Texture *texture_ptr_ = nullptr;
  ConstantTexture color_;
  bool use_constant_tex_ = false;
  explicit Lambertian(Texture *t) : texture_ptr_(t), use_constant_tex_(false) {}
  explicit Lambertian(vec3 c) : color_(c), use_constant_tex_(true) {}
  const Texture *texture_ptr() const { return use_constant_tex_ ? &color_ : texture_ptr_; }

  // lambertian.cu:19-31: rejection-sample the unit ball, then normalise onto
  // the sphere.  l = (float)pow((double)(x*x+y*y+z*z), 0.5): the float sum is
  // widened, square-rooted in double and narrowed, which equals sqrtf(sum)
  // (double rounding of sqrt is innocuous for p=24 -> q=53).
  static vec3 SphericalRand(Xorwow *state) {
    float x, y, z, l;
    do {
      x = random_float(-1, 1, state);
      y = random_float(-1, 1, state);
      z = random_float(-1, 1, state);
      l = (float)std::sqrt((double)(x * x + y * y + z * z));
    } while (l > 1);
    x /= l;
    y /= l;
    z /= l;
    return vec3(x, y, z);
  }

  bool Scatter(const Ray &ray, const HitRecord &record, Xorwow *state, vec3 *out_albedo,
               Ray *out_ray) override {
    if (dot(ray.direction(), record.normal) >= 0) return false;  // lambertian.cu:36
    vec3 p = ray.position() + float(record.t) * ray.direction();
    vec3 albedo = texture_ptr()->Value(record.u, record.v, p);
    vec3 next_dir = normalize(SphericalRand(state) + record.normal);
    *out_albedo = albedo;
    *out_ray = Ray(p, next_dir);
    return true;
  }
};

// metal.cu:7-36
struct Metal : Material {
  vec3 albedo_;
  float fuzz_;
  Metal(vec3 a, float fuzz) : albedo_(a), fuzz_(fuzz < 1 ? fuzz : 1) {}

  static vec3 RandomInUnitSphere(Xorwow *state) {  // metal.cu:27-36 (not normalised)
    float x, y, z, l;
    do {
      x = random_float(-1, 1, state);
      y = random_float(-1, 1, state);
      z = random_float(-1, 1, state);
      l = (float)std::sqrt((double)(x * x + y * y + z * z));
    } while (l > 1);
    return vec3(x, y, z);
  }

  bool Scatter(const Ray &ray, const HitRecord &record, Xorwow *state, vec3 *out_albedo,
               Ray *out_ray) override {
    if (dot(ray.direction(), record.normal) >= 0) return false;  // metal.cu:15
    vec3 p = ray.position() + float(record.t) * ray.direction();
    *out_albedo = albedo_;
    vec3 reflected = reflect(ray.direction(), record.normal);
    if (fuzz_ > 0) {
      *out_ray = Ray(p, reflected + fuzz_ * RandomInUnitSphere(state));
    } else {
      *out_ray = Ray(p, reflected);
    }
    return true;
  }
};

// dielectric.cu:10-44 — refraction only; TIR (zero/NaN vector) ends the path.
struct Dielectric : Material {
  vec3 attenuation_;
  double refractive_index_;
  Dielectric(vec3 a, double n) : attenuation_(a), refractive_index_(n) {}

  static bool VectorIsValid(vec3 v) {  // dielectric.cu:19-24
    if (v[0] == 0 && v[1] == 0 && v[2] == 0) return false;
    for (int i = 0; i < 3; i++)
      if (std::isnan(v[i])) return false;
    return true;
  }

  bool Scatter(const Ray &ray, const HitRecord &record, Xorwow *, vec3 *out_albedo,
               Ray *out_ray) override {
    if (dot(ray.direction(), record.normal) >= 0) {  // from inner to outer
      vec3 p = ray.position() + float(record.t) * ray.direction();
      vec3 d = refract(ray.direction(), -record.normal, (float)refractive_index_ / 1.0f);
      if (!VectorIsValid(d)) return false;
      *out_albedo = attenuation_;
      *out_ray = Ray(p, d);
      return true;
    } else {  // from outer to inner
      vec3 p = ray.position() + float(record.t) * ray.direction();
      vec3 d = refract(ray.direction(), record.normal, 1.0f / (float)refractive_index_);
      if (!VectorIsValid(d)) return false;
      *out_albedo = attenuation_;
      *out_ray = Ray(p, d);
      return true;
    }
  }
};

// diffuse_light.cu:5-17
struct DiffuseLight : Material {
  Texture *texture_ptr_;
  explicit DiffuseLight(Texture *t) : texture_ptr_(t) {}
  bool Scatter(const Ray &, const HitRecord &, Xorwow *, vec3 *, Ray *) override { return false; }
  vec3 Emit(double u, double v, const vec3 &p) const override { return texture_ptr_->Value(u, v, p); }
};

// sky.cu:3-14
struct SkyMaterial : Material {
  bool Scatter(const Ray &, const HitRecord &, Xorwow *, vec3 *, Ray *) override { return false; }
  vec3 Emit(double, double, const vec3 &p) const override {
    vec3 dir = normalize(p);
    float t = (float)(0.5 * ((double)dir.y + 1.0));  // sky.cu:12: double arithmetic, narrowed
    return (1.0f - t) * vec3(1.0f, 1.0f, 1.0f) + t * vec3(0.5f, 0.7f, 1.0f);
  }
};

// ---------------------------------------------------------------- utils.cu:49-85
static bool TriangleHit(const vec3 p[3], const Ray &ray, double t_from, double t_to, double *out_t,
                        vec3 *out_normal, double *out_u, double *out_v) {
  double eps = 1e-7;
  vec3 v0v1 = p[1] - p[0];
  vec3 v0v2 = p[2] - p[0];
  vec3 pvec = cross(ray.direction(), v0v2);
  float det = dot(v0v1, pvec);
  if (std::fabs((double)det) < eps) return false;  // fabs(float)->double compare, utils.cu:60
  float invDet = 1 / det;
  vec3 tvec = ray.position() - p[0];
  float u = dot(tvec, pvec) * invDet;
  if (u < 0 || u > 1) return false;
  vec3 qvec = cross(tvec, v0v1);
  float v = dot(ray.direction(), qvec) * invDet;
  if (v < 0 || u + v > 1) return false;
  float t = dot(v0v2, qvec) * invDet;
  if (!(t_from <= t && t <= t_to)) return false;
  *out_t = t;
  vec3 n = normalize(cross(v0v1, v0v2));
  *out_normal = dot(ray.direction(), n) < 0 ? n : -n;
  *out_u = u;
  *out_v = v;
  return true;
}

// ---------------------------------------------------------------- hitable_list.cu:7-29
struct HitableList : Hitable {
  static constexpr int kMaxHitables = 1024;  // hitable_list.cuh:10
  std::vector<Hitable *> list_;
  bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) override {
    bool ok = false;
    HitRecord hit_record;  // persists across iterations, as in the reference (line 10)
    for (size_t i = 0; i < list_.size(); i++) {
      bool ret = list_[i]->Hit(ray, t_from, t_to, &hit_record);
      if (ret) {
        if (!ok) {
          *out = hit_record;
          t_to = std::min(t_to, hit_record.t);
          ok = true;
        } else if (hit_record.t < t_to) {
          t_to = hit_record.t;
          *out = hit_record;
        }
      }
    }
    return ok;
  }
  bool Append(Hitable *obj) {
    if ((int)list_.size() >= kMaxHitables) return false;
    list_.push_back(obj);
    return true;
  }
};

// ---------------------------------------------------------------- sphere.cu:11-64
struct Sphere : Hitable {
  double radius_;
  vec3 position_;
  Material *material_ptr_;
  Sphere(vec3 p, double r, Material *m) : radius_(r), position_(p), material_ptr_(m) {}

  // sphere.cu:52-64
  static void GetUV(const vec3 &p, double *u, double *v) {
    const float pi = 3.14159265358979323846264338327950288f;  // glm::pi<float>()
    float theta = std::acos(-p.y);
    float phi = std::atan2(-p.z, p.x) + pi;
    *u = phi / (2 * pi);
    *v = theta / pi;
  }

  bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) override {
    // sphere.cu:13,15: pow(float, int) resolves to CUDA's float powif
    // overload (quirk g13): the square is taken in float, then widened.
    float la = length(ray.direction());
    double a = (double)(la * la);
    double b = (double)(2 * dot(ray.direction(), ray.position() - position_));
    float lc = length(ray.position() - position_);
    double c = (double)(lc * lc) - radius_ * radius_;
    double discriminant = b * b - 4 * a * c;
    if (discriminant < 0) return false;
    double t = (-b - std::sqrt(discriminant)) / (2 * a);  // pow(x, 0.5) == sqrt(x)
    if (t_from <= t && t <= t_to) {
      HitRecord record;
      record.t = t;
      vec3 p = ray.position() + float(t) * ray.direction();
      record.normal = normalize(p - position_);
      record.material_ptr = material_ptr_;
      GetUV(record.normal, &record.u, &record.v);
      *out = record;
      return true;
    }
    t = (-b + std::sqrt(discriminant)) / (2 * a);
    if (t_from <= t && t <= t_to) {
      HitRecord record;
      record.t = t;
      vec3 p = ray.position() + float(t) * ray.direction();
      record.normal = normalize(p - position_);
      record.material_ptr = material_ptr_;
      GetUV(record.normal, &record.u, &record.v);
      *out = record;
      return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------- triangle.cu:6-19
struct Triangle : Hitable {
  vec3 p_[3];
  Material *material_ptr_;
  Triangle(const vec3 p[3], Material *m) : material_ptr_(m) {
    for (int i = 0; i < 3; i++) p_[i] = p[i];
  }
  bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) override {
    if (TriangleHit(p_, ray, t_from, t_to, &out->t, &out->normal, &out->u, &out->v)) {
      out->material_ptr = material_ptr_;
      return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------- parallelogram.cu:10-44
struct Parallelogram : Hitable {
  vec3 p_[4];
  Material *material_ptr_;
  Parallelogram(const vec3 p[3], Material *m) {
    for (int i = 0; i <= 2; i++) p_[i] = p[i];
    p_[3] = p[1] + p[2] - p[0];
    material_ptr_ = m;
  }
  bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) override {
    HitRecord record;
    record.material_ptr = material_ptr_;
    double u, v;
    if (TriangleHit(p_, ray, t_from, t_to, &record.t, &record.normal, &u, &v)) {
      vec2 uv = vec2(0, 1) * (float)(1 - u - v) + vec2(1, 1) * (float)u + vec2(0, 0) * (float)v;
      record.u = uv.x;
      record.v = uv.y;
      *out = record;
      return true;
    }
    if (TriangleHit(p_ + 1, ray, t_from, t_to, &record.t, &record.normal, &u, &v)) {
      vec2 uv = vec2(1, 1) * (float)(1 - u - v) + vec2(0, 0) * (float)u + vec2(1, 0) * (float)v;
      record.u = uv.x;
      record.v = uv.y;
      *out = record;
      return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------- parallelepiped.cu:8-55
struct Parallelepiped : Hitable {
  HitableList list_;
  std::vector<std::unique_ptr<Parallelogram>> owned_;

  void AddCorner(const vec3 p[4], Material *m) {  // parallelepiped.cu:25-32
    for (int i = 1; i <= 3; i++) {
      int x = i, y = i + 1 == 4 ? 1 : x + 1;
      vec3 arr[3] = {p[0], p[x], p[y]};
      owned_.emplace_back(new Parallelogram(arr, m));
      list_.Append(owned_.back().get());
    }
  }
  // parallelepiped.cu:8-18
  Parallelepiped(const vec3 p[4], Material *m) {
    auto fourth = [](vec3 a, vec3 b, vec3 c) -> vec3 { return c + b - a; };
    vec3 q[4];
    q[3] = fourth(p[0], p[1], p[2]);
    q[2] = fourth(p[0], p[1], p[3]);
    q[1] = fourth(p[0], p[2], p[3]);
    q[0] = fourth(p[1], q[2], q[3]);
    AddCorner(p, m);
    AddCorner(q, m);
  }
  // parallelepiped.cu:34-55 with the transform already applied by the caller
  // (the callable is scene code: scenes/cornell_box.cu:62-68).
  typedef void (*TransformFn)(const float in[3], float out[3], void *user);
  Parallelepiped(vec3 lengths, Material *m, TransformFn transform, void *user) {
    vec3 p[4], q[4];
    p[0] = vec3(0.0f);
    for (int i = 1; i <= 3; i++) {
      p[i] = vec3(0.0f);
      p[i][i - 1] = lengths[i - 1];
    }
    q[0] = lengths;
    for (int i = 1; i <= 3; i++) {
      q[i] = lengths;
      q[i][i - 1] = 0;
    }
    for (int i = 0; i < 4; i++) {
      float in[3], o[3];
      in[0] = p[i].x, in[1] = p[i].y, in[2] = p[i].z;
      transform(in, o, user);
      p[i] = vec3(o[0], o[1], o[2]);
      in[0] = q[i].x, in[1] = q[i].y, in[2] = q[i].z;
      transform(in, o, user);
      q[i] = vec3(o[0], o[1], o[2]);
    }
    AddCorner(p, m);
    AddCorner(q, m);
  }
  bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) override {
    return list_.Hit(ray, t_from, t_to, out);
  }
};

// ---------------------------------------------------------------- sky.cu:16-29
struct Sky : Hitable {
  SkyMaterial material_;
  bool Hit(const Ray &, double t_from, double t_to, HitRecord *out) override {
    double t = 1e9;
    if (t_from <= t && t <= t_to) {
      out->material_ptr = &material_;
      out->t = t;  // normal/u/v deliberately left as they were (quirk a14)
      return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------- bvh.cu:6-30
struct AABB {
  vec3 min, max;
  bool CheckOnPlane(const Ray &ray, double t, double t_from, double t_to, int axis) const {
    if (std::isnan(t) || std::isinf(t)) return false;
    if (!(t_from <= t && t <= t_to)) return false;
    vec3 pt = ray.position() + (float)t * ray.direction();
    for (int i = 0; i < 3; i++) {
      if (i == axis) continue;
      if (min[i] <= pt[i] && pt[i] <= max[i]) continue;
      return false;
    }
    return true;
  }
  bool Hit(const Ray &ray, double t_from, double t_to) const {
    if (tl_counters) tl_counters->bvh_boxes++;
    vec3 dir = ray.direction(), pos = ray.position();
    for (int i = 0; i < 3; i++) {
      if (dir[i] == 0.f) continue;
      // float subtraction and float division, widened afterwards (bvh.cu:25)
      double ts[2] = {(double)((min[i] - pos[i]) / dir[i]), (double)((max[i] - pos[i]) / dir[i])};
      if (CheckOnPlane(ray, ts[0], t_from, t_to, i)) return true;
      if (CheckOnPlane(ray, ts[1], t_from, t_to, i)) return true;
    }
    return false;
  }
};

// bvh.cuh:51-66 (Face<false>) and :21-49 (Face<true>)
struct Face {
  vec3 positions_[3];
  vec2 tex_coords_[3];
};

static AABB GetBV(const Face *objs, int n) {  // bvh.cuh:71-82
  AABB aabb;
  aabb.min = vec3(INFINITY);
  aabb.max = vec3(-INFINITY);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++)
      for (int k = 0; k < 3; k++) {
        aabb.min[k] = std::min(objs[i].positions_[j][k], aabb.min[k]);
        aabb.max[k] = std::max(objs[i].positions_[j][k], aabb.max[k]);
      }
  return aabb;
}

// bvh.cuh:101-159
struct BVHNode {
  Face *objs_;
  int n_, mid_ = 0;
  bool has_uv_;
  std::unique_ptr<BVHNode> left_, right_;
  AABB bv_;

  BVHNode(Face *objs, int n, bool has_uv, int k_min) : objs_(objs), n_(n), has_uv_(has_uv) {
    bv_ = GetBV(objs_, n_);
    if (n_ <= k_min) return;
    // GetSplitAxis' result is unused (bvh.cuh:116); sort key is positions_[0].x
    // (bvh.cuh:96-98).  thrust::sort's order of equal keys is implementation
    // defined ("parity unpinned"); the oracle fixes it with a stable sort.
    std::stable_sort(objs_, objs_ + n_,
                     [](const Face &a, const Face &b) { return a.positions_[0].x < b.positions_[0].x; });
    mid_ = (n - 1) / 2;
    left_.reset(new BVHNode(objs_, mid_ + 1, has_uv, k_min));
    right_.reset(new BVHNode(objs_ + mid_ + 1, n - mid_ - 1, has_uv, k_min));
  }

  bool FaceHit(const Face &f, const Ray &ray, double t_from, double t_to, HitRecord *out) const {
    if (tl_counters) tl_counters->bvh_faces++;
    double u, v;
    bool ret = TriangleHit(f.positions_, ray, t_from, t_to, &out->t, &out->normal, &u, &v);
    if (ret && has_uv_) {  // bvh.cuh:40-46
      vec2 tc = f.tex_coords_[0] * (float)(1 - u - v) + f.tex_coords_[1] * (float)u +
                f.tex_coords_[2] * (float)v;
      out->u = tc.x;
      out->v = tc.y;
    }
    return ret;
  }

  bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) const {
    if (!left_) {  // leaf: last hit wins ties (t <= t_to inclusive)
      bool ret = false;
      for (int i = 0; i < n_; i++) {
        HitRecord hit_record;
        if (FaceHit(objs_[i], ray, t_from, t_to, &hit_record)) {
          t_to = hit_record.t;
          *out = hit_record;
          ret = true;
        }
      }
      return ret;
    }
    bool left_hit = false, right_hit = false;
    HitRecord left_record, right_record;
    if (left_->bv_.Hit(ray, t_from, t_to)) left_hit = left_->Hit(ray, t_from, t_to, &left_record);
    if (left_hit) t_to = left_record.t;
    if (right_->bv_.Hit(ray, t_from, t_to)) right_hit = right_->Hit(ray, t_from, t_to, &right_record);
    if (!left_hit && !right_hit) return false;
    if (right_hit)
      *out = right_record;
    else
      *out = left_record;
    return true;
  }
};

// bvh.cuh:161-183
struct BVH : Hitable {
  std::vector<Face> faces_;
  std::unique_ptr<BVHNode> root_;
  Material *material_ptr_;
  BVH(std::vector<Face> faces, bool has_uv, Material *m, int k_min) : faces_(std::move(faces)), material_ptr_(m) {
    root_.reset(new BVHNode(faces_.data(), (int)faces_.size(), has_uv, k_min));
  }
  bool Hit(const Ray &ray, double t_from, double t_to, HitRecord *out) override {
    if (root_->Hit(ray, t_from, t_to, out)) {
      if (material_ptr_ != nullptr) out->material_ptr = material_ptr_;
      return true;
    }
    return false;
  }
};

// ---------------------------------------------------------------- camera.cu:6-77
struct Camera {
  vec3 position_, lower_left_corner_, horizontal_, vertical_, u_, v_, w_;
  bool is_defocus_camera_ = false;
  double lens_radius_ = -1;

  void InitDefocus(vec3 position, vec3 look_at, vec3 up, double fov, double aspect, double aperture,
                   double focus_distance) {  // camera.cu:6-22
    is_defocus_camera_ = true;
    position_ = position;
    w_ = normalize(position - look_at);
    u_ = normalize(cross(up, w_));
    v_ = normalize(cross(w_, u_));
    double half_height = focus_distance * std::tan(fov / 2);
    double half_width = aspect * half_height;
    horizontal_ = u_ * static_cast<float>(2 * half_width);
    vertical_ = v_ * static_cast<float>(2 * half_height);
    lower_left_corner_ =
        position - w_ - u_ * static_cast<float>(half_width) - v_ * static_cast<float>(half_height);
    lens_radius_ = aperture / 2;
  }
  void InitPinhole(vec3 position, vec3 look_at, vec3 up, double fov, double aspect) {  // camera.cu:24-38
    is_defocus_camera_ = false;
    position_ = position;
    w_ = normalize(position - look_at);
    u_ = normalize(cross(up, w_));
    v_ = normalize(cross(w_, u_));
    double half_height = std::tan(fov / 2);
    double half_width = aspect * half_height;
    horizontal_ = u_ * static_cast<float>(2 * half_width);
    vertical_ = v_ * static_cast<float>(2 * half_height);
    lower_left_corner_ =
        position - w_ - u_ * static_cast<float>(half_width) - v_ * static_cast<float>(half_height);
  }
  void InitRaw(vec3 position, vec3 llc, vec3 horizontal, vec3 vertical) {  // camera.cu:40-47
    is_defocus_camera_ = false;
    position_ = position;
    lower_left_corner_ = llc;
    horizontal_ = horizontal;
    vertical_ = vertical;
  }
  // camera.cu:74-77 — a square, not a disk; arguments drawn left to right (quirk g6)
  vec2 DiskRand(float radius, Xorwow *state) const {
    float a = random_float(0, radius, state);
    float b = random_float(0, radius, state);
    return vec2(a, b);
  }
  // camera.cu:57-70
  Ray RayAt(double x, double y, Xorwow *state) const {
    x = (x + 1) / 2;
    y = (y + 1) / 2;
    vec3 target = lower_left_corner_ + static_cast<float>(x) * horizontal_ + static_cast<float>(y) * vertical_;
    vec3 origin;
    if (is_defocus_camera_) {
      vec2 offset = DiskRand((float)lens_radius_, state);
      origin = position_ + u_ * offset.x + v_ * offset.y;
    } else {
      origin = position_;
    }
    return Ray(origin, normalize(target - origin));
  }
};

// ---------------------------------------------------------------- ray_tracing.cu:12-54
// TRACE_DEPTH_LIMIT is a compile-time 10 in the reference (ray_tracing.cu:10);
// BASELINE configs ask for 8 / 50, so it is a run-time parameter here.
constexpr int kMaxDepthLimit = 64;  // == RTMI_MAX_DEPTH of include/rtmi.h; orc_render clamps to it
static vec3 Trace(HitableList *world, Ray ray, Xorwow *state, int depth_limit, uint32_t *ray_count) {
  // Layer storage[TRACE_DEPTH_LIMIT] (ray_tracing.cu:13): a fixed array, as in the reference; only
  // emitted/attenuation are read back (pos/target/t feed the dead DebugTracePath).
  struct Layer {
    vec3 emitted, attenuation;
  };
  Layer storage[kMaxDepthLimit];
  int n_layers = 0;
  vec3 result;
  for (int depth = 0;; depth++) {
    HitRecord record;
    (*ray_count)++;
    bool hit = world->Hit(ray, 1e-3, INFINITY, &record);
    if (!hit || depth >= depth_limit) {
      result = vec3(0, 0, 0);
      break;
    }
    Material *material_ptr = record.material_ptr;
    vec3 attenuation;
    Ray reflection;
    bool scattered = material_ptr->Scatter(ray, record, state, &attenuation, &reflection);
    vec3 hit_point = ray.position() + (float)record.t * ray.direction();
    vec3 emitted = material_ptr->Emit(record.u, record.v, hit_point);
    if (!scattered) {
      result = emitted;
      break;
    }
    storage[n_layers++] = {emitted, attenuation};
    ray = reflection;
  }
  for (int i = n_layers - 1; i >= 0; i--) result = storage[i].emitted + storage[i].attenuation * result;
  return result;
}

struct Scene {
  HitableList world;
  Camera camera;
  bool has_camera = false;
  std::vector<std::unique_ptr<Texture>> textures;
  std::vector<std::unique_ptr<Material>> materials;
  std::vector<std::unique_ptr<Hitable>> hitables;
  std::vector<HitableList *> open_lists;  // nested HitableLists under construction (innermost last)
  Counters totals;
};

// ray_tracing.cu:56-85, one pixel.
static void RenderPixel(Scene *s, int i, int j, int height, int width, int spp, int depth_limit, bool post,
                        Xorwow *state, float *out_rgb, uint32_t *out_rays) {
  vec3 color(0.0f);
  uint32_t rays = 0;
  for (int k = 0; k < spp; k++) {
    double x = ((double)random_float(0, 1, state) + double(j)) / double(width);
    double y = ((double)random_float(0, 1, state) + double(height - i)) / double(height);  // quirk g1
    x = 2 * x - 1;
    y = 2 * y - 1;
    Ray ray = s->camera.RayAt(x, y, state);
    vec3 temp = Trace(&s->world, ray, state, depth_limit, &rays);
    color = color + temp;
  }
  if (post) {
    color = color / float(spp);
    color = vec3(clampf(color.x, 0.f, 1.f), clampf(color.y, 0.f, 1.f), clampf(color.z, 0.f, 1.f));
    color = vec3(std::sqrt(color.x), std::sqrt(color.y), std::sqrt(color.z));
  }
  out_rgb[0] = color.x;
  out_rgb[1] = color.y;
  out_rgb[2] = color.z;
  if (out_rays) *out_rays = rays;
}

}  // namespace orc

// ======================================================================= C ABI
using namespace orc;

extern "C" {

typedef struct orc_scene orc_scene;
static Scene *S(orc_scene *s) { return reinterpret_cast<Scene *>(s); }
static vec3 V(const float *p) { return vec3(p[0], p[1], p[2]); }

orc_scene *orc_scene_new(void) { return reinterpret_cast<orc_scene *>(new Scene()); }
void orc_scene_free(orc_scene *s) { delete S(s); }

int orc_constant_texture(orc_scene *s, const float rgb[3]) {
  S(s)->textures.emplace_back(new ConstantTexture(V(rgb)));
  return (int)S(s)->textures.size() - 1;
}
int orc_image_texture(orc_scene *s, const uint8_t *rgba, int h, int w) {
  auto *t = new ImageTexture();
  t->rgba_.assign(rgba, rgba + (size_t)h * w * 4);
  t->h_ = h;
  t->w_ = w;
  S(s)->textures.emplace_back(t);
  return (int)S(s)->textures.size() - 1;
}
static int add_mat(orc_scene *s, Material *m) {
  S(s)->materials.emplace_back(m);
  return (int)S(s)->materials.size() - 1;
}
int orc_lambertian(orc_scene *s, const float rgb[3]) { return add_mat(s, new Lambertian(V(rgb))); }
int orc_lambertian_tex(orc_scene *s, int tex) { return add_mat(s, new Lambertian(S(s)->textures[tex].get())); }
int orc_metal(orc_scene *s, const float rgb[3], float fuzz) { return add_mat(s, new Metal(V(rgb), fuzz)); }
int orc_dielectric(orc_scene *s, const float rgb[3], double index) {
  return add_mat(s, new Dielectric(V(rgb), index));
}
int orc_diffuse_light(orc_scene *s, int tex) { return add_mat(s, new DiffuseLight(S(s)->textures[tex].get())); }

static Material *M(orc_scene *s, int mat) { return mat < 0 ? nullptr : S(s)->materials[mat].get(); }
static int add_hit(orc_scene *s, Hitable *h) {
  S(s)->hitables.emplace_back(h);
  HitableList *into = S(s)->open_lists.empty() ? &S(s)->world : S(s)->open_lists.back();
  return into->Append(h) ? 0 : -1;
}
// A HitableList appended to a HitableList (hitable_list.cuh:8: the list is itself a Hitable): the
// hitables added between begin and end are its entries, and its Hit() is the nested call of
// hitable_list.cu:7-25, not an inlined scan.
int orc_list_begin(orc_scene *s) {
  HitableList *l = new HitableList();
  if (add_hit(s, l) != 0) return -1;
  S(s)->open_lists.push_back(l);
  return 0;
}
int orc_list_end(orc_scene *s) {
  if (S(s)->open_lists.empty()) return -1;
  S(s)->open_lists.pop_back();
  return 0;
}
int orc_add_sphere(orc_scene *s, const float c[3], double r, int mat) {
  return add_hit(s, new Sphere(V(c), r, M(s, mat)));
}
int orc_add_triangle(orc_scene *s, const float p[9], int mat) {
  vec3 q[3] = {V(p), V(p + 3), V(p + 6)};
  return add_hit(s, new Triangle(q, M(s, mat)));
}
int orc_add_parallelogram(orc_scene *s, const float p[9], int mat) {
  vec3 q[3] = {V(p), V(p + 3), V(p + 6)};
  return add_hit(s, new Parallelogram(q, M(s, mat)));
}
int orc_add_parallelepiped(orc_scene *s, const float p[12], int mat) {
  vec3 q[4] = {V(p), V(p + 3), V(p + 6), V(p + 9)};
  return add_hit(s, new Parallelepiped(q, M(s, mat)));
}
int orc_add_parallelepiped_lengths(orc_scene *s, const float lengths[3], int mat,
                                   Parallelepiped::TransformFn transform, void *user) {
  return add_hit(s, new Parallelepiped(V(lengths), M(s, mat), transform, user));
}
int orc_add_sky(orc_scene *s) { return add_hit(s, new Sky()); }
// faces: n * 9 floats (positions); uvs: n * 6 floats or NULL (Face<false>).
int orc_add_bvh(orc_scene *s, const float *faces, const float *uvs, int n, int mat, int k_min) {
  std::vector<Face> f((size_t)n);
  for (int i = 0; i < n; i++)
    for (int j = 0; j < 3; j++) {
      f[i].positions_[j] = V(faces + (size_t)i * 9 + j * 3);
      if (uvs) f[i].tex_coords_[j] = vec2(uvs[(size_t)i * 6 + j * 2], uvs[(size_t)i * 6 + j * 2 + 1]);
    }
  return add_hit(s, new BVH(std::move(f), uvs != nullptr, M(s, mat), k_min));
}

void orc_camera_pinhole(orc_scene *s, const float pos[3], const float look_at[3], const float up[3], double fov,
                        double aspect) {
  S(s)->camera.InitPinhole(V(pos), V(look_at), V(up), fov, aspect);
  S(s)->has_camera = true;
}
void orc_camera_defocus(orc_scene *s, const float pos[3], const float look_at[3], const float up[3], double fov,
                        double aspect, double aperture, double focus) {
  S(s)->camera.InitDefocus(V(pos), V(look_at), V(up), fov, aspect, aperture, focus);
  S(s)->has_camera = true;
}
void orc_camera_raw(orc_scene *s, const float pos[3], const float llc[3], const float horiz[3],
                    const float vert[3]) {
  S(s)->camera.InitRaw(V(pos), V(llc), V(horiz), V(vert));
  S(s)->has_camera = true;
}
// out: position, llc, horizontal, vertical, u, v, w (21 floats)
void orc_camera_get(orc_scene *s, float out[21]) {
  const Camera &c = S(s)->camera;
  const vec3 *vs[7] = {&c.position_, &c.lower_left_corner_, &c.horizontal_, &c.vertical_, &c.u_, &c.v_, &c.w_};
  for (int i = 0; i < 7; i++) {
    out[i * 3] = vs[i]->x;
    out[i * 3 + 1] = vs[i]->y;
    out[i * 3 + 2] = vs[i]->z;
  }
}

// --------------------------------------------------------------------- RNG
// states: n * 6 uint32 {d, v0..v4}; state i is curand_init(seed, first + i, 0).
void orc_rng_init(uint64_t seed, uint32_t *states, int64_t first, int64_t n) {
  for (int64_t i = 0; i < n; i++) {
    Xorwow x = xorwow_init(seed, (uint64_t)(first + i));
    std::memcpy(states + i * 6, &x, 24);
  }
}
uint32_t orc_rng_next(uint32_t *state) { return xorwow_next(reinterpret_cast<Xorwow *>(state)); }
float orc_rng_uniform(uint32_t *state) { return xorwow_uniform(reinterpret_cast<Xorwow *>(state)); }
float orc_random_float(float mn, float mx, uint32_t *state) {
  return random_float(mn, mx, reinterpret_cast<Xorwow *>(state));
}
// Apply A^(2^(67+k)) to v (exposed so the test can compare with rocRAND).
void orc_rng_jump_pow2(uint32_t v[5], int k) { jump_apply(sequence_jump_matrices()[k], v); }
void orc_rng_step_v(uint32_t v[5]) { xorwow_step_v(v); }

// --------------------------------------------------------------------- utils.cu:111-113
int orc_get_workload(int rank, int world_size, int spp) { return spp / world_size + (int)(rank < (spp % world_size)); }

// --------------------------------------------------------------------- single-function probes for known-answer tests
// returns hit flag; out = {t, u, v, nx, ny, nz}
int orc_probe_hit(orc_scene *s, const float o[3], const float d[3], double t_from, double t_to, double out[6],
                  int *out_mat) {
  Ray r(V(o), V(d));
  HitRecord rec;
  bool hit = S(s)->world.Hit(r, t_from, t_to, &rec);
  if (hit) {
    out[0] = rec.t, out[1] = rec.u, out[2] = rec.v;
    out[3] = rec.normal.x, out[4] = rec.normal.y, out[5] = rec.normal.z;
    if (out_mat) {
      *out_mat = -1;
      for (size_t i = 0; i < S(s)->materials.size(); i++)
        if (S(s)->materials[i].get() == rec.material_ptr) *out_mat = (int)i;
    }
  }
  return hit ? 1 : 0;
}
// Material::Scatter for material `mat`; returns scattered flag, out = {att rgb, origin xyz, dir xyz}
int orc_probe_scatter(orc_scene *s, int mat, const float o[3], const float d[3], double t, const float n[3],
                      uint32_t *state, float out[9]) {
  Ray r(V(o), V(d));
  HitRecord rec;
  rec.t = t;
  rec.normal = V(n);
  vec3 att;
  Ray nr;
  bool sc = S(s)->materials[mat]->Scatter(r, rec, reinterpret_cast<Xorwow *>(state), &att, &nr);
  if (sc) {
    out[0] = att.x, out[1] = att.y, out[2] = att.z;
    out[3] = nr.position().x, out[4] = nr.position().y, out[5] = nr.position().z;
    out[6] = nr.direction().x, out[7] = nr.direction().y, out[8] = nr.direction().z;
  }
  return sc ? 1 : 0;
}
void orc_probe_camera_ray(orc_scene *s, double x, double y, uint32_t *state, float out[6]) {
  Ray r = S(s)->camera.RayAt(x, y, reinterpret_cast<Xorwow *>(state));
  out[0] = r.position().x, out[1] = r.position().y, out[2] = r.position().z;
  out[3] = r.direction().x, out[4] = r.direction().y, out[5] = r.direction().z;
}

// --------------------------------------------------------------------- render
// Renders pixels pixel_ids[0..n) (global idx = i*width + j; NULL => all H*W in
// order).  states/out_rgb/out_rays are indexed by GLOBAL pixel idx, like the
// reference's states+idx and out_image[idx] (ray_tracing.cu:64,84).
// Returns total rays (closest-hit queries).
uint64_t orc_render(orc_scene *s, int height, int width, int spp, int depth_limit, int post, uint32_t *states,
                    float *out_rgb, uint32_t *out_rays, const int32_t *pixel_ids, int64_t n_pixels,
                    int n_threads) {
  Scene *sc = S(s);
  if (!pixel_ids) n_pixels = (int64_t)height * width;
  if (n_threads < 1) n_threads = 1;
  if (depth_limit > kMaxDepthLimit) depth_limit = kMaxDepthLimit;
  std::atomic<int64_t> next(0);
  std::vector<Counters> per(n_threads);
  auto worker = [&](int tid) {
    tl_counters = &per[tid];
    const int64_t chunk = 8;  // pixels per grab: a 256x256 sample still gives 256 threads 32 grabs each
    for (;;) {
      int64_t b = next.fetch_add(chunk);
      if (b >= n_pixels) break;
      int64_t e = std::min(b + chunk, n_pixels);
      for (int64_t q = b; q < e; q++) {
        int64_t idx = pixel_ids ? pixel_ids[q] : q;
        int i = (int)(idx / width), j = (int)(idx % width);
        uint32_t rays = 0;
        RenderPixel(sc, i, j, height, width, spp, depth_limit, post != 0,
                    reinterpret_cast<Xorwow *>(states + idx * 6), out_rgb + idx * 3, &rays);
        if (out_rays) out_rays[idx] = rays;
        per[tid].rays += rays;
      }
    }
    tl_counters = nullptr;
  };
  std::vector<std::thread> th;
  for (int t = 1; t < n_threads; t++) th.emplace_back(worker, t);
  worker(0);
  for (auto &t : th) t.join();
  Counters tot;
  for (auto &c : per) {
    tot.rays += c.rays;
    tot.bvh_boxes += c.bvh_boxes;
    tot.bvh_faces += c.bvh_faces;
  }
  sc->totals = tot;
  return tot.rays;
}
void orc_last_counters(orc_scene *s, uint64_t out[3]) {
  out[0] = S(s)->totals.rays;
  out[1] = S(s)->totals.bvh_boxes;
  out[2] = S(s)->totals.bvh_faces;
}

// GatherImageData's root-side post-process (utils.cu:126-129) on a summed image.
void orc_post_process(float *rgb, int64_t n_pixels, int spp) {
  for (int64_t i = 0; i < n_pixels * 3; i++) rgb[i] = std::sqrt(clampf(rgb[i] / (float)spp, 0.f, 1.f));
}

}  // extern "C"
