// Device helpers of the trace kernel: triangle / box / sphere arithmetic, scalar record loads, texture fetch,
// the rejection sampler.  Included by kernels.hip inside namespace rtmi (not a stand-alone header).
#pragma once

// ------------------------------------------------------------------ trace helpers
template <bool DT>
struct TSel {
  typedef float type;
};
template <>
struct TSel<true> {
  typedef double type;
};

// Winner id: kind in the top 3 bits, index below; bit 28 marks the second
// triangle of a parallelogram.
constexpr uint32_t ID_NONE = 0xffffffffu;
constexpr uint32_t ID_INDEX_MASK = (1u << 28) - 1;
__device__ __forceinline__ uint32_t make_id(int kind, int index) { return ((uint32_t)kind << 29) | (uint32_t)index; }

// The smallest binary32 >= 1e-3 is 0.001f (it rounds up), so for a binary32 t the
// reference's double compare `1e-3 <= t` is `0.001f <= t`; likewise
// `fabs(det) < 1e-7` is `fabsf(det) < 1e-7f` because 1e-7f rounds up.
// (tests/test_host_logic.py::test_float_thresholds pins both facts.)
#define T_FROM_F 0.001f
#define DET_EPS_F 1e-7f

// Correctly rounded 1 / x in three instructions for 2^-126 <= |x| < 2^126: the hardware
// reciprocal (within 1 ulp) and one Newton step on the exact FMA residual.  That the result
// equals the IEEE quotient 1.0f / x for EVERY such x is not argued but checked: the
// arithmetic_selftest kernel compares all 2^32 bit patterns on the device it runs on
// (rtmi_selftest_arithmetic, tests/test_gpu_parity.py).  Outside that range (zero, denormal,
// huge, inf, NaN) callers use the division.
#define RCP_RN_LIMIT 0x1p126f
__device__ __forceinline__ float rcp_rn(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}

// Correctly rounded sqrt(x) in five instructions for 2^-100 <= x < 2^100: the hardware reciprocal square root
// (within 1 ulp), g = x * y and h = y / 2, and one correction on the exact FMA residual x - g * g.  Like rcp_rn this
// is established by exhaustion: arithmetic_selftest compares it with sqrtf on every binary32 of that range (1.68e9
// inputs, mismatches[4]; hipcc's own correctly rounded sqrtf is a 16-instruction sequence).  Outside the range
// (zero, denormal, huge, inf, NaN, negative) callers use sqrtf.
#define SQRT_RN_LO 0x1p-100f
#define SQRT_RN_HI 0x1p100f
__device__ __forceinline__ float sqrt_rn_core(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float g = x * y, h = 0.5f * y;
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
// sqrtf(x), wave-cooperative: the short form when every lane's operand is inside its domain.
#ifndef RTMI_OPT_SQRT
#define RTMI_OPT_SQRT 1
#endif
#ifndef RTMI_OPT_DIV3
#define RTMI_OPT_DIV3 1
#endif
__device__ __forceinline__ float sqrt_rn(float x) {
#if RTMI_OPT_SQRT
  if (__all(x >= SQRT_RN_LO && x < SQRT_RN_HI)) return sqrt_rn_core(x);
#endif
  return sqrtf(x);
}

// a / b for the three coordinates of a vector and one divisor, correctly rounded, without the 11-instruction IEEE
// sequence per quotient: y = RN(1 / b) (rcp_rn, exact by exhaustion), q0 = RN(a * y), and two corrections
// q <- RN(q + RN(a - b * q) * y) on the exact FMA residual.  q0 is within 1.5 ulp of a / b, the first correction
// brings it within (1/2 + 2^-23) ulp -- a faithful rounding -- and for a faithful q and a correctly rounded
// reciprocal the corrected value IS the correctly rounded quotient (Markstein's theorem); no residual underflows
// for a == 0 or |a| >= 2^-40, 2^-40 <= b <= 2 (the caller's operands: sampler coordinates, multiples of 2^-24,
// over their own length).  arithmetic_selftest runs it against the division on 2^32 sampler operand triples
// (mismatches[5], and [6] counts what a single correction would get wrong: 0 of 1e11 in tools/exp).
#define DIV3_RN_LO 0x1p-40f
__device__ __forceinline__ float div_rn_core(float a, float b, float y) {
  float q = a * y;
  q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
  q = __builtin_fmaf(__builtin_fmaf(-b, q, a), y, q);
  return q;
}
// lambertian.cu:25-29: l = sqrt(sum), vec /= l for a point (x, y, z) of the sampler -- coordinates that are
// multiples of 2^-24 in [-1, 1] (rng_pm1_of), sum = x*x + y*y + z*z <= 1 + 2^-23.  One wave-uniform domain check for
// the square root and the three quotients: sum >= 2^-80 (false only for the point (0, 0, 0), probability 2^-72,
// which takes the IEEE forms with the rest of its wave); then l >= 2^-40 and every nonzero |coordinate| >= 2^-24.
__device__ __forceinline__ V3 sampler_on_sphere(V3 a, float sum) {
#if RTMI_OPT_DIV3
  if (__all(sum >= 0x1p-80f)) {
    const float l = sqrt_rn_core(sum);
    const float y = rcp_rn(l);
    return mk(div_rn_core(a.x, l, y), div_rn_core(a.y, l, y), div_rn_core(a.z, l, y));
  }
  const float l = sqrtf(sum);
#else
  const float l = sqrt_rn(sum);
#endif
  return mk(a.x / l, a.y / l, a.z / l);
}

// glm::normalize = v * (1 / sqrt(v.v)) (vec.h: unit3) with the reciprocal taken by rcp_rn: a
// positive normal square root always lies inside rcp_rn's domain (2^-75 < sqrt(x) < 2^64); a
// zero, NaN or infinite one sends the whole wave through the division.
__device__ __forceinline__ V3 unit3_rn(V3 v) {
  const float dd = dot3(v, v);
  float inv;
#if RTMI_OPT_SQRT
  // one wave-uniform check for both short forms: 2^-100 <= v.v < 2^100 puts the square root inside rcp_rn's domain
  if (__all(dd >= SQRT_RN_LO && dd < SQRT_RN_HI)) {
    inv = rcp_rn(sqrt_rn_core(dd));
  } else {
    inv = 1.0f / sqrtf(dd);
  }
#else
  const float s = sqrtf(dd);
  if (__all(__builtin_amdgcn_class(s, 0x100))) {  // +normal
    inv = rcp_rn(s);
  } else {
    inv = 1.0f / s;
  }
#endif
  return v * inv;
}

// normalize(normalize(v)) -- Camera::RayAt then Ray's constructor (camera.cu:69, ray.cu:10), Lambertian::Scatter
// then Ray's constructor (lambertian.cu:41-42) -- with ONE domain check: when v.v passes it the first result has
// 1 - 2^-22 < u.u < 1 + 2^-22, which is inside both short forms' domains.
__device__ __forceinline__ V3 unit3_rn_twice(V3 v) {
#if RTMI_OPT_SQRT
  const float dd = dot3(v, v);
  if (__all(dd >= SQRT_RN_LO && dd < SQRT_RN_HI)) {
    const V3 u = v * rcp_rn(sqrt_rn_core(dd));
    return u * rcp_rn(sqrt_rn_core(dot3(u, u)));
  }
  const V3 u = v * (1.0f / sqrtf(dd));
  return u * (1.0f / sqrtf(dot3(u, u)));
#else
  return unit3_rn(unit3_rn(v));
#endif
}

// utils.cu:49-85 with the ray-independent terms precomputed.
template <typename T>
__device__ __forceinline__ bool tri_test(V3 p0, V3 e1, V3 e2, V3 o, V3 d, T t_to, float &t, float &u, float &v) {
  V3 pvec = cross3(d, e2);
  float det = dot3(e1, pvec);
  if (fabsf(det) < DET_EPS_F) return false;
  float inv;  // 1.0f / det (utils.cu:59)
  if (fabsf(det) < RCP_RN_LIMIT) {
    inv = rcp_rn(det);
  } else {
    inv = 1.0f / det;
  }
  V3 tvec = o - p0;
  u = dot3(tvec, pvec) * inv;
  if (u < 0.0f || u > 1.0f) return false;
  V3 qvec = cross3(tvec, e1);
  v = dot3(d, qvec) * inv;
  if (v < 0.0f || u + v > 1.0f) return false;
  t = dot3(e2, qvec) * inv;
  if (!(T_FROM_F <= t && (T)t <= t_to)) return false;
  return true;
}
// The same test as straight-line code.  In the world-list loop all 64 lanes test the SAME
// triangle with unrelated rays, so some lane nearly always survives each early-out and
// the exec-mask branches only cost scalar instructions; the boolean results are formed
// exactly as above (NaNs included), only without control flow.
// `pvec` = cross(d, e2) is passed in: a parallelogram's second triangle (p1,p2,p3) has the same
// e2 = p3 - p1 = p2 - p0 as the first whenever the corner arithmetic was exact (TRI_SAME_E2,
// decided on the host by comparing bit patterns) and then reuses the first one's product.
// `det_safe` (wave-uniform, SceneDev::det_safe): the scene's edges bound |det| below rcp_rn's limit for every
// unit direction, so no lane can need the division (a NaN direction gives NaN either way) and the wave-wide
// check -- a vector compare the scalar branch has to wait for -- is skipped.
template <typename T>
__device__ __forceinline__ bool tri_test_flat(V3 p0, V3 e1, V3 e2, V3 pvec, V3 o, V3 d, T t_to, float &t, float &u,
                                              float &v, bool det_safe = false) {
  float det = dot3(e1, pvec);
  bool ok = !(fabsf(det) < DET_EPS_F);
  // 1.0f / det (utils.cu:59).  Lanes with |det| < 1e-7 have ok == false and never look at inv;
  // for the others rcp_rn is the IEEE quotient unless some |det| >= 2^126 (or NaN), in which
  // case the whole wave divides.
  float inv;
  if (det_safe || __all(fabsf(det) < RCP_RN_LIMIT)) {
    inv = rcp_rn(det);
  } else {
    inv = 1.0f / det;
  }
  V3 tvec = o - p0;
  u = dot3(tvec, pvec) * inv;
  ok = ok & !((u < 0.0f) | (u > 1.0f));
  V3 qvec = cross3(tvec, e1);
  v = dot3(d, qvec) * inv;
  ok = ok & !((v < 0.0f) | (u + v > 1.0f));
  t = dot3(e2, qvec) * inv;
  ok = ok & ((T_FROM_F <= t) & ((T)t <= t_to));
  return ok;
}

// mask = 2 * mask + (a <= b): the compare writes the lane's verdict to VCC and the add takes it as its carry-in --
// two instructions where the compiler's select / or takes four (a VOP3 reads one SGPR at most on gfx9).
__device__ __forceinline__ void cull_step(uint32_t &mask, float a, float b) {
  asm("v_cmp_le_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(a), "v"(b) : "vcc");
}

// mask = 2 * mask + !(a < b)  (true for a NaN: what the pre-test cannot decide stays a candidate)
__device__ __forceinline__ void cull_step_nlt(uint32_t &mask, float a, float b) {
  asm("v_cmp_nlt_f32_e32 vcc, %1, %2\n\tv_addc_co_u32_e32 %0, vcc, %0, %0, vcc" : "+v"(mask) : "v"(a), "v"(b) : "vcc");
}

// Loads whose address space is spelt out.  `staged ? lds_pointer : global_pointer` followed by one load makes the
// compiler select the POINTER and issue a flat load (which takes the vector-memory path even for LDS data and
// waits on both counters); with the address spaces in the types the two branches stay a ds_read and a global_load.
#define RT_LDS __attribute__((address_space(3)))
#define RT_GLOBAL __attribute__((address_space(1)))
typedef float rt_f32x4 __attribute__((ext_vector_type(4)));
typedef float rt_f32x2 __attribute__((ext_vector_type(2)));
template <typename V>
struct LoadAs;
template <>
struct LoadAs<float4> {
  typedef rt_f32x4 raw;
  static __device__ __forceinline__ float4 cvt(raw v) { return make_float4(v[0], v[1], v[2], v[3]); }
};
template <>
struct LoadAs<float2> {
  typedef rt_f32x2 raw;
  static __device__ __forceinline__ float2 cvt(raw v) { return make_float2(v[0], v[1]); }
};
template <typename V>
__device__ __forceinline__ V load_lds(const void *p) {
  return LoadAs<V>::cvt(*(const RT_LDS typename LoadAs<V>::raw *)p);
}
template <typename V>
__device__ __forceinline__ V load_global(const void *p) {
  return LoadAs<V>::cvt(*(const RT_GLOBAL typename LoadAs<V>::raw *)p);
}

// World-list triangle records are read through the constant address space: a
// wave-uniform address there always selects scalar loads (one s_load_dwordx16 per
// record into SGPRs) instead of per-lane vector loads.
#define RT_CONSTANT __attribute__((address_space(4)))
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ f32x16 load_hot_tri(const HotTri *base, int idx) {
  return *(const RT_CONSTANT f32x16 *)(uintptr_t)(base + idx);
}
__device__ __forceinline__ f32x8 load_pair_box(const PairBox *base, int idx) {
  return *(const RT_CONSTANT f32x8 *)(uintptr_t)(base + idx);
}
__device__ __forceinline__ f32x8 load_sphere(const SphereRec *base, int idx) {
  return *(const RT_CONSTANT f32x8 *)(uintptr_t)(base + idx);
}
__device__ __forceinline__ f32x8 load_sph_group(const SphGroup *base, int idx) {
  return *(const RT_CONSTANT f32x8 *)(uintptr_t)(base + idx);
}
__device__ __forceinline__ f32x8 load_sph_member(const SphMember *base, int idx) {
  return *(const RT_CONSTANT f32x8 *)(uintptr_t)(base + idx);
}
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x4 load_run(const Run *base, int idx) {
  return *(const RT_CONSTANT i32x4 *)(uintptr_t)(base + idx);
}
__device__ __forceinline__ i32x16 load_bvh_rec(const BvhRec *base, int idx) {
  return *(const RT_CONSTANT i32x16 *)(uintptr_t)(base + idx);
}

// bvh.cu:6-30 — "the segment crosses the box surface"; a box that wholly
// contains [t_from, t_to] reports false (quirk g8).
template <typename T>
__device__ __forceinline__ bool aabb_test(const BvhNode &nd, V3 o, V3 d, T t_to) {
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
#pragma unroll
  for (int i = 0; i < 3; i++) {
    if (dd[i] == 0.f) continue;
#pragma unroll
    for (int s = 0; s < 2; s++) {
      float plane = s == 0 ? nd.mn[i] : nd.mx[i];
      float tf = (plane - oo[i]) / dd[i];
      if (!(fabsf(tf) < INFINITY)) continue;  // isnan || isinf
      if (!(T_FROM_F <= tf && (T)tf <= t_to)) continue;
      bool inside = true;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        if (a == i) continue;
        float pa = oo[a] + tf * dd[a];
        if (!(nd.mn[a] <= pa && pa <= nd.mx[a])) inside = false;
      }
      if (inside) return true;
    }
  }
  return false;
}

// The T-independent part of AABB::Hit (bvh.cu:6-30): the smallest plane-crossing time tf that is
// finite, >= t_from and whose crossing point lies inside the box on the other two axes -- a NaN if no
// plane qualifies.  AABB::Hit(box, [t_from, T]) is then exactly `crossing_time <= T`: each plane's
// own test is `tf <= T` AND these T-independent conditions, and the box test is their OR.  (The "none"
// marker must compare false against EVERY T: +infinity would not -- T is +infinity for a mesh that comes first in
// the world list -- and did not until round 3: a face accepted by the binary32 triangle test from far outside
// its leaf's box, rays 1e3 away from millimetre faces, was then reported although the reference never enters
// that leaf.  tests/test_gpu_round3.py::test_far_false_accepts_outside_their_leaf_box.)
__device__ __forceinline__ float aabb_crossing_time(const BvhNode &nd, V3 o, V3 d) {
  const float oo[3] = {o.x, o.y, o.z}, dd[3] = {d.x, d.y, d.z};
  float m = INFINITY;
#pragma unroll
  for (int i = 0; i < 3; i++) {
#pragma unroll
    for (int s = 0; s < 2; s++) {
      const float plane = s == 0 ? nd.mn[i] : nd.mx[i];
      const float tf = (plane - oo[i]) / dd[i];
      bool okp = dd[i] != 0.f && fabsf(tf) < INFINITY && T_FROM_F <= tf;
#pragma unroll
      for (int a = 0; a < 3; a++) {
        if (a == i) continue;
        const float pa = oo[a] + tf * dd[a];
        okp = okp && nd.mn[a] <= pa && pa <= nd.mx[a];
      }
      m = okp ? fminf(m, tf) : m;
    }
  }
  return m < INFINITY ? m : __int_as_float(0x7fc00000);  // (every qualifying tf is finite)
}

// ImageTexture::Value (textures/image_texture.cu:9-15): tex2D<float4> of an RGBA8 image, point filter, wrap,
// normalised coordinates.  The texel is fetched as its three colour bytes (r | g << 8 | b << 16) and turned into
// the texture unit's floats -- byte / 255 -- by texel_rgb: a layer of the radiance fold can then be kept as ONE
// word and its colour re-formed, by the same division, when the path is folded.
__device__ __forceinline__ uint32_t tex_fetch(const TexRec &tx, float u, float v) {
  float fu = u - floorf(u), fv = v - floorf(v);
  int ix = (int)floorf(fu * (float)tx.width);
  int iy = (int)floorf(fv * (float)tx.height);
  ix = ix > tx.width - 1 ? tx.width - 1 : ix;
  iy = iy > tx.height - 1 ? tx.height - 1 : iy;
  ix = ix < 0 ? 0 : ix;
  iy = iy < 0 ? 0 : iy;
  const uint32_t px = *reinterpret_cast<const uint32_t *>(tx.rgba + (size_t)iy * tx.pitch + (size_t)ix * 4);
  return px & 0xffffffu;
}
__device__ __forceinline__ V3 texel_rgb(uint32_t px) {
  return mk((float)(px & 0xffu) / 255.0f, (float)((px >> 8) & 0xffu) / 255.0f, (float)((px >> 16) & 0xffu) / 255.0f);
}
__device__ __forceinline__ V3 tex_sample(const TexRec &tx, float u, float v) { return texel_rgb(tex_fetch(tx, u, v)); }

// lambertian.cu:19-31 / metal.cu:27-36: rejection-sample the unit ball.
// l = (float)pow((double)(x*x+y*y+z*z), 0.5) == sqrtf(sum) (double rounding of a square
// root of a binary32 value is innocuous).  The loop condition `l > 1` is decided without
// the square root: sqrtf is monotone, sqrtf(1 + 2^-23) rounds to exactly 1 and
// sqrtf(1 + 2^-22) to 1 + 2^-23, so sqrtf(s) > 1  <=>  s > 1 + 2^-23
// (tests/test_host_logic.py::test_rejection_threshold).  Returns the accepted sum.
#define BALL_S_MAX 1.00000011920928955078125f /* 1 + 2^-23 */
// CudaRandomFloat(-1, 1) and (0, 1) (utils.cuh:22-27 over curand_uniform) in fewer instructions.
// (float)x * 2^-32 is exact, so curand_uniform's u = x * 2^-32 + 2^-33 is one fused
// multiply-add; u * (1 - (-1)) is an exact doubling that commutes with the rounding of u, so
// u * 2 + (-1) = fma(x, 2^-31, 2^-32) - 1; and u * (1 - 0) + 0 = u.  Both are also compared with
// rng_range() on all 2^32 draws by arithmetic_selftest.
__device__ __forceinline__ float rng_pm1_of(uint32_t x) { return __builtin_fmaf((float)x, 0x1p-31f, 0x1p-32f) - 1.0f; }
__device__ __forceinline__ float rng_01_of(uint32_t x) { return __builtin_fmaf((float)x, 0x1p-32f, 0x1p-33f); }
__device__ __forceinline__ float rng_pm1(Rng &s) { return rng_pm1_of(rng_next(s)); }
__device__ __forceinline__ float rng_01(Rng &s) { return rng_01_of(rng_next(s)); }
__device__ __forceinline__ V3 ball_sample(Rng &rng, float &sum) {
  float x, y, z;
  do {
    x = rng_pm1(rng);
    y = rng_pm1(rng);
    z = rng_pm1(rng);
    sum = x * x + y * y + z * z;
  } while (sum > BALL_S_MAX);
  return mk(x, y, z);
}

// Conservative "does the segment [lo, hi] of the ray come anywhere near this box" test, used
// (a) on the padded sub-tree nodes and (b) as a cheap pre-reject in front of the reference's
// exact AABB::Hit: if the ray never touches the box inflated by `pad`, no plane-crossing point
// can lie on its surface.  `inv_d` is the clamped reciprocal of safe_inverse(): finite, so no
// 0 * inf NaN can appear and a zero direction component needs no branch (an origin outside the
// slab then yields two huge same-sign crossings, i.e. a miss; inside, a huge interval).  The
// relative slack is applied once to the merged entry/exit (t - |t| eps is monotone in t).
__device__ __forceinline__ float safe_inverse(float x) {
  return fabsf(x) < 1e-30f ? copysignf(1e30f, x) : 1.0f / x;
}
__device__ __forceinline__ bool slab_touch(const BvhNode &nd, float pad, V3 o, V3 inv_d, float lo, float hi) {
  const float t0x = (nd.mn[0] - pad - o.x) * inv_d.x, t1x = (nd.mx[0] + pad - o.x) * inv_d.x;
  const float t0y = (nd.mn[1] - pad - o.y) * inv_d.y, t1y = (nd.mx[1] + pad - o.y) * inv_d.y;
  const float t0z = (nd.mn[2] - pad - o.z) * inv_d.z, t1z = (nd.mx[2] + pad - o.z) * inv_d.z;
  const float enter = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
  const float leave = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
  lo = fmaxf(lo, enter - fabsf(enter) * kSlabTimeRel);
  hi = fminf(hi, leave + fabsf(leave) * kSlabTimeRel);
  return lo <= hi;
}
