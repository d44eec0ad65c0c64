// HIP kernels of the trace path for gfx950 (MI355X): RNG seeding, the persistent
// per-pixel trace loop, tile scatter and post-process.
//
// Shape of the trace kernel (DESIGN.md "Kernels"):
//   * persistent lanes: a lane owns ONE pixel at a time and walks its samples
//     serially (the cuRAND stream of a pixel is consumed in order, so samples of
//     a pixel cannot be spread over lanes without changing the image); when a
//     path ends the lane immediately regenerates the next camera ray, when the
//     pixel ends it pulls the next work item from a global queue
//     (wave-aggregated atomic), so every lane of a wave enters the intersection
//     loop on every iteration;
//   * closest hit walks the world list in list order with wave-uniform control
//     flow: every lane tests the same primitive, whose record arrives in SGPRs
//     through scalar loads (s_load_dwordx8/x16, scalar cache) — no per-lane
//     geometry traffic at all for list scenes; lists of four or more triangle pairs
//     are culled first (each lane slab-tests every pair's padded bounds, the surviving
//     (ray, pair) candidates of the whole wave are tested by all 64 lanes from LDS and
//     folded per ray in list order);
//   * acceptance bookkeeping in the loop is 3 VGPRs (ok, t_to, winner id); the
//     winner's normal / material are resolved once per ray after the loop;
//   * the material table is staged in LDS once per workgroup, and so is the per-lane
//     stack of scattered-material ids that the back-to-front radiance fold of
//     ray_tracing.cu:50-52 needs (four bits or a byte per bounce instead of a 12-byte
//     attenuation in scratch memory);
//   * mesh (BVH) queries do not walk the reference's tree: one search of a mesh-wide
//     4-wide tree, done by the wave for all its rays at once (mesh_search), finds the best
//     face of every reference leaf that holds a hit, then the reference's box tests are
//     replayed on those leaves' root-to-leaf paths only, again one (leaf, level) per lane
//     (closest_hit, RUN_BVH);
//   * arithmetic follows the reference operation by operation (binary32 with
//     the binary64 islands of sphere.cu / ray_tracing.cu:68-73); the file is
//     compiled with -ffp-contract=off and IEEE divide/sqrt.  The one shortened
//     operation, 1.0f / det, is checked against the IEEE quotient on all 2^32 inputs
//     by arithmetic_selftest (as are the two fused uniform-variate conversions).
//
// Reference functions restated here (paths relative to
// /root/reference/ray-tracing-cuda/): PathTracing + Trace ray_tracing.cu:12-85,
// Camera::RayAt camera.cu:57-77, HitableList::Hit hitable_list.cu:7-25,
// Sphere::Hit sphere.cu:11-64, TriangleHit utils.cu:49-85, Parallelogram::Hit
// parallelogram.cu:17-44, Sky sky.cu:9-27, AABB::Hit bvh.cu:6-30, BVHNode::Hit
// bvh.cuh:123-158, Lambertian lambertian.cu:19-43, Metal metal.cu:12-36,
// Dielectric dielectric.cu:16-44, DiffuseLight diffuse_light.cu:5-13,
// ImageTexture::Value textures/image_texture.cu:9-15, CudaRandomInit utils.cu:43-47.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "kernels.h"

#include <cstdlib>
#include "scene_dev.h"
#include "vec.h"
#include "xorwow.h"

namespace rtmi {

// ------------------------------------------------------------------ frame math
__host__ __device__ __forceinline__ int64_t frame_pixel_of_rank(const FrameDev &fr, int rank, int64_t q) {
  int64_t lt = q >> 6;
  int w = (int)(q & 63);
  int64_t gt = lt * fr.world + rank;
  if (gt >= fr.n_tiles) return -1;
  int ty = (int)(gt / fr.tiles_x), tx = (int)(gt % fr.tiles_x);
  int i = ty * 8 + (w >> 3), j = tx * 8 + (w & 7);
  if (i >= fr.height || j >= fr.width) return -1;
  return (int64_t)i * fr.width + j;
}

int64_t frame_pixel_of(const FrameDev &fr, int rank, int64_t q) { return frame_pixel_of_rank(fr, rank, q); }

// ------------------------------------------------------------------ RNG init
// state(q) = seed scramble, then v <- A^(idx * 2^67) v, idx = global pixel index.
// The jump matrix for bit k is wave-uniform -> scalar loads; lanes whose bit is
// clear keep their vector.
__global__ __launch_bounds__(256) void rng_init_kernel(uint64_t seed, FrameDev fr,
                                                        const uint32_t *__restrict__ jump,
                                                        uint32_t *__restrict__ states) {
  int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (q >= fr.items) return;
  int64_t idx = frame_pixel_of_rank(fr, fr.rank, q);
  uint64_t sub = idx < 0 ? 0 : (uint64_t)idx;
  Rng s = rng_seed(seed);
  uint32_t v0 = s.v0, v1 = s.v1, v2 = s.v2, v3 = s.v3, v4 = s.v4;
  for (int k = 0; k < kJumpBits; k++) {
    if (!__any((sub >> k) != 0)) break;
    bool bit = (sub >> k) & 1;
    if (!__any(bit)) continue;
    const uint32_t *M = jump + (size_t)k * kJumpWords;
    uint32_t r0 = 0, r1 = 0, r2 = 0, r3 = 0, r4 = 0;
#pragma unroll 1
    for (int w = 0; w < 5; w++) {
      uint32_t word = w == 0 ? v0 : w == 1 ? v1 : w == 2 ? v2 : w == 3 ? v3 : v4;
#pragma unroll 4
      for (int b = 0; b < 32; b++) {
        uint32_t m = 0u - ((word >> b) & 1u);
        const uint32_t *row = M + (w * 32 + b) * 5;
        r0 ^= row[0] & m;
        r1 ^= row[1] & m;
        r2 ^= row[2] & m;
        r3 ^= row[3] & m;
        r4 ^= row[4] & m;
      }
    }
    if (bit) {
      v0 = r0, v1 = r1, v2 = r2, v3 = r3, v4 = r4;
    }
  }
  const int64_t n = fr.items;
  states[0 * n + q] = s.d;
  states[1 * n + q] = v0;
  states[2 * n + q] = v1;
  states[3 * n + q] = v2;
  states[4 * n + q] = v3;
  states[5 * n + q] = v4;
}

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

// glm::normalize = v * (1 / sqrt(v.v)) (vec.h: unit3) with the reciprocal taken by rcp_rn: a
// positive normal square root always lies inside rcp_rn's domain (2^-75 < sqrt(x) < 2^64); a
// zero, NaN or infinite one sends the whole wave through the division.
__device__ __forceinline__ V3 unit3_rn(V3 v) {
  const float s = sqrtf(dot3(v, v));
  float inv;
  if (__all(__builtin_amdgcn_class(s, 0x100))) {  // +normal
    inv = rcp_rn(s);
  } else {
    inv = 1.0f / s;
  }
  return v * inv;
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
template <typename T>
__device__ __forceinline__ bool tri_test_flat(V3 p0, V3 e1, V3 e2, V3 pvec, V3 o, V3 d, T t_to, float &t, float &u,
                                              float &v) {
  float det = dot3(e1, pvec);
  bool ok = !(fabsf(det) < DET_EPS_F);
  // 1.0f / det (utils.cu:59).  Lanes with |det| < 1e-7 have ok == false and never look at inv;
  // for the others rcp_rn is the IEEE quotient unless some |det| >= 2^126 (or NaN), in which
  // case the whole wave divides.
  float inv;
  if (__all(fabsf(det) < RCP_RN_LIMIT)) {
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
typedef int i32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ i32x4 load_run(const Run *base, int idx) {
  return *(const RT_CONSTANT i32x4 *)(uintptr_t)(base + idx);
}
__device__ __forceinline__ i32x8 load_bvh_rec(const BvhRec *base, int idx) {
  return *(const RT_CONSTANT i32x8 *)(uintptr_t)(base + idx);
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
// finite, >= t_from and whose crossing point lies inside the box on the other two axes (+inf if no
// plane qualifies).  AABB::Hit(box, [t_from, T]) is then exactly `crossing_time <= T`: each plane's
// own test is `tf <= T` AND these T-independent conditions, and the box test is their OR.
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
  return m;
}

__device__ __forceinline__ V3 tex_sample(const TexRec &tx, float u, float v) {
  float fu = u - floorf(u), fv = v - floorf(v);
  int ix = (int)floorf(fu * (float)tx.width);
  int iy = (int)floorf(fv * (float)tx.height);
  ix = ix > tx.width - 1 ? tx.width - 1 : ix;
  iy = iy > tx.height - 1 ? tx.height - 1 : iy;
  ix = ix < 0 ? 0 : ix;
  iy = iy < 0 ? 0 : iy;
  const uint8_t *px = tx.rgba + (size_t)iy * tx.pitch + (size_t)ix * 4;
  return mk((float)px[0] / 255.0f, (float)px[1] / 255.0f, (float)px[2] / 255.0f);
}

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
  lo = fmaxf(lo, enter - fabsf(enter) * 1e-5f);
  hi = fminf(hi, leave + fabsf(leave) * 1e-5f);
  return lo <= hi;
}


// ------------------------------------------------------------------ mesh search (wave-wide)
// The search of a mesh's 4-wide tree is done by the WAVE, not by the lane that owns the ray
// (DESIGN.md "Mesh queries").  What has to be found is every face the triangle test accepts with
// t_from <= t <= t_to -- no pruning by nearer hits, because the reference's box semantics (quirk g8)
// make farther hits matter -- so the order in which (ray, node) pairs are looked at is free.  All
// pending pairs of the wave's 64 rays sit on one stack in LDS; a step pops up to 64 of them, one per
// lane, whichever ray they belong to.  A wave whose rays need 3, 40 and 0 steps therefore takes
// ceil(43 / 64) steps per level instead of 40, and a single expensive ray is searched by all 64
// lanes: its latency is the depth of the tree, not the number of nodes it touches.
//
// Stack words: [ray lane : 6][payload : 26]; node entries (payload = node index) grow up from
// word 0, face-block entries (payload = first face * 8 + count) grow down from the top, so that a
// step pops entries of one kind.  A node step pops k <= 64 entries and pushes at most 4k; k is
// chosen so that `reserve` = 3 * depth + 3 + kMeshFaceSlack words stay free afterwards, or 1 when
// they would not.  Popping one node at a time is a depth-first search: above the level it started
// from the node end never holds more than 3 * depth entries, and the face end at most 67 (it is
// drained as soon as it holds 64, and a node adds at most 4), so from a state with `reserve` free
// words at least 4 stay free and the stack cannot overflow whatever the mesh.  (Should it ever, the
// search is abandoned and counters[2] reports it: no out-of-range access either way.)
__device__ __forceinline__ int lane_rank(unsigned long long mask) {  // set bits below my lane
  return (int)__builtin_amdgcn_mbcnt_hi((unsigned)(mask >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mask, 0u));
}
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  __builtin_amdgcn_wave_barrier();
}
__device__ __forceinline__ float ubyte_f32(uint32_t x, int byte) { return (float)((x >> (8 * byte)) & 0xffu); }

// Distance slack of the search boxes.  The binary32 Moller-Trumbore test can accept a ray that
// misses the exact triangle by about 4e-7 |o - p0| / sin(smallest angle) (the error of
// dot(tvec, pvec) / det); the padded, outward-quantised boxes cover a fixed margin, and every box is
// widened by this fraction of (|o|_inf + largest mesh coordinate) >= |o - p0|_inf on top of it.
#ifndef MESH_DIST_SLACK  // (a diagnostic build sets it to 0 to show what the far-face tests catch)
#define MESH_DIST_SLACK 0x1p-16f
#endif

// The ray in a node's grid: per axis the time per grid step (idq) and the constants of
// t_lo = qlo * idq + ka, t_hi = qhi * idq + kb for the child planes qlo - rho and qhi + rho.
struct NodeFrame {
  float ka[3], kb[3], idq[3];
};
__device__ __forceinline__ void node_frame(uint4 w0, float4 r0, float4 r2, float mag, NodeFrame &f) {
  const float delta = MESH_DIST_SLACK * (fmaxf(fmaxf(fabsf(r0.x), fabsf(r0.y)), fabsf(r0.z)) + mag);
  const float oo[3] = {r0.x, r0.y, r0.z}, ii[3] = {r2.x, r2.y, r2.z};
  const float org[3] = {__uint_as_float(w0.x), __uint_as_float(w0.y), __uint_as_float(w0.z)};
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const int ex = (int)(int8_t)((w0.w >> (8 * a)) & 0xffu);
    const float oq = ldexpf(oo[a] - org[a], -ex);
    const float rho = ldexpf(delta, -ex);
    // finite even for a clamped reciprocal on a coarse grid: |q - oq| >= rho > 0 keeps the product
    // away from 0 * inf, and med3 keeps it below infinity
    f.idq[a] = __builtin_amdgcn_fmed3f(ldexpf(ii[a], ex), -1e35f, 1e35f);
    f.ka[a] = -(oq + rho) * f.idq[a];
    f.kb[a] = -(oq - rho) * f.idq[a];
  }
}
__device__ __forceinline__ bool child_box_hit(const NodeFrame &f, float qlx, float qly, float qlz, float qhx, float qhy,
                                              float qhz, float lo0, float hi0) {
  const float ql[3] = {qlx, qly, qlz}, qh[3] = {qhx, qhy, qhz};
  float en = -INFINITY, le = INFINITY;
#pragma unroll
  for (int a = 0; a < 3; a++) {
    const float tl = __builtin_fmaf(ql[a], f.idq[a], f.ka[a]);
    const float th = __builtin_fmaf(qh[a], f.idq[a], f.kb[a]);
    en = fmaxf(en, fminf(tl, th));
    le = fminf(le, fmaxf(tl, th));
  }
  const float lo = fmaxf(lo0, __builtin_fmaf(-fabsf(en), 1e-5f, en));
  const float hi = fminf(hi0, __builtin_fmaf(fabsf(le), 1e-5f, le));
  return lo <= hi;
}

// Record a face that passed the triangle test with parameter t in its ray's candidate list (`rr` =
// the ray record).  Same leaf: the smaller t wins, the higher reference index among equal t (what an
// in-order scan with `t <= t_to` keeps, bvh.cuh:127-134).  A full list keeps the leaves that come
// first in visiting order and moves `cut` down to the first leaf it had to leave to the next pass.
__device__ __forceinline__ void hit_list_insert(const SceneDev &sc, int *rr, uint32_t code, int face, int orig,
                                                float t) {
  const int cnt = rr[12];
  const uint32_t cut = (uint32_t)rr[13], lo_code = (uint32_t)rr[14];
  if (!(code >= lo_code && code < cut)) return;
  int found = -1, jmax = 0;
  uint32_t cmax = 0u;
#pragma unroll
  for (int j = 0; j < kHitSlots; j++) {
    if (j < cnt) {
      const uint32_t cj = (uint32_t)rr[16 + j * kHitWords];
      if (cj == code) found = j;
      if (cj >= cmax) cmax = cj, jmax = j;
    }
  }
  int slot = -1;
  if (found >= 0) {
    const float tj = __int_as_float(rr[16 + found * kHitWords + 2]);
    bool better = t < tj;
    if (t == tj) better = orig > sc.faces[rr[16 + found * kHitWords + 1]].orig;
    if (better) slot = found;
  } else if (cnt < kHitSlots) {
    slot = cnt;
    rr[12] = cnt + 1;
  } else if (code > cmax) {
    rr[13] = (int)code;  // this leaf and everything after it: next pass
  } else {
    rr[13] = (int)cmax;  // drop the last listed leaf instead
    slot = jmax;
  }
  if (slot >= 0) {
    rr[16 + slot * kHitWords] = (int)code;
    rr[16 + slot * kHitWords + 1] = face;
    rr[16 + slot * kHitWords + 2] = __float_as_int(t);
  }
}

// Diagnostic build only (-DRTMI_STATS, tools/mesh_stats.sh): wave-level step counts of the search.
#ifdef RTMI_STATS
struct MeshStats {
  unsigned searches, node_steps, face_steps, nodes_popped, blocks_popped, insert_rounds, steps_hist[6];
  // shader cycles (s_memtime) of this wave: [0] sample bookkeeping + camera ray, [1] world list before the mesh,
  // [2] mesh search, [3] replay, [4] shading; of the node steps: [5] pop + node/ray fetch, [6] box tests, [7] pushes;
  // [8] face steps incl. inserts
  unsigned long long cull_bits, cull_rays, cull_iters;  // culled list scan: candidate pairs, rays, wave iterations
  unsigned long long calib;  // two stamps back to back, once per search: what a stamp costs
  unsigned long long cyc[11];  // [9] search setup before the first step, [10] between steps (loop control)
};
__device__ unsigned long long g_wave_stats[16384][16];  // per wave: life, cyc[0..8], wave_queries, node_steps, face_steps
__device__ __forceinline__ unsigned long long stat_real() {
  __builtin_amdgcn_sched_barrier(0);
  unsigned long long t;
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#if RTMI_STATS == 9  // counts and wave lifetimes only: no stamps inside the loop, timing as in the product build
__device__ __forceinline__ unsigned long long stat_now() { return 0ull; }
#else
__device__ __forceinline__ unsigned long long stat_now() { return stat_real(); }
#endif
#define RTMI_STAT(x) x
#if RTMI_STATS == 2
#define RTMI_STAT2(x) x  // per-step stamps (a stamp costs several hundred cycles: they distort what they measure)
#else
#define RTMI_STAT2(x)
#endif
#if RTMI_STATS == 3
#define RTMI_STAT3(x) x  // replay sections instead of the per-step stamps: [5] order + offsets, [6] (a), [7] (b), [8] (c)
#else
#define RTMI_STAT3(x)
#endif
#else
#define RTMI_STAT(x)
#define RTMI_STAT2(x)
#define RTMI_STAT3(x)
#endif

// One search pass for the lanes with `need`: afterwards every such lane's record holds, per
// reference leaf with lo_code <= code < cut, the best face with t_from <= t <= bt_to.
template <typename T, bool DT>
__device__ __forceinline__ void mesh_search(const SceneDev &sc, int sub_root, const BvhNode *top, float mag, int *wl,
                                            bool need, V3 o, V3 d, V3 inv_d, T bt_to, uint32_t lo_code,
                                            unsigned long long *overflow
#ifdef RTMI_STATS
                                            , MeshStats &st
#endif
) {
  const int lane = (int)(threadIdx.x & 63u);
  RTMI_STAT(st.searches++; unsigned my_steps = 0; const unsigned long long tset0 = stat_now();)
  int *stack = wl + 64 * kMeshRayWords;
  const float lo0 = T_FROM_F * 0.999f;
  if (need) {
    int *rr = wl + lane * kMeshRayWords;
    int w3 = 0, w7 = 0;
    if (DT) {
      const double td = (double)bt_to;
      w3 = __double2loint(td), w7 = __double2hiint(td);
    } else {
      w3 = __float_as_int((float)bt_to);
    }
    *reinterpret_cast<int4 *>(rr + 0) = make_int4(__float_as_int(o.x), __float_as_int(o.y), __float_as_int(o.z), w3);
    *reinterpret_cast<int4 *>(rr + 4) = make_int4(__float_as_int(d.x), __float_as_int(d.y), __float_as_int(d.z), w7);
    *reinterpret_cast<int4 *>(rr + 8) = make_int4(__float_as_int(inv_d.x), __float_as_int(inv_d.y), __float_as_int(inv_d.z),
                                                  __float_as_int((float)bt_to * 1.0001f + 1e-6f));
    *reinterpret_cast<int4 *>(rr + 12) = make_int4(0, (int)kCodeNone, (int)lo_code, 0);
  }
  const unsigned long long nm = __ballot(need);
  const int reserve = sc.sub_reserve;
  int sn = 0, sf = 0;
  if (__popcll(nm) <= kTopRays) {
    // Few rays (the tail of a frame, or a wave that holds outlier pixels): the levels below the root,
    // where a step has next to nothing to do, are skipped.  Lane l holds sub-tree l of the mesh's top
    // table; one ray at a time is tested against all of them at once and the sub-trees it touches go
    // onto the stack.  (Boxes as conservative as the node boxes they stand for: scene.hip.)
    const BvhNode te = top[lane];
    const float far = (float)bt_to * 1.0001f + 1e-6f;
    for (unsigned long long m = nm; m != 0ull; m &= m - 1ull) {
      const int rl = __builtin_ctzll(m);  // wave-uniform
      if (kMeshStackWords - sn - sf - reserve < kTopEntries) {  // no room for a whole table: start this ray at the root
        if (lane == 0) stack[sn] = (rl << 26) | sub_root;
        sn++;
        continue;
      }
      const V3 ro = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.x), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.y), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(o.z), rl)));
      const V3 ri = mk(__int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_d.x), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_d.y), rl)),
                       __int_as_float(__builtin_amdgcn_readlane(__float_as_int(inv_d.z), rl)));
      const float rfar = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(far), rl));
      const float delta = MESH_DIST_SLACK * (fmaxf(fmaxf(fabsf(ro.x), fabsf(ro.y)), fabsf(ro.z)) + mag);
      const bool hit = te.left != -1 && slab_touch(te, delta, ro, ri, lo0, rfar);
      const bool pn = hit && te.left >= 0, pf = hit && te.left < 0;
      const unsigned long long mn_ = __builtin_amdgcn_ballot_w64(pn), mf_ = __builtin_amdgcn_ballot_w64(pf);
      const uint32_t owner_bits = (uint32_t)rl << 26;
      if (pn) stack[sn + lane_rank(mn_)] = (int)(owner_bits | (uint32_t)te.left);
      if (pf) stack[kMeshStackWords - 1 - sf - lane_rank(mf_)] = (int)(owner_bits | (uint32_t)(-(te.left + 1)));
      sn += __popcll(mn_);
      sf += __popcll(mf_);
    }
  } else {
    if (need) stack[lane_rank(nm)] = (lane << 26) | sub_root;
    sn = __popcll(nm);
  }
  wave_lds_fence();
  RTMI_STAT(unsigned long long tprev = stat_now(); st.cyc[9] += tprev - tset0;
            { const unsigned long long cb = stat_now(); st.calib += cb - tprev; tprev = cb; } (void)tprev;)
  while ((sn | sf) != 0) {
    if (sf >= 64 || sn == 0) {
      // ---------------------------------------------------------------- face step
      // Few blocks pending (the tail of a search, or a wave with one deep ray among 64): four lanes
      // per block, one face each -- one memory round trip and one triangle test deep.  Otherwise one
      // lane per block of up to four faces.
      const bool wide = sf <= 16, pair = !wide && sf <= 32;  // four / two lanes per block, or one
      const int kf = sf < 64 ? sf : 64;
      RTMI_STAT(st.face_steps++; st.blocks_popped += kf; my_steps++;)
      RTMI_STAT2(const unsigned long long tf0 = stat_now(); st.cyc[10] += tf0 - tprev;)
      const int slot = wide ? (lane >> 2) : pair ? (lane >> 1) : lane;
      const bool mine = slot < kf;
      const int fstride = pair ? 2 : 1;  // result bit j stands for face first + j * fstride
      int e = 0;
      if (mine) e = stack[kMeshStackWords - sf + slot];
      sf -= kf;
      wave_lds_fence();
      unsigned pend = 0u;  // bit j: face first + j passed the test ...
      float pt0 = 0.f, pt1 = 0.f, pt2 = 0.f, pt3 = 0.f;  // ... with this t ...
      float po0 = 0.f, po1 = 0.f, po2 = 0.f, po3 = 0.f, pc0 = 0.f, pc1 = 0.f, pc2 = 0.f, pc3 = 0.f;  // ... orig, leaf
      const int owner = (int)((unsigned)e >> 26), fcnt = e & 7;
      int first = (e >> 3) & (kMeshMaxFaces - 1);
      int *rr = wl + owner * kMeshRayWords;
      if (wide) {
        const int j = lane & 3;
        first += j;
        if (mine && j < fcnt) {
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r1 = *reinterpret_cast<const float4 *>(rr + 4);
          T t_to;
          if (DT) {
            t_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
          } else {
            t_to = (T)r0.w;
          }
          const float4 *fp4 = reinterpret_cast<const float4 *>(sc.faces + first);
          const float4 a = fp4[0], b = fp4[1], c = fp4[2];
          float t = 0.f, u = 0.f, v = 0.f;
          bool th = tri_test<T>(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), mk(r0.x, r0.y, r0.z),
                                mk(r1.x, r1.y, r1.z), t_to, t, u, v);
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 2
          th = th && t < -1.f;  // never
#endif
          if (th) pend = 1u, pt0 = t, po0 = c.y, pc0 = c.w;
        }
      } else if (pair) {
        first += lane & 1;  // this lane's faces: first, first + 2
        if (mine) {
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r1 = *reinterpret_cast<const float4 *>(rr + 4);
          T t_to;
          if (DT) {
            t_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
          } else {
            t_to = (T)r0.w;
          }
          const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
          const float4 *fp4 = reinterpret_cast<const float4 *>(sc.faces + first);
          const int left = fcnt - (lane & 1);  // faces first + 2 * fi exist for 2 * fi < left
          float4 q[6];
#pragma unroll
          for (int w = 0; w < 3; w++) q[w] = fp4[w], q[3 + w] = fp4[6 + w];  // `faces` carries 4 records of padding
#pragma unroll
          for (int fi = 0; fi < 2; fi++) {
            if (2 * fi < left) {
              const float4 a = q[fi * 3], b = q[fi * 3 + 1], c = q[fi * 3 + 2];
              float t = 0.f, u = 0.f, v = 0.f;
              bool th = tri_test<T>(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), ro, rd, t_to, t, u, v);
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 2
              th = th && t < -1.f;  // never
#endif
              if (th) {
                pend |= 1u << fi;
                if (fi == 0) pt0 = t, po0 = c.y, pc0 = c.w;
                if (fi == 1) pt1 = t, po1 = c.y, pc1 = c.w;
              }
            }
          }
        }
      } else if (mine) {
        const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r1 = *reinterpret_cast<const float4 *>(rr + 4);
        T t_to;
        if (DT) {
          t_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
        } else {
          t_to = (T)r0.w;
        }
        const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
        const float4 *fp4 = reinterpret_cast<const float4 *>(sc.faces + first);
#pragma unroll
        for (int half = 0; half < 2; half++) {
          if (half * 2 < fcnt) {
            float4 q[6];
#pragma unroll
            for (int w = 0; w < 6; w++) q[w] = fp4[half * 6 + w];  // `faces` carries 4 records of padding
#pragma unroll
            for (int fi = 0; fi < 2; fi++) {
              if (half * 2 + fi < fcnt) {
                const float4 a = q[fi * 3], b = q[fi * 3 + 1], c = q[fi * 3 + 2];
                float t = 0.f, u = 0.f, v = 0.f;
                bool th = tri_test<T>(mk(a.x, a.y, a.z), mk(a.w, b.x, b.y), mk(b.z, b.w, c.x), ro, rd, t_to, t, u, v);
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 2
                th = th && t < -1.f;  // never
#endif
                if (th) {
                  pend |= 1u << (half * 2 + fi);
                  if (half * 2 + fi == 0) pt0 = t, po0 = c.y, pc0 = c.w;
                  if (half * 2 + fi == 1) pt1 = t, po1 = c.y, pc1 = c.w;
                  if (half * 2 + fi == 2) pt2 = t, po2 = c.y, pc2 = c.w;
                  if (half * 2 + fi == 3) pt3 = t, po3 = c.y, pc3 = c.w;
                }
              }
            }
          }
        }
      }
      // hits go to their ray's list; two lanes with hits for the same ray take turns
      while (__ballot(pend != 0u) != 0ull) {
        RTMI_STAT(st.insert_rounds++;)
        const bool has = pend != 0u;
        if (has) rr[15] = lane;
        wave_lds_fence();
        if (has && rr[15] == lane) {
          const int j = __builtin_ctz(pend);
          const float t = j == 0 ? pt0 : j == 1 ? pt1 : j == 2 ? pt2 : pt3;
          const float fo = j == 0 ? po0 : j == 1 ? po1 : j == 2 ? po2 : po3;
          const float fc = j == 0 ? pc0 : j == 1 ? pc1 : j == 2 ? pc2 : pc3;
          hit_list_insert(sc, rr, (uint32_t)__float_as_int(fc), first + j * fstride, __float_as_int(fo), t);
          pend &= pend - 1u;
        }
        wave_lds_fence();
      }
      RTMI_STAT2(tprev = stat_now(); st.cyc[8] += tprev - tf0;)
    } else {
      // ---------------------------------------------------------------- node step
      // four lanes per entry with one child box each, two with two, or one with all four (see the face step)
      const bool wide = sn <= 16, pair = !wide && sn <= 32;
      const int kmax = wide ? 16 : pair ? 32 : 64;
      int k = (kMeshStackWords - sn - sf - reserve) / 3;
      k = k < 1 ? 1 : k;
      k = k > kmax ? kmax : k;
      k = k > sn ? sn : k;
      if (3 * k > kMeshStackWords - sn - sf) {  // cannot happen (see above); never write out of range
        if (lane == 0) atomicAdd(overflow, 1ull);
        break;
      }
      RTMI_STAT(st.node_steps++; st.nodes_popped += k; my_steps++;)
      RTMI_STAT2(const unsigned long long tn0 = stat_now(); unsigned long long tn1 = tn0; st.cyc[10] += tn0 - tprev;)
      const int slot = wide ? (lane >> 2) : pair ? (lane >> 1) : lane;
      const bool mine = slot < k;
      int e = 0;
      if (mine) e = stack[sn - 1 - slot];
      sn -= k;
      wave_lds_fence();
      const uint32_t owner_bits = (uint32_t)e & 0xfc000000u;
      const int owner = (int)((unsigned)e >> 26), idx = e & (kMeshMaxNodes - 1);
      const uint4 *np = reinterpret_cast<const uint4 *>(sc.qnodes + idx);
      const int *rr = wl + owner * kMeshRayWords;
      // children that were touched: nodes onto the node end, face blocks onto the face end
#define RTMI_PUSH_CHILD(H, C)                                                           \
  {                                                                                     \
    const bool pn = (H) && (C) >= 0, pf = (H) && (C) < 0;                               \
    const unsigned long long mn_ = __builtin_amdgcn_ballot_w64(pn), mf_ = __builtin_amdgcn_ballot_w64(pf); \
    if (pn) stack[sn + lane_rank(mn_)] = (int)(owner_bits | (uint32_t)(C));             \
    if (pf) stack[kMeshStackWords - 1 - sf - lane_rank(mf_)] = (int)(owner_bits | (uint32_t)(-((C) + 1))); \
    sn += __popcll(mn_);                                                                \
    sf += __popcll(mf_);                                                                \
  }
      if (pair) {
        bool ha = false, hb = false;
        int ca = -1, cb = -1;
        if (mine) {
          const int c = lane & 1;  // this lane's children: c, c + 2
          const uint4 w0 = np[0], w1 = np[1];
          const uint2 w2 = *reinterpret_cast<const uint2 *>(np + 2);
          ca = reinterpret_cast<const int *>(np + 3)[c], cb = reinterpret_cast<const int *>(np + 3)[c + 2];
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r2 = *reinterpret_cast<const float4 *>(rr + 8);
          NodeFrame nf;
          node_frame(w0, r0, r2, mag, nf);
          const int sa = 8 * c, sb = 8 * c + 16;
          const bool ba = child_box_hit(nf, (float)((w1.x >> sa) & 0xffu), (float)((w1.y >> sa) & 0xffu),
                                        (float)((w1.z >> sa) & 0xffu), (float)((w1.w >> sa) & 0xffu),
                                        (float)((w2.x >> sa) & 0xffu), (float)((w2.y >> sa) & 0xffu), lo0, r2.w);
          const bool bb = child_box_hit(nf, (float)((w1.x >> sb) & 0xffu), (float)((w1.y >> sb) & 0xffu),
                                        (float)((w1.z >> sb) & 0xffu), (float)((w1.w >> sb) & 0xffu),
                                        (float)((w2.x >> sb) & 0xffu), (float)((w2.y >> sb) & 0xffu), lo0, r2.w);
          ha = (ca != -1) & ba, hb = (cb != -1) & bb;
        }
        RTMI_PUSH_CHILD(ha, ca)
        RTMI_PUSH_CHILD(hb, cb)
        wave_lds_fence();
      } else if (wide) {
        bool hit = false;
        int child = -1;
        if (mine) {
          const int c = lane & 3;
          const uint4 w0 = np[0], w1 = np[1];
          const uint2 w2 = *reinterpret_cast<const uint2 *>(np + 2);
          child = reinterpret_cast<const int *>(np + 3)[c];
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r2 = *reinterpret_cast<const float4 *>(rr + 8);
          RTMI_STAT2(tn1 = stat_now();)
          NodeFrame nf;
          node_frame(w0, r0, r2, mag, nf);
          const int sh = 8 * c;
          // (tested whether or not the slot is used: the child word must not gate the other loads)
          const bool bh = child_box_hit(nf, (float)((w1.x >> sh) & 0xffu), (float)((w1.y >> sh) & 0xffu),
                                        (float)((w1.z >> sh) & 0xffu), (float)((w1.w >> sh) & 0xffu),
                                        (float)((w2.x >> sh) & 0xffu), (float)((w2.y >> sh) & 0xffu), lo0, r2.w);
          hit = (child != -1) & bh;
        }
        RTMI_STAT2(const unsigned long long tn2 = stat_now(); st.cyc[5] += tn1 - tn0; st.cyc[6] += tn2 - tn1;)
        const bool pn = hit && child >= 0, pf = hit && child < 0;
        const unsigned long long mn_ = __builtin_amdgcn_ballot_w64(pn), mf_ = __builtin_amdgcn_ballot_w64(pf);
        if (pn) stack[sn + lane_rank(mn_)] = (int)(owner_bits | (uint32_t)child);
        if (pf) stack[kMeshStackWords - 1 - sf - lane_rank(mf_)] = (int)(owner_bits | (uint32_t)(-(child + 1)));
        sn += __popcll(mn_);
        sf += __popcll(mf_);
        wave_lds_fence();
        RTMI_STAT2(tprev = stat_now(); st.cyc[7] += tprev - tn2;)
      } else {
        bool h0 = false, h1 = false, h2 = false, h3 = false;
        int c0 = -1, c1 = -1, c2 = -1, c3 = -1;
        if (mine) {
          const uint4 w0 = np[0], w1 = np[1], w2 = np[2], w3 = np[3];
          const float4 r0 = *reinterpret_cast<const float4 *>(rr + 0), r2 = *reinterpret_cast<const float4 *>(rr + 8);
          RTMI_STAT2(tn1 = stat_now();)
          NodeFrame nf;
          node_frame(w0, r0, r2, mag, nf);
          const uint32_t qlo[3] = {w1.x, w1.y, w1.z}, qhi[3] = {w1.w, w2.x, w2.y};
          const int cch[4] = {(int)w3.x, (int)w3.y, (int)w3.z, (int)w3.w};
          bool hh[4];
#pragma unroll
          for (int c = 0; c < 4; c++)
            hh[c] = cch[c] != -1 && child_box_hit(nf, ubyte_f32(qlo[0], c), ubyte_f32(qlo[1], c), ubyte_f32(qlo[2], c),
                                                  ubyte_f32(qhi[0], c), ubyte_f32(qhi[1], c), ubyte_f32(qhi[2], c), lo0, r2.w);
          h0 = hh[0], h1 = hh[1], h2 = hh[2], h3 = hh[3];
          c0 = cch[0], c1 = cch[1], c2 = cch[2], c3 = cch[3];
        }
        RTMI_STAT2(const unsigned long long tn2 = stat_now(); st.cyc[5] += tn1 - tn0; st.cyc[6] += tn2 - tn1;)
        RTMI_PUSH_CHILD(h0, c0)
        RTMI_PUSH_CHILD(h1, c1)
        RTMI_PUSH_CHILD(h2, c2)
        RTMI_PUSH_CHILD(h3, c3)
#undef RTMI_PUSH_CHILD
        wave_lds_fence();
        RTMI_STAT2(tprev = stat_now(); st.cyc[7] += tprev - tn2;)
      }
    }
  }
  RTMI_STAT(st.steps_hist[my_steps <= 1 ? 0 : my_steps <= 4 ? 1 : my_steps <= 8 ? 2 : my_steps <= 12 ? 3 : my_steps <= 20 ? 4 : 5]++;)
}

struct Hit {
  bool ok;
  float t;        // float(record.t)
  uint32_t win;   // winner id
  int32_t aux;    // BVH record index of the winner
  float u, v;     // raw barycentrics of the winning triangle
};

// ================================================================== closest hit
// HitableList::Hit (hitable_list.cu:7-25) over the flattened world.  A nested
// Parallelepiped list is equivalent to its six parallelograms inlined at its
// position (DESIGN.md "List flattening").
// `live`: mesh variants are entered by ALL lanes of the wave (the mesh search borrows idle
// lanes); a lane that is not tracing passes live = false and gets an unused result.  The other
// variants are only entered by tracing lanes and pass true.
template <uint32_t F>
__device__ __forceinline__ Hit closest_hit(const SceneDev &sc, const BvhNode *s_nodes, int lds_nodes, const int *s_paths,
                                           int lds_paths, const float4 *s_pairs, int *ll, int *wl, unsigned long long *overflow, V3 o,
                                           V3 d, bool live
#ifdef RTMI_STATS
                                           , MeshStats &st
#endif
) {
  constexpr bool DT = (F & F_SPHERE) != 0;
  typedef typename TSel<DT>::type T;
  bool ok = false;
  T t_to = (T)INFINITY;
  uint32_t win = ID_NONE;
  int32_t aux = 0;
  float bu = 0.f, bv = 0.f;

  double sa = 0.0, sa2 = 0.0;
  float saf = 0.f;
  if (F & F_SPHERE) {
    float la = len3(d);       // sphere.cu:13: pow(length(dir), 2) in float, then widened
    saf = la * la;
    sa = (double)saf;
    sa2 = 2 * sa;
  }

  const int lane = (int)(threadIdx.x & 63u);
  // the ray as the culled list scan wants it: 1/d (the hardware reciprocal will do: the test is conservative by
  // a margin of 1e-5, not 1e-7), and -(o +- delta)/d per axis, delta = the distance slack of the mesh search
  V3 cull_inv = splat(0.f), cull_klo = splat(0.f), cull_khi = splat(0.f);
  if ((F & F_TRIS) && s_pairs != nullptr) {
    const float ix = __builtin_amdgcn_rcpf(d.x), iy = __builtin_amdgcn_rcpf(d.y), iz = __builtin_amdgcn_rcpf(d.z);
    cull_inv = mk(fabsf(d.x) < 1e-30f ? copysignf(1e30f, d.x) : ix, fabsf(d.y) < 1e-30f ? copysignf(1e30f, d.y) : iy,
                  fabsf(d.z) < 1e-30f ? copysignf(1e30f, d.z) : iz);
    const float delta = MESH_DIST_SLACK * (fmaxf(fmaxf(fabsf(o.x), fabsf(o.y)), fabsf(o.z)) + sc.list_mag);
    cull_klo = mk(-(o.x + delta) * cull_inv.x, -(o.y + delta) * cull_inv.y, -(o.z + delta) * cull_inv.z);  // lower planes, moved out
    cull_khi = mk(-(o.x - delta) * cull_inv.x, -(o.y - delta) * cull_inv.y, -(o.z - delta) * cull_inv.z);  // upper planes
  }

  for (int ri = 0; ri < sc.n_runs; ri++) {
    const i32x4 rv = load_run(sc.runs, ri);
    Run run;
    run.kind = rv[0], run.first = rv[1], run.count = rv[2], run.pad = 0;
    if (live && run.kind == RUN_SKY) {
      // sky.cu:18-27: t = 1e9; t_from <= 1e9 always holds
      const T ts = (T)1e9f;
      bool hit = ts <= t_to;
      bool acc = hit && (!ok || ts < t_to);
      ok = ok || acc;
      t_to = acc ? ts : t_to;
      win = acc ? make_id(RUN_SKY, 0) : win;
    }
    if ((F & F_TRIS) && run.kind == RUN_TRIS && s_pairs != nullptr) {
      // Culled scan (DESIGN.md "World-list scan").  The reference tests every entry of the list against
      // every ray (hitable_list.cu:11-22); what it RETURNS only depends on the entries whose test can
      // succeed, visited in list order.  Lanes of a wave carry unrelated rays, so no entry can be
      // skipped for the whole wave -- but each lane can skip its own: (1) every pair's padded bounds
      // (one s_load_dwordx8, wave-uniform) against the lane's ray: a slab test, 27 instructions
      // instead of the 140 of two triangle tests, builds a bit mask of the pairs this ray comes near;
      // (2) while any lane has bits left, each lane takes ITS next pair -- a different one per lane,
      // corners gathered from LDS -- and runs the reference's two triangle tests on it.  A lane visits
      // its pairs in list order with its own running t_to, so acceptance and ties are as in the full
      // scan; a pair outside the mask cannot pass the triangle test (the bounds carry the same padding
      // and distance slack as the mesh search boxes).
      const int pair0 = run.first >> 1;
      const float lo0 = T_FROM_F * 0.999f;
      for (int c0 = 0; c0 < run.count; c0 += 32) {
        const int nc = run.count - c0 < 32 ? run.count - c0 : 32;
        const float hi0 = (float)t_to * 1.0001f + 1e-6f;
        uint32_t mask = 0u;
        RTMI_STAT2(const unsigned long long tc0 = stat_now();)
        f32x8 nxt = load_pair_box(sc.pair_boxes, pair0 + c0);
        for (int i = 0; i < nc; i++) {
          __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): see the plain scan below
          __builtin_amdgcn_sched_barrier(0);
          const f32x8 bx = nxt;
          nxt = load_pair_box(sc.pair_boxes, pair0 + c0 + i + 1);  // (one inert record of padding at the end)
          __builtin_amdgcn_sched_barrier(0);
          // t = (plane -+ delta - o) / d as one FMA per plane: plane * (1/d) - (o +- delta) * (1/d).  The rounding of
          // the two products is an error of ~6e-8 of the plane's coordinate in space, far inside delta.
          const float t0x = __builtin_fmaf(bx[0], cull_inv.x, cull_klo.x), t1x = __builtin_fmaf(bx[3], cull_inv.x, cull_khi.x);
          const float t0y = __builtin_fmaf(bx[1], cull_inv.y, cull_klo.y), t1y = __builtin_fmaf(bx[4], cull_inv.y, cull_khi.y);
          const float t0z = __builtin_fmaf(bx[2], cull_inv.z, cull_klo.z), t1z = __builtin_fmaf(bx[5], cull_inv.z, cull_khi.z);
          const float en = fmaxf(fmaxf(fminf(t0x, t1x), fminf(t0y, t1y)), fminf(t0z, t1z));
          const float le = fminf(fminf(fmaxf(t0x, t1x), fmaxf(t0y, t1y)), fmaxf(t0z, t1z));
          mask |= fmaxf(lo0, en) <= fminf(hi0, le) ? 1u << i : 0u;
        }
        if (!live) mask = 0u;  // a lane without a ray of its own only helps
        RTMI_STAT2(const unsigned long long tc1 = stat_now(); st.cyc[5] += tc1 - tc0;)
        RTMI_STAT(st.cull_bits += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(true)) * 0u; { unsigned pc = __builtin_popcount(mask); for (int off = 32; off > 0; off >>= 1) pc += __shfl_down(pc, off); st.cull_bits += __builtin_amdgcn_readfirstlane(pc); } st.cull_rays += (unsigned)__popcll(__builtin_amdgcn_ballot_w64(true));)
        {
          // ---- the candidates of all 64 rays are worked off by all 64 lanes.  A ray comes near 2.2 pairs on
          // average but the unluckiest of 64 near 6, and a lane-by-lane loop runs as long as that one.  So
          // (a) every lane writes its ray and one task per candidate pair to LDS (offsets: prefix sum of the
          // candidate counts by bit planes), (b) lane l takes task l, l + 64, ...: reads that ray and that
          // pair's corners and runs BOTH triangle tests against the ray's t_to at the start of the chunk,
          // (c) every lane folds the results of its own candidates in list order with its running t_to:
          // a test that passed against the older, larger t_to passes now iff its t <= the current one,
          // which is the only place t_to enters the test (utils.cu:74).
          constexpr int RW = (F & F_TEX) ? 6 : 2;  // result words per task: t of the two triangles (+ their u, v)
          int *tasks = ll + 64 * 8, *results = ll + 64 * 8 + kListTasks;
          const int cnt = __builtin_popcount(mask);
          int base = 0;
#pragma unroll
          for (int bit = 0; bit < 6; bit++)
            base += lane_rank(__builtin_amdgcn_ballot_w64(((cnt >> bit) & 1) != 0)) << bit;
          if (cnt != 0) {
            int *rr = ll + lane * 8;
            int w3 = 0, w7 = 0;
            if (DT) {
              const double td = (double)t_to;
              w3 = __double2loint(td), w7 = __double2hiint(td);
            } else {
              w3 = __float_as_int((float)t_to);
            }
            *reinterpret_cast<int4 *>(rr) = make_int4(__float_as_int(o.x), __float_as_int(o.y), __float_as_int(o.z), w3);
            *reinterpret_cast<int4 *>(rr + 4) = make_int4(__float_as_int(d.x), __float_as_int(d.y), __float_as_int(d.z), w7);
          }
          bool todo = cnt != 0;
          while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
            RTMI_STAT(st.cull_iters++;)
            const int lo_t = __builtin_amdgcn_readlane(base, __builtin_ctzll(__builtin_amdgcn_ballot_w64(todo)));
            const bool now = todo && base + cnt - lo_t <= kListTasks;
            const int n_now = __builtin_amdgcn_readlane(base + cnt, 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(now))) - lo_t;
            if (now) {  // (a)
              int k = base - lo_t;
              for (uint32_t m = mask; m != 0u; m &= m - 1u) tasks[k++] = (lane << 5) | __builtin_ctz(m);
            }
            wave_lds_fence();
            for (int t0 = 0; t0 < n_now; t0 += 64) {  // (b)
              const int ti = t0 + lane;
              if (ti < n_now) {
                const int w = tasks[ti];
                const int *orr = ll + (w >> 5) * 8;
                const float4 r0 = *reinterpret_cast<const float4 *>(orr), r1 = *reinterpret_cast<const float4 *>(orr + 4);
                T t0_to;
                if (DT) {
                  t0_to = (T)__hiloint2double(__float_as_int(r1.w), __float_as_int(r0.w));
                } else {
                  t0_to = (T)r0.w;
                }
                const V3 ro = mk(r0.x, r0.y, r0.z), rd = mk(r1.x, r1.y, r1.z);
                const float4 *pp = s_pairs + (size_t)(pair0 + c0 + (w & 31)) * 4;
                const float4 qa = pp[0], qb = pp[1], qc = pp[2], qd = pp[3];
                const V3 p0 = mk(qa.x, qa.y, qa.z), p1 = mk(qa.w, qb.x, qb.y), p2 = mk(qb.z, qb.w, qc.x), p3 = mk(qc.y, qc.z, qc.w);
                float ta = 0.f, ua = 0.f, va = 0.f, tb = 0.f, ub = 0.f, vb = 0.f;
                const V3 e1 = p1 - p0, e2 = p2 - p0;  // utils.cu:54-55, the subtractions scene.hip: make_tri does
                const bool hit_a = tri_test_flat<T>(p0, e1, e2, cross3(rd, e2), ro, rd, t0_to, ta, ua, va);
                bool hit_b = false;
                if (__float_as_int(qd.x) & PAIR_SECOND) {
                  const V3 e1b = p2 - p1, e2b = p3 - p1;
                  hit_b = tri_test_flat<T>(p1, e1b, e2b, cross3(rd, e2b), ro, rd, t0_to, tb, ub, vb);
                }
                int *res = results + ti * RW;
                res[0] = hit_a ? __float_as_int(ta) : (int)0xffffffff;  // (a NaN pattern no t can have)
                res[1] = hit_b ? __float_as_int(tb) : (int)0xffffffff;
                if (F & F_TEX) {
                  res[2] = __float_as_int(ua), res[3] = __float_as_int(va), res[4] = __float_as_int(ub), res[5] = __float_as_int(vb);
                }
              }
            }
            wave_lds_fence();
            RTMI_STAT2(const unsigned long long tc2 = stat_now();)
            if (now) {  // (c)
              int k = base - lo_t;
              for (uint32_t m = mask; m != 0u; m &= m - 1u, k++) {
                const int tri = run.first + 2 * (c0 + __builtin_ctz(m));
                const int *res = results + k * RW;
                const int ia = res[0], ib = res[1];
                const float ta = __int_as_float(ia), tb = __int_as_float(ib);
                // parallelogram.cu:25-33 with the t_to of THIS moment: the first triangle, else the second
                const bool hit_a = ia != (int)0xffffffff && (T)ta <= t_to;
                const bool hit_b = !hit_a && ib != (int)0xffffffff && (T)tb <= t_to;
                const float t = hit_a ? ta : tb;
                const bool acc = (hit_a || hit_b) && (!ok || (T)t < t_to);
                ok = ok || acc;
                t_to = acc ? (T)t : t_to;
                win = acc ? make_id(RUN_TRIS, tri + (hit_a ? 0 : 1)) : win;
                if (F & F_TEX) {
                  bu = acc ? __int_as_float(hit_a ? res[2] : res[4]) : bu;
                  bv = acc ? __int_as_float(hit_a ? res[3] : res[5]) : bv;
                }
              }
              todo = false;
            }
            wave_lds_fence();
            RTMI_STAT2(st.cyc[7] += stat_now() - tc2;)
          }
        }
        RTMI_STAT2(st.cyc[6] += stat_now() - tc1;)  // (a) + (b) + (c); [7] is (c) alone
      }
    } else if ((F & F_TRIS) && live && run.kind == RUN_TRIS) {
      // Plain scan (lists too long for the LDS staging of the culled one).
      // Records come in (first, second) pairs: a Parallelogram's two triangles, or a lone
      // Triangle followed by an inert record.  Two SGPR buffers ping-pong: while record A
      // is tested the fetch of B is in flight, and vice versa.  Scalar-memory waits are
      // all-or-nothing (lgkmcnt counts SMEM out of order), so the order is pinned: wait
      // for the buffer about to be used, only then issue the next fetch, then test.
      const HotTri *base = sc.tris + run.first;
      f32x16 A = load_hot_tri(base, 0);
      for (int i = 0; i < run.count; i++) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): A has landed
        __builtin_amdgcn_sched_barrier(0);
        const f32x16 B = load_hot_tri(base, 2 * i + 1);
        __builtin_amdgcn_sched_barrier(0);
        float t = 0.f, u = 0.f, v = 0.f;
        const V3 pv_a = cross3(d, mk(A[6], A[7], A[8]));
        const bool hit_a = tri_test_flat<T>(mk(A[0], A[1], A[2]), mk(A[3], A[4], A[5]), mk(A[6], A[7], A[8]), pv_a, o, d,
                                            t_to, t, u, v);
        {
          bool acc = hit_a && (!ok || (T)t < t_to);
          ok = ok || acc;
          t_to = acc ? (T)t : t_to;
          win = acc ? make_id(RUN_TRIS, run.first + 2 * i) : win;
          if (F & F_TEX) {
            bu = acc ? u : bu;
            bv = acc ? v : bv;
          }
        }
        __builtin_amdgcn_s_waitcnt(0xc07f);  // B has landed
        __builtin_amdgcn_sched_barrier(0);
        A = load_hot_tri(base, 2 * i + 2);  // next pair (or the inert padding pair)
        __builtin_amdgcn_sched_barrier(0);
        if (__float_as_int(B[13]) & TRI_SECOND) {  // wave-uniform: a lone Triangle has no second record
          // parallelogram.cu:33: the second triangle is tried only when the first missed
          V3 pv_b = pv_a;
          if (!(__float_as_int(B[13]) & TRI_SAME_E2)) pv_b = cross3(d, mk(B[6], B[7], B[8]));  // wave-uniform
          bool hit_b = tri_test_flat<T>(mk(B[0], B[1], B[2]), mk(B[3], B[4], B[5]), mk(B[6], B[7], B[8]), pv_b, o, d,
                                        t_to, t, u, v);
          hit_b = hit_b && !hit_a;
          bool acc = hit_b && (!ok || (T)t < t_to);
          ok = ok || acc;
          t_to = acc ? (T)t : t_to;
          win = acc ? make_id(RUN_TRIS, run.first + 2 * i + 1) : win;
          if (F & F_TEX) {
            bu = acc ? u : bu;
            bv = acc ? v : bv;
          }
        }
      }
    }
    if ((F & F_SPHERE) && live && run.kind == RUN_SPHERE) {
      f32x8 nxt = load_sphere(sc.spheres, run.first);
      for (int i = 0; i < run.count; i++) {
        __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): see the triangle loop
        __builtin_amdgcn_sched_barrier(0);
        const f32x8 cur = nxt;
        nxt = load_sphere(sc.spheres, run.first + i + 1);
        __builtin_amdgcn_sched_barrier(0);
        const double r2 = __hiloint2double(__float_as_int(cur[7]), __float_as_int(cur[6]));
        V3 oc = o - mk(cur[0], cur[1], cur[2]);
        const float bf = 2.0f * dot3(d, oc);
        {
          // Wave-level cull.  The reference's discriminant is b*b - 4*a*c in double from the
          // binary32 values b, a = |d|^2, |oc|^2 (sphere.cu:13-17).  The same expression in
          // binary32 (with dot(oc,oc) for |oc|^2, i.e. without the square root) is off by at
          // most a few ulp of its largest term; when it is below -1e-5 of the terms' magnitude
          // on EVERY lane the exact discriminant is negative on every lane, no lane can hit, and
          // the binary64 part is skipped.  Lanes of a wave carry unrelated rays, but a small
          // sphere is in the way of few of them.  (NaN/inf compare false: not skipped.)
          const float oc2 = dot3(oc, oc), r2f = (float)r2;
          const float disc_f = bf * bf - 4.0f * saf * (oc2 - r2f);
          const float mag = bf * bf + 4.0f * saf * (oc2 + r2f);
          if (!__any(!(disc_f < -1e-5f * mag))) continue;
        }
        double b = (double)bf;
        float lc = len3(oc);
        double c = (double)(lc * lc) - r2;
        double disc = b * b - 4 * sa * c;
        bool hit = false;
        double t = 0.0;
        if (!(disc < 0)) {
          double sq = sqrt(disc);
          t = (-b - sq) / sa2;
          hit = (1e-3 <= t && t <= (double)t_to);
          if (!hit) {
            t = (-b + sq) / sa2;
            hit = (1e-3 <= t && t <= (double)t_to);
          }
        }
        bool acc = hit && (!ok || t < (double)t_to);
        ok = ok || acc;
        t_to = acc ? (T)t : t_to;
        win = acc ? make_id(RUN_SPHERE, run.first + i) : win;
      }
    }
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 3
    if (false) {
#else
    if ((F & F_BVH) && run.kind == RUN_BVH) {
#endif
      // BVH::Hit (bvh.cuh:123-183) answered without walking the reference's tree.
      //
      // What the reference computes: a depth-first walk, left subtree first, with a running
      // t_to; a child is entered iff AABB::Hit(child box, [t_from, t_to]) holds at that moment
      // (the root's box is never tested); an entered leaf scans its faces in order, accepts
      // every t_from <= t <= t_to and lowers t_to to it; the last acceptance is the answer.
      // Hence (DESIGN.md "Mesh queries"):
      //  * a leaf's contribution at running bound T is its best face -- smallest t, highest
      //    reference index among equal t -- provided that t <= T; it does not depend on T
      //    otherwise, nor on the order the faces are looked at;
      //  * leaves without a hit change nothing; whether their boxes were entered is irrelevant;
      //  * a box test is only ever needed on the root-to-leaf path of a leaf that holds a hit,
      //    and it sees the t_to left by the hit leaves before it in visiting order.
      // So: (1) one search of the mesh-wide 4-wide tree (mesh_search: by the whole wave, for all
      // its rays at once) collects, per reference leaf, the best face in the ray's small list
      // keyed by the leaf's path code; (2) each lane replays its listed leaves in visiting order,
      // evaluating the reference's exact box test on the path nodes not shared with the previously
      // replayed leaf.  If more leaves hold hits than the list has slots, the leaves beyond `cut`
      // are left to a further search pass.
      const V3 inv_d = mk(safe_inverse(d.x), safe_inverse(d.y), safe_inverse(d.z));
      int *rr = wl + lane * kMeshRayWords;
      for (int i = 0; i < run.count; i++) {
        BvhRec br;
        {
          const i32x8 bw = load_bvh_rec(sc.bvhs, run.first + i);  // wave-uniform: scalar load
          br.root = bw[0], br.mat = bw[1], br.has_uv = bw[2], br.face_base = bw[3], br.sub_root = bw[4];
          br.mag = __int_as_float(bw[5]);
          br.ref_depth = bw[6], br.path_base = bw[7];
        }
        T bt_to = t_to;
        bool bhit = false;
        int bface = 0;
        float fu = 0.f, fv = 0.f;
        // replay state: path code of the last replayed leaf and, left-aligned like the code,
        // one bit per level "that node of its path was entered"
        bool have_prev = false;
        uint32_t prev_code = 0u, entered = 0u;
        uint32_t lo_code = 0u;
        bool need = live;
        // All 64 lanes walk this loop together (lanes without a ray with need == false): the search
        // is the wave's.
        while (__ballot(need) != 0ull) {
          // ---- (1) search: best face per leaf with lo_code <= code < cut, t <= bt_to
          RTMI_STAT(const unsigned long long ts0 = stat_now();)
          mesh_search<T, DT>(sc, br.sub_root, sc.tops + (size_t)(run.first + i) * kTopEntries, br.mag, wl, need, o, d, inv_d,
                             bt_to, lo_code, overflow
#ifdef RTMI_STATS
                             , st
#endif
          );
          RTMI_STAT(const unsigned long long ts1 = stat_now(); st.cyc[2] += ts1 - ts0;)
          // ---- (2) replay the listed leaves in the reference's visiting order.  The box tests are
          // spread over the wave: AABB::Hit(box, [t_from, T]) is `crossing time <= T` with a crossing time
          // that does not depend on T (aabb_crossing_time), so (a) every lane with listed leaves writes one
          // word per leaf (its lane, the leaf's row in the mesh's path table) into the search's (now empty)
          // stack, (b) all 64 lanes work off the (leaf, level) pairs, whoever's they are: node from the path
          // table, its box, that ray, the crossing time, (c) each lane walks its leaves with the running
          // t_to, looking the crossing times up.  A lane whose leaves do not fit next to the others' waits
          // for the next round.
          RTMI_STAT(const unsigned long long tr0 = stat_now(); (void)tr0;)
          uint32_t cut = kCodeNone;
          int cnt = 0;
          int4 hs[kHitSlots];  // my entries: leaf, face, t
#pragma unroll
          for (int j = 0; j < kHitSlots; j++) hs[j] = make_int4((int)kCodeNone, 0, 0, 0);
          if (need) {
            const int4 head = *reinterpret_cast<const int4 *>(rr + 12);
            cnt = head.x, cut = (uint32_t)head.y;
            const int4 wa = *reinterpret_cast<const int4 *>(rr + 16), wb = *reinterpret_cast<const int4 *>(rr + 20),
                       wc = *reinterpret_cast<const int4 *>(rr + 24);
            const int4 e4[kHitSlots] = {make_int4(wa.x, wa.y, wa.z, 0), make_int4(wa.w, wb.x, wb.y, 0),
                                        make_int4(wb.z, wb.w, wc.x, 0), make_int4(wc.y, wc.z, wc.w, 0)};
#pragma unroll
            for (int j = 0; j < kHitSlots; j++)
              if (j < cnt && (uint32_t)e4[j].x < cut) hs[j] = e4[j];  // (the rest was pushed beyond `cut`: next pass)
          }
          // ... in visiting order = by ascending leaf ordinal (never kCodeNone): a five-exchange network
#define RTMI_ORDER(A, B)                                        \
  {                                                             \
    const bool sw = (uint32_t)hs[B].x < (uint32_t)hs[A].x;      \
    const int4 lo_ = sw ? hs[B] : hs[A], hi_ = sw ? hs[A] : hs[B]; \
    hs[A] = lo_, hs[B] = hi_;                                   \
  }
          RTMI_ORDER(0, 1) RTMI_ORDER(2, 3) RTMI_ORDER(0, 2) RTMI_ORDER(1, 3) RTMI_ORDER(1, 2)
#undef RTMI_ORDER
          int nleaf = 0;
#pragma unroll
          for (int j = 0; j < kHitSlots; j++) nleaf += (uint32_t)hs[j].x != kCodeNone ? 1 : 0;
          const int depth_r = br.ref_depth;
          const int log_d = depth_r < 8 ? 3 : depth_r < 16 ? 4 : 5;  // a leaf's row: its path code + the crossing times, 8, 16 or 32 words
          const int rows_max = (kMeshStackWords - 64) >> log_d;
          // exclusive prefix sum of nleaf (0..4) over the wave, bit plane by bit plane: no LDS round trips
          const int base = lane_rank(__builtin_amdgcn_ballot_w64((nleaf & 1) != 0)) +
                           2 * lane_rank(__builtin_amdgcn_ballot_w64((nleaf & 2) != 0)) +
                           4 * lane_rank(__builtin_amdgcn_ballot_w64((nleaf & 4) != 0));
          int *leaves = wl + 64 * kMeshRayWords;  // [64] one word per listed leaf of this round
          int *times = leaves + 64;                // rows of crossing times
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 4
          bool todo = false;  // (diagnostic: no replay, nearest listed hit)
#else
          bool todo = nleaf > 0 && depth_r > 0;
#endif
          RTMI_STAT3(unsigned long long tr1 = stat_now(); st.cyc[5] += tr1 - ts1;)
          while (__builtin_amdgcn_ballot_w64(todo) != 0ull) {
            const int lo = __builtin_amdgcn_readlane(base, __builtin_ctzll(__builtin_amdgcn_ballot_w64(todo)));
            const bool now = todo && base + nleaf - lo <= rows_max;
            const int n_rows = __builtin_amdgcn_readlane(base + nleaf, 63 - __builtin_clzll(__builtin_amdgcn_ballot_w64(now))) - lo;
            if (now) {  // (a)
#pragma unroll
              for (int k = 0; k < kHitSlots; k++)
                if (k < nleaf) leaves[base - lo + k] = (lane << 26) | hs[k].x;
            }
            wave_lds_fence();
            RTMI_STAT3(const unsigned long long tr2 = stat_now(); st.cyc[6] += tr2 - tr1;)
            // (b) crossing times, one (leaf, level) per lane and round
            const int n_now = n_rows << log_d;
            for (int t0 = 0; t0 < n_now; t0 += 64) {
              const int t = t0 + lane, lvl = t & ((1 << log_d) - 1);  // word 0 of a row: the leaf's path code
              if (t < n_now && lvl <= depth_r) {
                const int w = leaves[t >> log_d];
                const int pi = br.path_base + (w & (kMeshMaxNodes - 1)) * (depth_r + 1) + lvl;
                const int ni = pi < lds_paths ? s_paths[pi] : sc.leaf_paths[pi];
                float m = __int_as_float(lvl == 0 ? ni : (int)0xffffffff);  // marker: past the leaf
                if (lvl != 0 && ni >= 0) {
                  const int *orr = wl + (int)((unsigned)w >> 26) * kMeshRayWords;
                  const float4 r0 = *reinterpret_cast<const float4 *>(orr + 0), r1 = *reinterpret_cast<const float4 *>(orr + 4);
                  BvhNode nd;
                  if (ni < lds_nodes) {
                    nd = s_nodes[ni];
                  } else {
                    nd = sc.nodes[ni];
                  }
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 5
                  m = nd.mn[0] * r0.x + r1.x;  // (diagnostic: the loads without the arithmetic)
#else
                  m = aabb_crossing_time(nd, mk(r0.x, r0.y, r0.z), mk(r1.x, r1.y, r1.z));
#endif
                }
                times[t] = __float_as_int(m);
              }
            }
            wave_lds_fence();
            RTMI_STAT3(const unsigned long long tr3 = stat_now(); st.cyc[7] += tr3 - tr2;)
            // (c) the walk of bvh.cuh:123-158 over my leaves
            // (c) the walk of bvh.cuh:123-158 over my leaves, without branches: per leaf one bit per level,
            // left-aligned like the path code -- `valid` the levels of its path (a prefix), `pass` the levels whose
            // box is entered: as the previous leaf found it where the two paths coincide, by `crossing time <=
            // t_to` below.  The walk stops at the first level that is not entered; the leaf's faces count iff
            // there is none.
            if (__builtin_amdgcn_ballot_w64(now) != 0ull) {
#pragma unroll
              for (int k = 0; k < kHitSlots; k++) {
                if (__builtin_amdgcn_ballot_w64(now && k < nleaf) == 0ull) break;  // wave-uniform
                const bool mine = now && k < nleaf;
                const int *row = times + (mine ? (base - lo + k) << log_d : 0);
                uint32_t code = 0u, valid = 0u, below = 0u;
                for (int c = 0; c <= depth_r; c += 8) {  // wave-uniform trip count
                  const int4 ma = *reinterpret_cast<const int4 *>(row + c), mb = *reinterpret_cast<const int4 *>(row + c + 4);
                  const int mv[8] = {ma.x, ma.y, ma.z, ma.w, mb.x, mb.y, mb.z, mb.w};
                  if (c == 0) code = (uint32_t)ma.x;
#pragma unroll
                  for (int j = 0; j < 8; j++) {
                    const uint32_t bit = 0x80000000u >> ((c + j - 1) & 31);  // level c + j (>= 1)
                    const bool in_row = c + j >= 1 && c + j <= depth_r;
                    valid |= in_row && mv[j] != (int)0xffffffff ? bit : 0u;
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 1
                    below |= in_row ? bit : 0u;
#else
                    below |= in_row && (T)__int_as_float(mv[j]) <= bt_to ? bit : 0u;  // (the marker is a NaN: false)
#endif
                  }
                }
                const int shared = have_prev ? __clz((int)(prev_code ^ code)) : 0;  // decisions in common
                const uint32_t common = shared ? 0xffffffffu << (32 - shared) : 0u;
                const uint32_t pass = (entered & common) | (below & ~common);
                const uint32_t fail = valid & ~pass;
                const int upto = fail ? __clz((int)fail) : __popc(valid);  // levels entered before the walk stopped
                const uint32_t walked = upto ? 0xffffffffu << (32 - upto) : 0u;
                const uint32_t bits = (entered & common) | (walked & ~common);
                if (mine) {
                  have_prev = true;
                  prev_code = code;
                  entered = bits;
                  const T tj = (T)__int_as_float(hs[k].z);
                  if (fail == 0u && tj <= bt_to) {
                    bt_to = tj;
                    bhit = true;
                    bface = hs[k].y;
                  }
                }
              }
              if (now) todo = false;
            }
            wave_lds_fence();
            RTMI_STAT3(tr1 = stat_now(); st.cyc[8] += tr1 - tr3;)
          }
#if defined(RTMI_ABLATE) && RTMI_ABLATE == 4
          if (nleaf > 0) {
#else
          if (nleaf > 0 && depth_r == 0) {  // the root is the only leaf: nothing to test (its box never is)
#endif
#pragma unroll
            for (int k = 0; k < kHitSlots; k++) {
              if (k < nleaf) {
                const T tj = (T)__int_as_float(hs[k].z);
                if (tj <= bt_to) bt_to = tj, bhit = true, bface = hs[k].y;
              }
            }
          }
          need = need && cut != kCodeNone;  // leaves were deferred: search again from `cut` on
          lo_code = cut;
          RTMI_STAT(st.cyc[3] += stat_now() - ts1;)
        }
        if (bhit && (F & F_TEX)) {
          // barycentrics of the winner (the same binary32 operations as in the search)
          const FaceRec f = sc.faces[bface];
          float t = 0.f;
          const V3 fe2 = mk(f.e2[0], f.e2[1], f.e2[2]);
          (void)tri_test_flat<T>(mk(f.p0[0], f.p0[1], f.p0[2]), mk(f.e1[0], f.e1[1], f.e1[2]), fe2, cross3(d, fe2), o, d,
                                 bt_to, t, fu, fv);
        }
        bool acc = bhit && (!ok || bt_to < t_to);
        ok = ok || acc;
        t_to = acc ? bt_to : t_to;
        win = acc ? make_id(RUN_BVH, bface) : win;
        aux = acc ? run.first + i : aux;
        bu = acc ? fu : bu;
        bv = acc ? fv : bv;
      }
    }
  }
  Hit h;
  h.ok = ok;
  h.t = (float)t_to;
  h.win = win;
  h.aux = aux;
  h.u = bu;
  h.v = bv;
  return h;
}

// ================================================================== trace kernel
// Dynamic LDS: [ material records: lds_mats * 32 B ][ id stack: max_depth * blockDim entries ]
// The id stack is laid out [depth][thread] so the lanes of a wave touch consecutive
// bytes; entries are 4 bits when there are at most 16 materials (two levels per byte, lc.wide_ids == 2:
// half the LDS, which is what lets a sixth wave per SIMD of the list kernel in at depth 50), uint8 when
// every material id fits a byte, else uint16 (lc.wide_ids == 1).
struct LaunchCfg {
  int32_t lds_mats;    // materials staged in LDS (0: read them from global memory)
  int32_t wide_ids;    // 0: uint8 stack entries, 1: uint16, 2: 4-bit (two levels per byte)
  int32_t stack_off;   // byte offset of the id stack inside dynamic LDS
  int32_t nodes_off;   // byte offset of the staged reference-tree nodes
  int32_t lds_nodes;   // reference-tree nodes staged in LDS (the first lds_nodes of SceneDev::nodes)
  int32_t mesh_off;    // byte offset of the per-wave mesh-search regions (kMeshWaveWords words each; BVH variants)
  int32_t exclusive;   // 1: while a wave holds an outlier pixel, its other lanes take no new pixels (they work for it)
  int32_t pairs_off;   // byte offset of the staged PairPts records, -1: not staged (plain list scan)
  int32_t list_off;    // byte offset of the per-wave regions of the shared candidate tests, -1: each lane tests its own
  int32_t paths_off;   // byte offset of the staged leaf-path words
  int32_t lds_paths;   // leaf-path words staged in LDS (the first lds_paths of SceneDev::leaf_paths)
  int32_t pad2[1];
  const uint32_t *tile_order;  // optional: the queue hands out local tile tile_order[k] as its k-th tile
  const uint32_t *sparse_items;  // optional (with tile_order): leading work items handed to every sparse_stride-th lane only
  int32_t sparse_stride;         // power of two (RenderTuning::sparse_stride)
};

template <uint32_t F>
__device__ __forceinline__ void render_body(const SceneDev &sc, const FrameDev &fr, const LaunchCfg &lc,
                                            uint32_t *__restrict__ states, float *__restrict__ out,
                                            uint32_t *__restrict__ ray_counts,
                                            unsigned long long *__restrict__ counters) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  MatRec *s_mats = reinterpret_cast<MatRec *>(smem);
  // id stack: byte offset of entry [level][thread] in LDS, kept as 32-bit arithmetic (pointer
  // arithmetic on the generic pointers costs a register pair per live address)
  const uint32_t ids_shift = lc.wide_ids == 1 ? 1u : 0u;
  const bool nibble_ids = lc.wide_ids == 2;
  auto ids_offset = [&](int level) -> uint32_t {  // (nibble_ids: the byte of levels 2k and 2k + 1 is row k)
    return (uint32_t)lc.stack_off + (((uint32_t)level * (uint32_t)blockDim.x + threadIdx.x) << ids_shift);
  };
  const BvhNode *s_nodes = reinterpret_cast<const BvhNode *>(smem + lc.nodes_off);
  int *wl = nullptr;  // this wave's mesh-search region
  if (F & F_BVH)
    wl = reinterpret_cast<int *>(smem + lc.mesh_off) +
         __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * kMeshWaveWords;
  const float4 *s_pairs = nullptr;  // corners of the world-list pairs (culled scan) or nullptr (plain scan)
  if ((F & F_TRIS) && lc.pairs_off >= 0) {
    s_pairs = reinterpret_cast<const float4 *>(smem + lc.pairs_off);
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sc.pair_pts);
    uint32_t *dst = reinterpret_cast<uint32_t *>(smem + lc.pairs_off);
    for (int w = threadIdx.x; w < sc.n_pairs * 16; w += blockDim.x) dst[w] = src[w];
  }
  int *ll = nullptr;  // this wave's region for the shared candidate tests of the culled list scan
  if ((F & F_TRIS) && lc.list_off >= 0)
    ll = reinterpret_cast<int *>(smem + lc.list_off) +
         __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) * (64 * 8 + kListTasks * (1 + ((F & F_TEX) ? 6 : 2)));
  const bool mats_in_lds = lc.lds_mats > 0;
  const bool fast_fold = mats_in_lds && lc.wide_ids != 1 && sc.unsigned_colours;  // see the radiance fold
  if (mats_in_lds) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sc.mats);
    uint32_t *dst = reinterpret_cast<uint32_t *>(s_mats);
    for (int w = threadIdx.x; w < lc.lds_mats * 8; w += blockDim.x) dst[w] = src[w];
  }
  if ((F & F_BVH) && lc.lds_nodes > 0) {
    const uint32_t *src = reinterpret_cast<const uint32_t *>(sc.nodes);
    uint32_t *dst = reinterpret_cast<uint32_t *>(smem + lc.nodes_off);
    for (int w = threadIdx.x; w < lc.lds_nodes * 8; w += blockDim.x) dst[w] = src[w];
  }
  const int *s_paths = reinterpret_cast<const int *>(smem + lc.paths_off);
  if ((F & F_BVH) && lc.lds_paths > 0) {
    int *dst = reinterpret_cast<int *>(smem + lc.paths_off);
    for (int w = threadIdx.x; w < lc.lds_paths; w += blockDim.x) dst[w] = sc.leaf_paths[w];
  }
  __syncthreads();

  const int64_t n_items = fr.items;
  const bool w_pow2 = (fr.width & (fr.width - 1)) == 0, h_pow2 = (fr.height & (fr.height - 1)) == 0;
  const double inv_w = 1.0 / (double)fr.width, inv_h = 1.0 / (double)fr.height;
  // per-lane pixel state
  int64_t q = 0;
  int pi = 0, pj = 0, k = 0;
  bool has_px = false, done = false, active = false;
  bool heavy = false;  // a pixel of the queue's sparse head (see below)
  V3 color = splat(0.f);
  uint32_t rays = 0;
  unsigned long long ray_total = 0;
  Rng rng = {0, 0, 0, 0, 0, 0};
  // per-lane path state
  V3 o = splat(0.f), d = splat(0.f);
  int depth = 0;
  // Layer stack of ray_tracing.cuh:9-15.  Layer::emitted is 0 for every material that
  // scatters (only DiffuseLight and Sky emit, and neither scatters), so a layer is its
  // attenuation.  Without image textures the attenuation is the material's constant
  // colour and the layer is stored as a material id in LDS; with image textures the
  // sampled colour itself is kept (private memory).
  constexpr int kAttFloats = (F & F_TEX) ? RTMI_KERNEL_MAX_DEPTH * 3 : 3;
  float att[kAttFloats];

  // Mesh variants, frames dominated by a few outlier tiles (their pixels bounce to the depth limit
  // inside the mesh, tens of times the median cost): the frame time is the serial chain of the
  // slowest pixel, and what shortens a chain is the wave-cooperative search, which needs few rays
  // per wave.  The first sparse_limit work items (the outlier tiles, longest-first order) are
  // therefore spread thin -- one pixel per lc.sparse_stride lanes -- while the rest of the frame runs
  // with full waves.
  unsigned long long sparse_limit = 0ull;
  if ((F & F_BVH) && lc.sparse_items) sparse_limit = *lc.sparse_items;
  auto take_item = [&](int64_t item) -> bool {  // false: ragged-tile padding (or nothing to sample), written as black
    q = item;
    int64_t idx = frame_pixel_of_rank(fr, fr.rank, q);
    if (idx < 0 || fr.spp <= 0) {
      out[q * 3 + 0] = 0.f, out[q * 3 + 1] = 0.f, out[q * 3 + 2] = 0.f;
      if (ray_counts) ray_counts[q] = 0;
      return false;
    }
    pi = (int)(idx / fr.width);
    pj = (int)(idx % fr.width);
    rng.d = states[0 * n_items + q];
    rng.v0 = states[1 * n_items + q];
    rng.v1 = states[2 * n_items + q];
    rng.v2 = states[3 * n_items + q];
    rng.v3 = states[4 * n_items + q];
    rng.v4 = states[5 * n_items + q];
    k = 0;
    rays = 0;
    color = splat(0.f);
    has_px = true;
    return true;
  };

  RTMI_STAT(MeshStats st = {}; unsigned wave_queries = 0; const unsigned long long t_begin = stat_real();)
  for (;;) {
    RTMI_STAT(const unsigned long long tq0 = stat_now();)
    // -------------------------------------------------------- sample / pixel bookkeeping
    if (!active && !done && has_px && k >= fr.spp) {
      V3 c = color;
      if (fr.post) {  // ray_tracing.cu:78-83
        c = c / (float)fr.spp;
        c = mk(clamp1(c.x, 0.f, 1.f), clamp1(c.y, 0.f, 1.f), clamp1(c.z, 0.f, 1.f));
        c = mk(sqrtf(c.x), sqrtf(c.y), sqrtf(c.z));
      }
      out[q * 3 + 0] = c.x;
      out[q * 3 + 1] = c.y;
      out[q * 3 + 2] = c.z;
      if (ray_counts) ray_counts[q] = rays;
      ray_total += rays;
      states[0 * n_items + q] = rng.d;
      states[1 * n_items + q] = rng.v0;
      states[2 * n_items + q] = rng.v1;
      states[3 * n_items + q] = rng.v2;
      states[4 * n_items + q] = rng.v3;
      states[5 * n_items + q] = rng.v4;
      has_px = false;
    }
    const bool wave_heavy = (F & F_BVH) && lc.exclusive &&
                            __builtin_amdgcn_ballot_w64(has_px && heavy && (active || k < fr.spp)) != 0ull;
    if (!active && !done) {
      while (!has_px && !done) {
        if ((F & F_BVH) && sparse_limit != 0ull && (threadIdx.x & (uint32_t)(lc.sparse_stride - 1)) != 0) {
          if (wave_heavy) break;  // this wave is busy with an outlier pixel: stay a helper
          // the head of the queue holds the outlier tiles: only every sparse_stride-th lane takes
          // pixels there (the others look again next round), so that a wave carries few rays
          // and the mesh search runs in its cooperative mode
          if (__hip_atomic_load(&counters[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < sparse_limit) break;
        }
        const unsigned long long nq = atomicAdd(&counters[0], 1ull);
        if ((int64_t)nq >= n_items) {
          done = true;
          break;
        }
        int64_t item = (int64_t)nq;
        if (lc.tile_order) item = (int64_t)lc.tile_order[nq >> 6] * 64 + (int64_t)(nq & 63);
        if (!take_item(item)) continue;
        heavy = nq < sparse_limit;
      }
      if (has_px) {
        // ray_tracing.cu:68-74 + camera.cu:57-70
        float r1 = rng_01(rng);
        float r2 = rng_01(rng);
        // division by a power of two is an exact scaling: multiply by the (exact) reciprocal
        double x = (double)r1 + (double)pj;
        double y = (double)r2 + (double)(fr.height - pi);
        x = w_pow2 ? x * inv_w : x / (double)fr.width;
        y = h_pow2 ? y * inv_h : y / (double)fr.height;
        x = 2 * x - 1;
        y = 2 * y - 1;
        x = (x + 1) / 2;
        y = (y + 1) / 2;
        V3 target = sc.cam.llc + (float)x * sc.cam.horizontal + (float)y * sc.cam.vertical;
        V3 origin = sc.cam.position;
        if (F & F_DEFOCUS) {
          if (sc.cam.defocus) {  // camera.cu:63-65,74-77 (a square, drawn left to right)
            float ox = rng_range(0.f, sc.cam.lens_radius, rng);
            float oy = rng_range(0.f, sc.cam.lens_radius, rng);
            origin = sc.cam.position + sc.cam.u * ox + sc.cam.v * oy;
          }
        }
        o = origin;
        d = unit3_rn(unit3_rn(target - origin));  // RayAt normalises, Ray's constructor normalises again
        k++;
        depth = 0;
        active = true;
      }
    }
    if (!__any(active)) {
      if (!(F & F_BVH) || __all(done)) break;
      continue;  // lanes held back from the sparse head of the queue: it has just moved on
    }

    RTMI_STAT(wave_queries++; const unsigned long long tq1 = stat_now(); st.cyc[0] += tq1 - tq0;
              const unsigned long long in0 = st.cyc[2] + st.cyc[3];)
    Hit h = {};
    const bool all_lanes_in = (F & F_BVH) || ((F & F_TRIS) && s_pairs != nullptr);  // wave-uniform
    if (all_lanes_in)  // every lane goes in, with or without a ray of its own: see closest_hit
      h = closest_hit<F>(sc, s_nodes, lc.lds_nodes, s_paths, lc.lds_paths, s_pairs, ll, wl, counters + 2, o, d, active
#ifdef RTMI_STATS
                         , st
#endif
      );
    RTMI_STAT(const unsigned long long tq2 = stat_now(); st.cyc[1] += (tq2 - tq1) - (st.cyc[2] + st.cyc[3] - in0);)
    if (active) {
      if (!all_lanes_in)
        h = closest_hit<F>(sc, s_nodes, 0, s_paths, 0, s_pairs, nullptr, nullptr, nullptr, o, d, true
#ifdef RTMI_STATS
                           , st
#endif
        );
      rays++;

      V3 result = splat(0.f);
      bool ended = true;
      if (h.ok && depth < fr.max_depth) {  // ray_tracing.cu:23
        const uint32_t kind = h.win >> 29;
        const uint32_t index = h.win & ID_INDEX_MASK;
        V3 p = o + h.t * d;  // ray_tracing.cu:32 and the materials' own `p`
        if (kind == RUN_SKY) {
          // sky.cu:9-14: Scatter false; Emit(p) = gradient on normalize(p)
          V3 dir = unit3_rn(p);
          float tg = (float)(0.5 * ((double)dir.y + 1.0));
          float w0 = 1.0f - tg;
          result = mk(w0 * 1.0f + tg * 0.5f, w0 * 1.0f + tg * 0.7f, w0 * 1.0f + tg * 1.0f);
        } else {
          V3 nrm = splat(0.f);
          int mat = 0;
          float tu = 0.f, tv = 0.f;  // record.u, record.v (only read by image textures)
          if ((F & F_TRIS) && kind == RUN_TRIS) {
            const HotTri &tr = sc.tris[index];  // per-lane gather of the winner (L1/L2 resident)
            V3 n = mk(tr.n[0], tr.n[1], tr.n[2]);
            nrm = dot3(d, n) < 0.f ? n : -n;  // utils.cu:80
            mat = tr.mat;
            if (F & F_TEX) {
              const int flags = tr.flags;
              if (flags & TRI_PGRAM) {  // parallelogram.cu:26-29,35-38
                float w = (float)((1.0 - (double)h.u) - (double)h.v);
                if (!(flags & TRI_SECOND)) {
                  tu = (0.f * w + 1.f * h.u) + 0.f * h.v;
                  tv = (1.f * w + 1.f * h.u) + 0.f * h.v;
                } else {
                  tu = (1.f * w + 0.f * h.u) + 1.f * h.v;
                  tv = (1.f * w + 0.f * h.u) + 0.f * h.v;
                }
              } else {
                tu = h.u, tv = h.v;  // triangle.cu:13
              }
            }
          }
          if ((F & F_SPHERE) && kind == RUN_SPHERE) {
            const SphereRec &sr = sc.spheres[index];
            nrm = unit3_rn(p - mk(sr.cx, sr.cy, sr.cz));  // sphere.cu:25-26
            mat = sr.mat;
            if (F & F_TEX) {  // sphere.cu:60-63
              const float pi_f = 3.14159265358979323846264338327950288f;
              float theta = acosf(-nrm.y);
              float phi = atan2f(-nrm.z, nrm.x) + pi_f;
              tu = phi / (2 * pi_f);
              tv = theta / pi_f;
            }
          }
          if ((F & F_BVH) && kind == RUN_BVH) {
            const FaceRec &fc = sc.faces[index];
            // utils.cu:79: normalize(cross(v0v1, v0v2)), recomputed for the winning face only
            V3 n = unit3_rn(cross3(mk(fc.e1[0], fc.e1[1], fc.e1[2]), mk(fc.e2[0], fc.e2[1], fc.e2[2])));
            nrm = dot3(d, n) < 0.f ? n : -n;
            const BvhRec br = sc.bvhs[h.aux];
            mat = br.mat;
            if ((F & F_TEX) && br.has_uv) {  // bvh.cuh:41-45
              const float *tc = sc.face_uv + (size_t)(br.face_base + fc.orig) * 6;
              float w = (float)((1.0 - (double)h.u) - (double)h.v);
              tu = (tc[0] * w + tc[2] * h.u) + tc[4] * h.v;
              tv = (tc[1] * w + tc[3] * h.u) + tc[5] * h.v;
            }
          }
          const MatRec m = mats_in_lds ? s_mats[mat] : sc.mats[mat];
          V3 rgb = mk(m.r, m.g, m.b);
          RTMI_STAT2(if (!(F & F_BVH)) { const unsigned long long tsa = stat_now();  // (divergent code: first active lane reports)
            if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) g_wave_stats[((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & 16383u][9] += tsa - tq2; })
          if (F & F_TEX) {
            if (m.tex >= 0 && (m.kind == MAT_LAMBERTIAN || m.kind == MAT_LIGHT)) {
              // image_texture.cu:11-13: v = 1.0 - v in double, then float coordinates
              rgb = tex_sample(sc.texs[m.tex], tu, (float)(1.0 - (double)tv));
            }
          }
          if (m.kind == MAT_LIGHT) {
            result = rgb;  // diffuse_light.cu:5-13
          } else {
            const float dn = dot3(d, nrm);
            V3 nd = splat(0.f);
            bool scattered = false;
            if (m.kind == MAT_LAMBERTIAN) {  // lambertian.cu:33-43
              if (!(dn >= 0.f)) {
                float sum;
                V3 s = ball_sample(rng, sum);
                const float l = sqrtf(sum);
                s = mk(s.x / l, s.y / l, s.z / l);
                nd = unit3_rn(s + nrm);
                scattered = true;
              }
            } else if (m.kind == MAT_METAL) {  // metal.cu:12-25
              if (!(dn >= 0.f)) {
                V3 refl = reflect3(d, nrm);
                if (m.param > 0.f) {
                  float sum;
                  V3 s = ball_sample(rng, sum);
                  nd = refl + m.param * s;
                } else {
                  nd = refl;
                }
                scattered = true;
              }
            } else {  // MAT_DIELECTRIC, dielectric.cu:16-44
              if (dn >= 0.f)
                nd = refract3(d, -nrm, m.param / 1.0f);
              else
                nd = refract3(d, nrm, 1.0f / m.param);
              bool zero = (nd.x == 0.f && nd.y == 0.f && nd.z == 0.f);
              bool nan = (nd.x != nd.x) || (nd.y != nd.y) || (nd.z != nd.z);
              scattered = !(zero || nan);
            }
            if (scattered) {
              if (F & F_TEX) {
                att[depth * 3 + 0] = rgb.x;
                att[depth * 3 + 1] = rgb.y;
                att[depth * 3 + 2] = rgb.z;
              } else if (nibble_ids) {
                const uint32_t at = ids_offset(depth >> 1);  // this lane's own byte: no other lane writes it
                const uint32_t old = smem[at];
                smem[at] = (uint8_t)((depth & 1) ? ((old & 0x0fu) | ((uint32_t)mat << 4)) : ((old & 0xf0u) | (uint32_t)mat));
              } else if (lc.wide_ids) {
                *reinterpret_cast<uint16_t *>(smem + ids_offset(depth)) = (uint16_t)mat;
              } else {
                smem[ids_offset(depth)] = (uint8_t)mat;
              }
              depth++;
              o = p;
              d = unit3_rn(nd);  // Ray's constructor
              ended = false;
            }
          }
        }
      }
      RTMI_STAT2(if (!(F & F_BVH)) { const unsigned long long tsb = stat_now();
        if ((int)(threadIdx.x & 63u) == __builtin_ctzll(__ballot(1))) g_wave_stats[((blockIdx.x * blockDim.x + threadIdx.x) >> 6) & 16383u][10] += tsb - tq2; })
      if (ended) {
        // ray_tracing.cu:50-52 with emitted == 0 on every stored layer: result = emitted +
        // attenuation * result, deepest layer first.  The addition only matters for a product of -0,
        // which needs a colour with its sign bit set (sc.unsigned_colours).
        int i = depth - 1;
        if (!(F & F_TEX) && fast_fold) {
          // common case (byte ids, material table in LDS, no signed colours) without the per-layer
          // uniform branches: four layers at a time, ids first, then colours, then the products
          const uint32_t step = blockDim.x;
          if (nibble_ids) {
            if (i >= 0 && !(i & 1)) {  // an even top level sits alone in the low half of its byte
              const int m0 = smem[ids_offset(i >> 1)] & 15;
              result = mk(s_mats[m0].r * result.x, s_mats[m0].g * result.y, s_mats[m0].b * result.z);
              i--;
            }
            for (; i >= 3; i -= 4) {  // i odd: bytes (i >> 1) and (i >> 1) - 1 hold levels i, i - 1 and i - 2, i - 3
              const uint32_t at = ids_offset(i >> 1);
              const uint32_t b0 = smem[at], b1 = smem[at - step];
              const int m0 = b0 >> 4, m1 = b0 & 15, m2 = b1 >> 4, m3 = b1 & 15;
              const V3 a0 = mk(s_mats[m0].r, s_mats[m0].g, s_mats[m0].b), a1 = mk(s_mats[m1].r, s_mats[m1].g, s_mats[m1].b);
              const V3 a2 = mk(s_mats[m2].r, s_mats[m2].g, s_mats[m2].b), a3 = mk(s_mats[m3].r, s_mats[m3].g, s_mats[m3].b);
              result = mk(a0.x * result.x, a0.y * result.y, a0.z * result.z);
              result = mk(a1.x * result.x, a1.y * result.y, a1.z * result.z);
              result = mk(a2.x * result.x, a2.y * result.y, a2.z * result.z);
              result = mk(a3.x * result.x, a3.y * result.y, a3.z * result.z);
            }
            for (; i >= 1; i -= 2) {
              const uint32_t b0 = smem[ids_offset(i >> 1)];
              const int m0 = b0 >> 4, m1 = b0 & 15;
              result = mk(s_mats[m0].r * result.x, s_mats[m0].g * result.y, s_mats[m0].b * result.z);
              result = mk(s_mats[m1].r * result.x, s_mats[m1].g * result.y, s_mats[m1].b * result.z);
            }
          }
          for (; i >= 3; i -= 4) {
            const uint32_t at = ids_offset(i);
            const int m0 = smem[at], m1 = smem[at - step], m2 = smem[at - 2u * step], m3 = smem[at - 3u * step];
            const V3 a0 = mk(s_mats[m0].r, s_mats[m0].g, s_mats[m0].b), a1 = mk(s_mats[m1].r, s_mats[m1].g, s_mats[m1].b);
            const V3 a2 = mk(s_mats[m2].r, s_mats[m2].g, s_mats[m2].b), a3 = mk(s_mats[m3].r, s_mats[m3].g, s_mats[m3].b);
            result = mk(a0.x * result.x, a0.y * result.y, a0.z * result.z);
            result = mk(a1.x * result.x, a1.y * result.y, a1.z * result.z);
            result = mk(a2.x * result.x, a2.y * result.y, a2.z * result.z);
            result = mk(a3.x * result.x, a3.y * result.y, a3.z * result.z);
          }
          for (; i >= 0; i--) {
            const int m0 = smem[ids_offset(i)];
            result = mk(s_mats[m0].r * result.x, s_mats[m0].g * result.y, s_mats[m0].b * result.z);
          }
        }
        for (; i >= 0; i--) {
          V3 a;
          if (F & F_TEX) {
            a = mk(att[i * 3 + 0], att[i * 3 + 1], att[i * 3 + 2]);
          } else {
            const int mi = nibble_ids     ? (int)((smem[ids_offset(i >> 1)] >> ((i & 1) * 4)) & 15u)
                           : lc.wide_ids ? (int)*reinterpret_cast<const uint16_t *>(smem + ids_offset(i))
                                         : (int)smem[ids_offset(i)];
            if (mats_in_lds) {
              a = mk(s_mats[mi].r, s_mats[mi].g, s_mats[mi].b);
            } else {
              a = mk(sc.mats[mi].r, sc.mats[mi].g, sc.mats[mi].b);
            }
          }
          if (sc.unsigned_colours) {
            result = mk(a.x * result.x, a.y * result.y, a.z * result.z);
          } else {
            result = mk(0.f + a.x * result.x, 0.f + a.y * result.y, 0.f + a.z * result.z);
          }
        }
        color = color + result;
        active = false;
      }
    }
    RTMI_STAT(st.cyc[4] += stat_now() - tq2;)
  }

  // total closest-hit queries: wave reduce, one atomic per wave
  for (int off = 32; off > 0; off >>= 1) ray_total += __shfl_down(ray_total, off);
  if ((threadIdx.x & 63) == 0 && ray_total) atomicAdd(&counters[1], ray_total);
#ifdef RTMI_STATS
  if ((threadIdx.x & 63) == 0) {
    const unsigned v[13] = {wave_queries, st.searches, st.node_steps, st.face_steps, st.nodes_popped, st.blocks_popped,
                            st.insert_rounds, st.steps_hist[0], st.steps_hist[1], st.steps_hist[2], st.steps_hist[3],
                            st.steps_hist[4], st.steps_hist[5]};
    for (int i = 0; i < 13; i++) atomicAdd(&counters[4 + i], (unsigned long long)v[i]);
    for (int i = 0; i < 9; i++) atomicAdd(&counters[17 + i], st.cyc[i]);
    const unsigned long long life = stat_real() - t_begin;
    atomicAdd(&counters[26], life);
    atomicMax(&counters[27], life);
    atomicAdd(&counters[28], 1ull);
    atomicAdd(&counters[29], st.calib);
    atomicAdd(&counters[30], st.cull_bits);
    atomicAdd(&counters[31], st.cull_iters);
    atomicAdd(&counters[3], st.cull_rays);
    const unsigned wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (!(F & F_BVH) && wid < 16384u) {
      atomicAdd(&counters[25], g_wave_stats[wid][9]);
      atomicAdd(&counters[6], g_wave_stats[wid][10]);
      g_wave_stats[wid][9] = 0, g_wave_stats[wid][10] = 0;
    } else if (wid < 16384u) {
      g_wave_stats[wid][0] = life;
      for (int i = 0; i < 9; i++) g_wave_stats[wid][1 + i] = st.cyc[i];
      g_wave_stats[wid][10] = wave_queries, g_wave_stats[wid][11] = st.node_steps, g_wave_stats[wid][12] = st.face_steps;
      g_wave_stats[wid][13] = st.nodes_popped, g_wave_stats[wid][14] = st.insert_rounds;
      g_wave_stats[wid][15] = (st.cyc[9] << 32) | (st.cyc[10] >> 8);  // setup cycles | loop-control cycles / 256
    }
  }
#endif
}

// The trace kernel proper, and the same code under a second name for the scheduler's 2-spp
// cost probe (so per-kernel profiles keep the two apart).
// Second launch bound = waves per SIMD the register allocation must allow: the list-only variants
// sit at the 80-VGPR / 6-wave step and are issue-bound (one wave less costs 5 %), so the step is
// held explicitly instead of being left to the allocator's luck.
#ifndef RTMI_TRIS_WAVES
#define RTMI_TRIS_WAVES 6
#endif
#define RTMI_MIN_WAVES(F) (((F) & (F_BVH | F_TEX | F_SPHERE)) ? 1 : ((F) & F_TRIS) ? RTMI_TRIS_WAVES : 6)
// Mesh variants share their per-workgroup tables (reference-tree nodes, materials) between more waves:
// workgroups of up to 512 lanes, two of which fill a CU's LDS with 16 waves' search regions.
#define RTMI_MAX_THREADS(F) (((F) & F_BVH) ? 512 : 256)
template <uint32_t F>
__global__ __launch_bounds__(RTMI_MAX_THREADS(F), RTMI_MIN_WAVES(F)) void render_kernel(SceneDev sc, FrameDev fr, LaunchCfg lc,
                                                      uint32_t *__restrict__ states, float *__restrict__ out,
                                                      uint32_t *__restrict__ ray_counts,
                                                      unsigned long long *__restrict__ counters) {
  render_body<F>(sc, fr, lc, states, out, ray_counts, counters);
}
template <uint32_t F>
__global__ __launch_bounds__(RTMI_MAX_THREADS(F), RTMI_MIN_WAVES(F)) void probe_kernel(SceneDev sc, FrameDev fr, LaunchCfg lc,
                                                     uint32_t *__restrict__ states, float *__restrict__ out,
                                                     uint32_t *__restrict__ ray_counts,
                                                     unsigned long long *__restrict__ counters) {
  render_body<F>(sc, fr, lc, states, out, ray_counts, counters);
}

// ------------------------------------------------------------------ untile / post
template <typename E, int C>
__global__ __launch_bounds__(256) void untile_kernel(FrameDev fr, const E *__restrict__ tiles, E *__restrict__ image) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  int64_t total = fr.items * fr.world;
  if (g >= total) return;
  int rank = (int)(g / fr.items);
  int64_t q = g % fr.items;
  int64_t idx = frame_pixel_of_rank(fr, rank, q);
  if (idx < 0) return;
#pragma unroll
  for (int c = 0; c < C; c++) image[idx * C + c] = tiles[g * C + c];
}

__global__ __launch_bounds__(256) void post_kernel(float *__restrict__ img, int64_t n, int spp) {
  int64_t g = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (g >= n) return;
  img[g] = sqrtf(clamp1(img[g] / (float)spp, 0.f, 1.f));  // utils.cu:127-128
}

// ------------------------------------------------------------------ longest-first tile order
// A pixel is an indivisible serial chain (its samples share one RNG stream), so the end of a
// frame is a tail of lanes finishing their last pixel.  Handing out the expensive tiles first
// shortens that tail (longest-processing-time-first).  Cost estimate: rays per tile measured by
// a 2-spp probe pass on a scratch copy of the RNG states.  The order only changes which lane
// renders which pixel when; every pixel's arithmetic is unchanged.
__global__ __launch_bounds__(256) void tile_cost_kernel(const uint32_t *__restrict__ ray_counts, int n_tiles,
                                                         uint32_t *__restrict__ cost, uint32_t *__restrict__ max_cost) {
  int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= n_tiles) return;
  const uint4 *p = reinterpret_cast<const uint4 *>(ray_counts + (size_t)t * 64);
  uint32_t s = 0;
#pragma unroll
  for (int i = 0; i < 16; i++) {
    uint4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  cost[t] = s;
  atomicMax(max_cost, s);
}

// One workgroup: counting sort of the tiles into 256 cost buckets, most expensive bucket first.
// Also sizes the "sparse" head of the queue (sparse_items, in work items; render_body, mesh
// variants): the outlier tiles -- at least twice the mean cost, in a frame whose most expensive
// tile costs at least three times the mean -- up to `sparse_cap` items, what the grid can hold at
// one pixel per kSparseStride lanes.
__global__ __launch_bounds__(1024) void tile_order_kernel(const uint32_t *__restrict__ cost,
                                                          const uint32_t *__restrict__ max_cost, int n_tiles,
                                                          uint32_t *__restrict__ order,
                                                          uint32_t *__restrict__ sparse_items, uint32_t sparse_cap,
                                                          uint32_t outlier_x10) {
  __shared__ uint32_t bins[256];
  __shared__ uint32_t base[256];
  __shared__ unsigned long long total;
  __shared__ uint32_t outliers;
  for (int i = threadIdx.x; i < 256; i += blockDim.x) bins[i] = 0;
  if (threadIdx.x == 0) total = 0ull, outliers = 0u;
  __syncthreads();
  const uint32_t mx = *max_cost > 0 ? *max_cost : 1;
  unsigned long long part = 0ull;
  for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
    uint32_t b = 255u - (uint32_t)(((unsigned long long)cost[t] * 255ull) / mx);  // bucket 0 = most expensive
    atomicAdd(&bins[b], 1u);
    part += cost[t];
  }
  atomicAdd(&total, part);
  __syncthreads();
  {
    const unsigned long long sum = total;  // mean = sum / n_tiles; compare cost * n_tiles with k * sum
    uint32_t mine = 0;
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x)
      if ((unsigned long long)cost[t] * (unsigned long long)n_tiles * 10ull >= (unsigned long long)outlier_x10 * sum) mine++;
    if (mine) atomicAdd(&outliers, mine);
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; i++) {
      base[i] = run;
      run += bins[i];
    }
    const bool skewed = (unsigned long long)mx * (unsigned long long)n_tiles >= 3ull * total;
    const uint32_t items = outliers * 64u;
    *sparse_items = skewed ? (items < sparse_cap ? items : sparse_cap) : 0u;
  }
  __syncthreads();
  for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
    uint32_t b = 255u - (uint32_t)(((unsigned long long)cost[t] * 255ull) / mx);
    order[atomicAdd(&base[b], 1u)] = (uint32_t)t;
  }
}

hipError_t launch_tile_order(const uint32_t *d_ray_counts, int n_tiles, uint32_t *d_cost, uint32_t *d_meta,
                             uint32_t *d_order, uint32_t sparse_cap, int outlier_x10, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_meta, 0, 16 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(tile_cost_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, d_ray_counts, n_tiles, d_cost,
                     d_meta);
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, d_cost, d_meta, n_tiles, d_order, d_meta + 1,
                     sparse_cap, (uint32_t)outlier_x10);
  return hipGetLastError();
}

// ------------------------------------------------------------------ self-test
// Runs every shortened operation against the expression it replaces on all 2^32 inputs:
// bad[0] rcp_rn(x) vs 1.0f / x inside rcp_rn's domain (must be 0), bad[1] outside it
// (informative), bad[2] rng_pm1_of(x) vs the reference's uniform(-1, 1) expression, bad[3]
// rng_01_of(x) vs uniform(0, 1) (both must be 0).
__device__ __forceinline__ float ref_uniform(uint32_t x) { return (float)x * 2.3283064e-10f + 1.16415322e-10f; }
__device__ __forceinline__ float ref_range(float mn, float mx, uint32_t x) { return ref_uniform(x) * (mx - mn) + mn; }
__global__ __launch_bounds__(256) void arithmetic_selftest(unsigned long long *bad, float mn1, float mx1, float mn0,
                                                           float mx0) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long in_domain = 0, outside = 0, pm1 = 0, u01 = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const uint32_t bits = (uint32_t)i;
    const float x = __uint_as_float(bits);
    const float ref = 1.0f / x, got = rcp_rn(x);
    const bool same = __float_as_uint(ref) == __float_as_uint(got) || (ref != ref && got != got);
    const bool dom = fabsf(x) >= 0x1p-126f && fabsf(x) < RCP_RN_LIMIT;
    if (!same) {
      if (dom) in_domain++; else outside++;
    }
    // the range bounds arrive as kernel arguments so that the reference expression is
    // evaluated operation by operation, not folded at compile time
    if (__float_as_uint(ref_range(mn1, mx1, bits)) != __float_as_uint(rng_pm1_of(bits))) pm1++;
    if (__float_as_uint(ref_range(mn0, mx0, bits)) != __float_as_uint(rng_01_of(bits))) u01++;
  }
  if (in_domain) atomicAdd(&bad[0], in_domain);
  if (outside) atomicAdd(&bad[1], outside);
  if (pm1) atomicAdd(&bad[2], pm1);
  if (u01) atomicAdd(&bad[3], u01);
}
#ifdef RTMI_STATS
hipError_t copy_wave_stats(unsigned long long *host, size_t bytes) {
  return hipMemcpyFromSymbol(host, HIP_SYMBOL(g_wave_stats), bytes);
}
#endif
hipError_t launch_arithmetic_selftest(unsigned long long *d_bad, hipStream_t stream) {
  hipLaunchKernelGGL(arithmetic_selftest, dim3(4096), dim3(256), 0, stream, d_bad, -1.f, 1.f, 0.f, 1.f);
  return hipGetLastError();
}

// ------------------------------------------------------------------ launchers
static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }

hipError_t launch_rng_init(uint64_t seed, const FrameDev &fr, const uint32_t *d_jump, uint32_t *d_states,
                           hipStream_t stream) {
  if (fr.items == 0) return hipSuccess;
  hipLaunchKernelGGL(rng_init_kernel, dim3((unsigned)cdiv(fr.items, 256)), dim3(256), 0, stream, seed, fr, d_jump,
                     d_states);
  return hipGetLastError();
}

// RTMI_PLAIN_LIST=1 (read once): keep the uncultured list scan, for A/B measurements
static bool plain_list_scan() {
  static const bool v = [] {
    const char *e = getenv("RTMI_PLAIN_LIST");
    return e && atoi(e) != 0;
  }();
  return v;
}

static LaunchCfg make_cfg(uint32_t variant, const SceneDev &sc, const FrameDev &fr, int threads, size_t *lds_bytes) {
  LaunchCfg lc{};
  lc.tile_order = nullptr;
  lc.lds_mats = sc.n_mats <= kLdsMats ? sc.n_mats : 0;
  lc.wide_ids = sc.n_mats > 256 ? 1 : sc.n_mats <= 16 ? 2 : 0;
  size_t off = ((size_t)lc.lds_mats * sizeof(MatRec) + 15) & ~(size_t)15;
  lc.stack_off = (int32_t)off;
  const size_t levels = (size_t)(fr.max_depth > 0 ? fr.max_depth : 1);
  size_t stack = (variant & F_TEX) ? 0 : (lc.wide_ids == 2 ? (levels + 1) / 2 : levels) * threads * (lc.wide_ids == 1 ? 2 : 1);
  size_t noff = (off + stack + 15) & ~(size_t)15;
  lc.nodes_off = (int32_t)noff;
  lc.lds_nodes = (variant & F_BVH) ? (sc.n_nodes < kLdsNodes ? sc.n_nodes : kLdsNodes) : 0;
  size_t paoff = (noff + (size_t)lc.lds_nodes * sizeof(BvhNode) + 15) & ~(size_t)15;
  lc.paths_off = (int32_t)paoff;
  lc.lds_paths = (variant & F_BVH) ? (sc.n_leaf_paths < kLdsPaths ? sc.n_leaf_paths : kLdsPaths) : 0;
  size_t soff = (paoff + (size_t)lc.lds_paths * sizeof(int32_t) + 15) & ~(size_t)15;
  lc.mesh_off = (int32_t)soff;
  size_t poff = soff + ((variant & F_BVH) ? (size_t)(threads / 64) * kMeshWaveWords * sizeof(int) : 0);
  const bool cull = (variant & F_TRIS) && sc.n_pairs >= kCullMinPairs && sc.n_pairs <= kLdsPairs && !plain_list_scan();
  lc.pairs_off = cull ? (int32_t)poff : -1;
  size_t loff = (poff + (cull ? (size_t)sc.n_pairs * sizeof(PairPts) : 0) + 15) & ~(size_t)15;
  const bool share = cull;  // the culled scan always shares its candidate tests over the wave
  lc.list_off = share ? (int32_t)loff : -1;
  *lds_bytes = loff + (share ? (size_t)(threads / 64) * (64 * 8 + kListTasks * (1 + ((variant & F_TEX) ? 6 : 2))) * sizeof(int) : 0);
  return lc;
}

template <uint32_t F>
static hipError_t launch_render_t(const SceneDev &sc, const FrameDev &fr, uint32_t *d_states, float *d_out,
                                  uint32_t *d_ray_counts, unsigned long long *d_counters, const SchedPlan &plan,
                                  bool probe, int blocks, int threads, const RenderTuning &tune, hipStream_t stream) {
  size_t lds = 0;
  LaunchCfg lc = make_cfg(F, sc, fr, threads, &lds);
  lc.tile_order = plan.tile_order;
  lc.sparse_items = plan.sparse_items;
  lc.sparse_stride = tune.sparse_stride;
  lc.exclusive = tune.exclusive;
  if (lds > 64 * 1024) {  // above the default dynamic-LDS limit: ask for it (160 KiB per CU on gfx950)
    hipError_t e = hipFuncSetAttribute(probe ? reinterpret_cast<const void *>(probe_kernel<F>)
                                             : reinterpret_cast<const void *>(render_kernel<F>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  if (probe) {
    hipLaunchKernelGGL(probe_kernel<F>, dim3(blocks), dim3(threads), lds, stream, sc, fr, lc, d_states, d_out,
                       d_ray_counts, d_counters);
  } else {
    hipLaunchKernelGGL(render_kernel<F>, dim3(blocks), dim3(threads), lds, stream, sc, fr, lc, d_states, d_out,
                       d_ray_counts, d_counters);
  }
  return hipGetLastError();
}

template <uint32_t F>
static int occupancy_t(const SceneDev &sc, const FrameDev &fr, int threads) {
  int nb = 0;
  size_t lds = 0;
  (void)make_cfg(F, sc, fr, threads, &lds);
  if (threads > RTMI_MAX_THREADS(F) || lds > 160 * 1024) return 0;
  if (lds > 64 * 1024 &&
      hipFuncSetAttribute(reinterpret_cast<const void *>(render_kernel<F>), hipFuncAttributeMaxDynamicSharedMemorySize,
                          (int)lds) != hipSuccess)
    return 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, render_kernel<F>, threads, lds) != hipSuccess) nb = 0;
  return nb;
}

// The specialisations that are instantiated; a feature set outside them uses F_ALL.
#define RTMI_FOR_EACH_VARIANT(X)                     \
  X(0u)                                              \
  X(F_TRIS)                                          \
  X(F_SPHERE)                                        \
  X(F_TRIS | F_SPHERE)                               \
  X(F_TRIS | F_BVH)                                  \
  X(F_TRIS | F_SPHERE | F_TEX)                       \
  X(F_ALL)

uint32_t pick_variant(uint32_t features) {
#define X(V) \
  if ((features & ~(uint32_t)(V)) == 0) return (V);
  RTMI_FOR_EACH_VARIANT(X)
#undef X
  return F_ALL;
}

int render_occupancy(uint32_t variant, const SceneDev &sc, const FrameDev &fr, int threads) {
#define X(V) \
  if (variant == (uint32_t)(V)) return occupancy_t<(V)>(sc, fr, threads);
  RTMI_FOR_EACH_VARIANT(X)
#undef X
  return 0;
}

hipError_t launch_render(uint32_t variant, const SceneDev &sc, const FrameDev &fr, uint32_t *d_states, float *d_out,
                         uint32_t *d_ray_counts, unsigned long long *d_counters, const SchedPlan &plan, bool probe,
                         int blocks, int threads, const RenderTuning &tune, hipStream_t stream) {
#define X(V) \
  if (variant == (uint32_t)(V)) \
    return launch_render_t<(V)>(sc, fr, d_states, d_out, d_ray_counts, d_counters, plan, probe, blocks, threads, tune, \
                                stream);
  RTMI_FOR_EACH_VARIANT(X)
#undef X
  return hipErrorInvalidValue;
}

hipError_t launch_untile(const FrameDev &fr, const float *d_tiles, float *d_image, hipStream_t stream) {
  int64_t total = fr.items * fr.world;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL((untile_kernel<float, 3>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, fr, d_tiles,
                     d_image);
  return hipGetLastError();
}

hipError_t launch_untile_u32(const FrameDev &fr, const uint32_t *d_tiles, uint32_t *d_image, hipStream_t stream) {
  int64_t total = fr.items * fr.world;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL((untile_kernel<uint32_t, 1>), dim3((unsigned)cdiv(total, 256)), dim3(256), 0, stream, fr,
                     d_tiles, d_image);
  return hipGetLastError();
}

hipError_t launch_post(float *d_img, int64_t n, int spp, hipStream_t stream) {
  if (n == 0) return hipSuccess;
  hipLaunchKernelGGL(post_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, stream, d_img, n, spp);
  return hipGetLastError();
}

}  // namespace rtmi
