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

#include "trace_helpers.h"
#include "mesh_search.h"
#include "closest_hit.h"
#include "render_body.h"

// The trace kernel proper, and the same code under a second name for the scheduler's 2-spp
// cost probe (so per-kernel profiles keep the two apart).
// Second launch bound = waves per SIMD the register allocation must allow: the list-only variants
// sit at the 80-VGPR / 6-wave step and are issue-bound (one wave less costs 5 %), so the step is
// held explicitly instead of being left to the allocator's luck.
#ifndef RTMI_TRIS_WAVES
#define RTMI_TRIS_WAVES 6
#endif
#ifndef RTMI_BVH_WAVES
#define RTMI_BVH_WAVES 3
#endif
// Sphere and image-texture variants: four waves (at most 128 VGPRs: they sit just below that step, and the allocator's
// count swings by tens of registers with unrelated edits when it is not held) -- except the three that gain from a
// fifth (96 VGPRs, 6-21 spilled dwords) now that their task regions leave room for a fifth workgroup in a CU's LDS
// (scene_dev.h: kListTasks): triangles + spheres + textures (C5 shard 3.16 -> 2.93 s), the grouped sphere scan (spheres
// 1024^2: 22.2 -> 20.9 ms), short sphere runs (+2 %).  Spheres + triangles without textures lose 4-5 % at five (spills
// with no occupancy to show for them) and stay at four.  (tools/gpu_variant_time.py for the variants no workload uses)
#ifndef RTMI_SPH_WAVES
#define RTMI_SPH_WAVES(F) (((F) == (F_TRIS | F_SPHERE | F_TEX) || (F) == (F_SPHERE | F_SGROUP) || (F) == F_SPHERE) ? 5 : 4)
#endif
#define RTMI_MIN_WAVES(F) (((F) & F_BVH) ? RTMI_BVH_WAVES : ((F) & (F_TEX | F_SPHERE | F_SGROUP)) ? RTMI_SPH_WAVES(F) : ((F) & F_TRIS) ? RTMI_TRIS_WAVES : 6)
// Mesh variants share their per-workgroup tables (reference-tree nodes, materials) between more waves:
// workgroups of up to 512 lanes, two of which fill a CU's LDS with 16 waves' search regions.
#define RTMI_MAX_THREADS(F) (((F) & F_BVH) ? 512 : 256)
// Everything the trace kernel is told lives in ONE block of device memory (written by params_write_kernel, stream-ordered,
// just before the launch) and the kernel's only argument is its address.  By-value kernel arguments are all loaded in
// the kernel's first block and stay live from there: with ~150 dwords of them the list kernel parked 76-114 scalars in
// spill lanes (v_writelane / v_readlane at every use), the mesh kernel 285.  Read through the constant address space
// the fields arrive by scalar loads where they are used -- the hot loop's stay in SGPRs, the rest never occupy one.
struct RenderParams {
  SceneDev sc;
  FrameDev fr;
  LaunchCfg lc;
  uint32_t *states;
  float *out;
  uint32_t *ray_counts;
  unsigned long long *counters;
};
__global__ void params_write_kernel(RenderParams p, RenderParams *dst) {
  if (threadIdx.x == 0 && blockIdx.x == 0) *dst = p;
}
size_t render_params_bytes() { return (sizeof(RenderParams) + 255) & ~(size_t)255; }

template <uint32_t F>
__global__ __launch_bounds__(RTMI_MAX_THREADS(F), RTMI_MIN_WAVES(F)) void render_kernel(const RenderParams *p) {
  // (global -> constant address space -> generic: the compiler's address-space inference turns every access through
  // kp back into a constant-address-space load, i.e. a scalar load that nothing in the kernel can clobber)
  const RenderParams *kp = (const RenderParams *)(const RT_CONSTANT RenderParams *)(uintptr_t)p;
  render_body<F>(kp->sc, kp->fr, kp->lc, kp->states, kp->out, kp->ray_counts, kp->counters);
}
template <uint32_t F>
__global__ __launch_bounds__(RTMI_MAX_THREADS(F), RTMI_MIN_WAVES(F)) void probe_kernel(const RenderParams *p) {
  const RenderParams *kp = (const RenderParams *)(const RT_CONSTANT RenderParams *)(uintptr_t)p;
  render_body<F>(kp->sc, kp->fr, kp->lc, kp->states, kp->out, kp->ray_counts, kp->counters);
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

// One workgroup: the head of the queue, pixel by pixel.  A pixel's chain is as long as its ray count, and a
// wave's query costs about as much per outlier ray it carries (measured: tools/mesh_stats.sh marks); outlier
// pixels also sit in tiles that are not outliers as tiles, where they would run their first thousand queries in a
// full wave.  So the head is made of PIXELS, from the whole frame: those whose probe count is >= pct64 % of the
// frame's largest get a wave each, >= pct32 % two per wave, >= pct16 % one per 16 lanes -- in a frame whose
// largest count is at least three times the mean, and as long as that takes no more than a quarter of the grid's
// waves and kHeadCap entries; else first the lightest class goes, then the heaviest pixels share waves two by two,
// then there is no head.  Listed pixels get bit 31 of their ray_counts word set: the ordinary queue passes them
// over.  meta[0] = head entries, meta[1], meta[2] = ends of the first two classes.
// (four small launches over the whole frame instead of one workgroup walking it three times: 0.6 ms -> tens of us)
// ws[0] largest count, ws[2..3] sum (64 bit), ws[4..6] class counts, ws[8..10] thresholds, ws[12..14] scatter cursors
__global__ __launch_bounds__(256) void head_scan_kernel(const uint32_t *__restrict__ ray_counts, int n_items,
                                                        uint32_t *__restrict__ ws) {
  uint32_t m = 0u;
  unsigned long long part = 0ull;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x)
    m = max(m, ray_counts[i]), part += ray_counts[i];
  for (int off = 32; off > 0; off >>= 1) m = max(m, (uint32_t)__shfl_down((int)m, off)), part += __shfl_down(part, off);
  if ((threadIdx.x & 63) == 0) {
    atomicMax(&ws[0], m);
    atomicAdd(reinterpret_cast<unsigned long long *>(ws + 2), part);  // (d_meta is 8-byte aligned: capi.hip)
  }
}
__global__ __launch_bounds__(256) void head_count_kernel(const uint32_t *__restrict__ ray_counts, int n_items,
                                                         uint32_t *__restrict__ ws, uint32_t pct64, uint32_t pct32,
                                                         uint32_t pct16) {
  const uint32_t cmax = ws[0];
  const unsigned long long total = *reinterpret_cast<const unsigned long long *>(ws + 2);
  const bool skewed = (unsigned long long)cmax * (unsigned long long)n_items >= 3ull * total && cmax >= 4u;
  if (!skewed) return;
  const uint32_t t64 = (cmax * pct64 + 99u) / 100u, t32 = (cmax * pct32 + 99u) / 100u, t16 = (cmax * pct16 + 99u) / 100u;
  uint32_t c0 = 0u, c1 = 0u, c2 = 0u;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
    const uint32_t c = ray_counts[i];
    if (c >= t16) (c >= t64 ? c0 : c >= t32 ? c1 : c2)++;
  }
  for (int off = 32; off > 0; off >>= 1) c0 += __shfl_down(c0, off), c1 += __shfl_down(c1, off), c2 += __shfl_down(c2, off);
  if ((threadIdx.x & 63) == 0) {
    if (c0) atomicAdd(&ws[4], c0);
    if (c1) atomicAdd(&ws[5], c1);
    if (c2) atomicAdd(&ws[6], c2);
  }
}
__global__ void head_plan_kernel(int n_items, uint32_t *__restrict__ meta, uint32_t *__restrict__ ws, uint32_t grid_waves,
                                 uint32_t pct64, uint32_t pct32, uint32_t pct16) {
  const uint32_t cmax = ws[0], none = 0xffffffffu;
  const unsigned long long total = *reinterpret_cast<const unsigned long long *>(ws + 2);
  const bool skewed = (unsigned long long)cmax * (unsigned long long)n_items >= 3ull * total && cmax >= 4u;
  uint32_t t64 = skewed ? (cmax * pct64 + 99u) / 100u : none, t32 = skewed ? (cmax * pct32 + 99u) / 100u : none,
           t16 = skewed ? (cmax * pct16 + 99u) / 100u : none;
  uint32_t a = ws[4], b = ws[5], c3 = ws[6];
  auto over = [&]() { return a + (b + 1u) / 2u + (c3 + 3u) / 4u > grid_waves / 4u || a + b + c3 > (uint32_t)kHeadCap; };
  if (over()) c3 = 0u, t16 = t32;
  if (over()) b += a, a = 0u, t64 = none;
  if (over()) b = 0u, t32 = none, t16 = none;
  meta[0] = a + b + c3, meta[1] = a, meta[2] = a + b;
  ws[8] = t64, ws[9] = t32, ws[10] = t16;
  ws[12] = 0u, ws[13] = a, ws[14] = a + b;  // scatter cursors
}
__global__ __launch_bounds__(256) void head_scatter_kernel(uint32_t *__restrict__ ray_counts, int n_items,
                                                           uint32_t *__restrict__ head, uint32_t *__restrict__ ws) {
  const uint32_t t64 = ws[8], t32 = ws[9], t16 = ws[10];
  if (t16 == 0xffffffffu) return;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n_items; i += gridDim.x * blockDim.x) {
    const uint32_t c = ray_counts[i];
    if (c >= t16) {
      head[atomicAdd(&ws[12 + (c >= t64 ? 0 : c >= t32 ? 1 : 2)], 1u)] = (uint32_t)i;
      ray_counts[i] = c | 0x80000000u;
    }
  }
}

// The queue's order is kept per QUARTER tile (16 work items = two rows of a tile).  List scenes: the tiles in
// longest-first order, each tile's quarters one after the other.
__global__ __launch_bounds__(256) void expand_order_kernel(const uint32_t *__restrict__ order, int n_tiles,
                                                            uint32_t *__restrict__ qmap) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n_tiles * 4) qmap[i] = order[i >> 2] * 4u + (uint32_t)(i & 3);
}
// Mesh scenes with a cost probe: a quarter's cost = the lane-steps the probe's mesh searches spent on its 16 pixels
// (+ their ray counts, so that it is never zero) ...
__global__ __launch_bounds__(256) void quarter_cost_kernel(const uint32_t *__restrict__ work, const uint32_t *__restrict__ rays,
                                                            int n_quarters, uint32_t *__restrict__ cost,
                                                            uint32_t *__restrict__ max_cost) {
  const int qd = blockIdx.x * blockDim.x + threadIdx.x;
  if (qd >= n_quarters) return;
  uint32_t s = 0;
  for (int i = 0; i < 16; i++) s += work[(size_t)qd * 16 + i] + (rays[(size_t)qd * 16 + i] & 0x7fffffffu);
  cost[qd] = s;
  atomicMax(max_cost, s);
}
// ... one workgroup sorts the quarters by cost (256 buckets, dearest first) ...
__global__ __launch_bounds__(1024) void quarter_sort_kernel(const uint32_t *__restrict__ cost, const uint32_t *__restrict__ max_cost,
                                                             int n_quarters, uint32_t *__restrict__ sorted) {
  __shared__ uint32_t bins[256];
  __shared__ uint32_t base[256];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) bins[i] = 0;
  __syncthreads();
  const uint32_t mx = *max_cost > 0 ? *max_cost : 1;
  for (int t = threadIdx.x; t < n_quarters; t += blockDim.x)
    atomicAdd(&bins[255u - (uint32_t)(((unsigned long long)cost[t] * 255ull) / mx)], 1u);
  __syncthreads();
  if (threadIdx.x == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; i++) base[i] = run, run += bins[i];
  }
  __syncthreads();
  for (int t = threadIdx.x; t < n_quarters; t += blockDim.x)
    sorted[atomicAdd(&base[255u - (uint32_t)(((unsigned long long)cost[t] * 255ull) / mx)], 1u)] = (uint32_t)t;
}
// ... and every 64 consecutive work items of the queue -- what the lanes of a wave pick up together -- are dealt one
// quarter from each quartile of that order (a snake: b, 2n - 1 - b, 2n + b, 4n - 1 - b): no wave starts on 64 rays of a
// dense tile (150-260 k cycles per query of a full wave against 40 k for the frame's typical mix; such waves WERE the
// frame's last per cent), every wave's first 64 pixels cost about the same, and the dearest quarters still go first.
__global__ __launch_bounds__(256) void snake_map_kernel(const uint32_t *__restrict__ sorted, int n_tiles,
                                                         uint32_t *__restrict__ qmap) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= n_tiles) return;
  const int n = n_tiles;
  qmap[4 * b + 0] = sorted[b];
  qmap[4 * b + 1] = sorted[2 * n - 1 - b];
  qmap[4 * b + 2] = sorted[2 * n + b];
  qmap[4 * b + 3] = sorted[4 * n - 1 - b];
}
hipError_t launch_quarter_order(const uint32_t *d_order, const uint32_t *d_work, const uint32_t *d_rays, int n_tiles,
                                uint32_t *d_qcost, uint32_t *d_qsorted, uint32_t *d_qmax, uint32_t *d_qmap, hipStream_t stream) {
  if (!d_work) {
    hipLaunchKernelGGL(expand_order_kernel, dim3((n_tiles * 4 + 255) / 256), dim3(256), 0, stream, d_order, n_tiles, d_qmap);
    return hipGetLastError();
  }
  hipError_t e = hipMemsetAsync(d_qmax, 0, sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(quarter_cost_kernel, dim3((n_tiles * 4 + 255) / 256), dim3(256), 0, stream, d_work, d_rays, n_tiles * 4,
                     d_qcost, d_qmax);
  hipLaunchKernelGGL(quarter_sort_kernel, dim3(1), dim3(1024), 0, stream, d_qcost, d_qmax, n_tiles * 4, d_qsorted);
  hipLaunchKernelGGL(snake_map_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, d_qsorted, n_tiles, d_qmap);
  return hipGetLastError();
}

// One thread per chain (wave r of SIMD s, c = r S + s), walked backwards so that every tile learns what follows it.
// The SIMD's k-th tile is rank k S + (k odd ? S - 1 - s : s); the wave's j-th tile is the SIMD's k = j R + (j odd ? R - 1 - r : r).
__global__ __launch_bounds__(256) void chain_link_kernel(const uint32_t *__restrict__ order, const uint32_t *__restrict__ cost,
                                                          int n_tiles, int S, int R, float scale, int32_t *__restrict__ first,
                                                          int32_t *__restrict__ next, uint32_t *__restrict__ fut) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= S * R) return;
  const int s = c % S, r = c / S;
  auto rank = [&](int j) -> int64_t {
    const int64_t k = (int64_t)j * R + ((j & 1) ? R - 1 - r : r);
    return k * S + ((k & 1) ? S - 1 - s : s);
  };
  int steps = 0;
  while (rank(steps) < n_tiles) steps++;
  float acc = 0.f;
  int32_t after = -1;
  for (int j = steps - 1; j >= 0; j--) {
    const uint32_t t = order[rank(j)];
    next[t] = after;
    fut[t] = (uint32_t)fminf(acc, 4.0e9f);
    acc += (float)cost[t] * scale;
    after = (int32_t)t;
  }
  first[c] = after;
}
hipError_t launch_chain_plan(const uint32_t *d_order, const uint32_t *d_cost, int n_tiles, int simds, int rounds, int spp,
                             int probe_spp, int32_t *d_first, int32_t *d_next, uint32_t *d_fut, hipStream_t stream) {
  const int n = simds * rounds;
  if (n <= 0) return hipSuccess;
  hipLaunchKernelGGL(chain_link_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_order, d_cost, n_tiles, simds, rounds,
                     (float)spp / (64.f * (float)probe_spp), d_first, d_next, d_fut);
  return hipGetLastError();
}

hipError_t launch_tile_order(uint32_t *d_ray_counts, int n_tiles, uint32_t *d_cost, uint32_t *d_meta,
                             uint32_t *d_order, uint32_t *d_head, uint32_t sparse_cap, int grid_waves, int outlier_x10,
                             const int head_pct[3], hipStream_t stream) {
  hipError_t e = hipMemsetAsync(d_meta, 0, 32 * sizeof(uint32_t), stream);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(tile_cost_kernel, dim3((n_tiles + 255) / 256), dim3(256), 0, stream, d_ray_counts, n_tiles, d_cost,
                     d_meta);
  hipLaunchKernelGGL(tile_order_kernel, dim3(1), dim3(1024), 0, stream, d_cost, d_meta, n_tiles, d_order, d_meta + 1,
                     sparse_cap, (uint32_t)outlier_x10);
  if (d_head) {
    const int p64 = head_pct[0], p32 = head_pct[1], p16 = head_pct[2];  // (rtmi_render_opts.head_pct)
    uint32_t *ws = d_meta + 16;  // 16 words of workspace behind the 16 of d_meta (zeroed with them)
    const int n_items = n_tiles * 64, blocks = n_items / 4096 > 0 ? (n_items / 4096 < 1024 ? n_items / 4096 : 1024) : 1;
    hipLaunchKernelGGL(head_scan_kernel, dim3(blocks), dim3(256), 0, stream, d_ray_counts, n_items, ws);
    hipLaunchKernelGGL(head_count_kernel, dim3(blocks), dim3(256), 0, stream, d_ray_counts, n_items, ws, (uint32_t)p64,
                       (uint32_t)p32, (uint32_t)p16);
    hipLaunchKernelGGL(head_plan_kernel, dim3(1), dim3(1), 0, stream, n_items, d_meta + 1, ws, (uint32_t)grid_waves,
                       (uint32_t)p64, (uint32_t)p32, (uint32_t)p16);
    hipLaunchKernelGGL(head_scatter_kernel, dim3(blocks), dim3(256), 0, stream, d_ray_counts, n_items, d_head, ws);
  }
  return hipGetLastError();
}

// ------------------------------------------------------------------ self-test
// Runs every shortened operation against the expression it replaces on all 2^32 inputs:
// bad[0] rcp_rn(x) vs 1.0f / x inside rcp_rn's domain (must be 0), bad[1] outside it
// (informative), bad[2] rng_pm1_of(x) vs the reference's uniform(-1, 1) expression, bad[3]
// rng_01_of(x) vs uniform(0, 1) (both must be 0), bad[4] sqrt_rn_core(x) vs sqrtf(x) on its domain (must be 0),
// bad[5] div_rn_core vs a / l on 2^32 triples of sampler draws (must be 0), bad[6] the same with a single
// correction (informative).
__device__ __forceinline__ float ref_uniform(uint32_t x) { return (float)x * 2.3283064e-10f + 1.16415322e-10f; }
__device__ __forceinline__ float ref_range(float mn, float mx, uint32_t x) { return ref_uniform(x) * (mx - mn) + mn; }
__device__ __forceinline__ uint32_t selftest_mix(uint64_t &s) {  // splitmix64
  s += 0x9e3779b97f4a7c15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}
__global__ __launch_bounds__(256) void arithmetic_selftest(unsigned long long *bad, float mn1, float mx1, float mn0,
                                                           float mx0) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long in_domain = 0, outside = 0, pm1 = 0, u01 = 0, sq = 0, dv = 0, dv1 = 0;
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
    // sqrt_rn_core against sqrtf on its whole domain
    if (x >= SQRT_RN_LO && x < SQRT_RN_HI && __float_as_uint(sqrtf(x)) != __float_as_uint(sqrt_rn_core(x))) sq++;
  }
  // div_rn_core against the division on the Lambertian sampler's own operands (lambertian.cu:19-31): 2^32 triples
  // of draws, every eighth shrunk towards the corner (-1, -1, -1), every eighth towards the origin (tiny lengths)
  uint64_t st = 0x1234567ull + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x632be59bd9b4e019ull;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    uint32_t rx = selftest_mix(st), ry = selftest_mix(st), rz = selftest_mix(st);
    const uint32_t sh = selftest_mix(st);
    if ((i & 7) == 0) rx >>= (sh & 31), ry >>= ((sh >> 5) & 31), rz >>= ((sh >> 10) & 31);
    if ((i & 7) == 1)
      rx = 0x80000000u + (rx >> (8 + (sh & 15))), ry = 0x80000000u - (ry >> (8 + ((sh >> 4) & 15))),
      rz = 0x80000000u + (rz >> (8 + ((sh >> 8) & 15)));
    const float cx = rng_pm1_of(rx), cy = rng_pm1_of(ry), cz = rng_pm1_of(rz);
    const float sum = cx * cx + cy * cy + cz * cz;
    if (sum > BALL_S_MAX) continue;
    const float l = sqrtf(sum);
    if (!(l >= DIV3_RN_LO)) continue;
    const float y = rcp_rn(l);
    const float v[3] = {cx, cy, cz};
    for (int k = 0; k < 3; k++) {
      const float ref = v[k] / l;
      const float q0 = v[k] * y;
      if (__float_as_uint(ref) != __float_as_uint(div_rn_core(v[k], l, y))) dv++;
      if (__float_as_uint(ref) != __float_as_uint(__builtin_fmaf(__builtin_fmaf(-l, q0, v[k]), y, q0))) dv1++;
    }
  }
  if (in_domain) atomicAdd(&bad[0], in_domain);
  if (outside) atomicAdd(&bad[1], outside);
  if (pm1) atomicAdd(&bad[2], pm1);
  if (u01) atomicAdd(&bad[3], u01);
  if (sq) atomicAdd(&bad[4], sq);
  if (dv) atomicAdd(&bad[5], dv);
  if (dv1) atomicAdd(&bad[6], dv1);
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

static LaunchCfg make_cfg(uint32_t variant, const SceneDev &sc, const FrameDev &fr, int threads, size_t *lds_bytes,
                          bool mats_in_lds = true) {
  LaunchCfg lc{};
  lc.threads = threads;
  lc.tile_order = nullptr;
  lc.lds_mats = mats_in_lds && sc.n_mats <= kLdsMats ? sc.n_mats : 0;
  lc.wide_ids = sc.n_mats > 256 ? 1 : sc.n_mats <= 16 ? 2 : 0;
  size_t off = ((size_t)lc.lds_mats * sizeof(MatRec) + 15) & ~(size_t)15;
  lc.stack_off = (int32_t)off;
  const size_t levels = (size_t)(fr.max_depth > 0 ? fr.max_depth : 1);
  // image-textured variants: one 32-bit word per level (material id, or the sampled texel)
  size_t stack = (variant & F_TEX) ? levels * threads * 4 : (lc.wide_ids == 2 ? (levels + 1) / 2 : levels) * threads * (lc.wide_ids == 1 ? 2 : 1);
  size_t noff = (off + stack + 15) & ~(size_t)15;
  lc.nodes_off = (int32_t)noff;
  lc.lds_nodes = (variant & F_BVH) ? (sc.n_nodes < kLdsNodes ? sc.n_nodes : kLdsNodes) : 0;
  size_t paoff = (noff + (size_t)lc.lds_nodes * sizeof(BvhNode) + 15) & ~(size_t)15;
  lc.paths_off = (int32_t)paoff;
  lc.lds_paths = (variant & F_BVH) ? (sc.n_leaf_paths < kLdsPaths ? sc.n_leaf_paths : kLdsPaths) : 0;
  size_t soff = (paoff + (size_t)lc.lds_paths * sizeof(int32_t) + 15) & ~(size_t)15;
  lc.mesh_off = (int32_t)soff;
  size_t poff = soff + ((variant & F_BVH) ? (size_t)(threads / 64) * kMeshWaveWords * sizeof(int) : 0);
  // the culled list scan: any list of four or more pairs; its records are staged in LDS up to kLdsPairs pairs
  const bool cull = (variant & F_TRIS) && sc.n_pairs >= kCullMinPairs && !plain_list_scan();
  const bool staged = cull && sc.n_pairs <= kLdsPairs;
  lc.pairs_off = staged ? (int32_t)poff : -1;
  size_t noff2 = (poff + (staged ? (size_t)sc.n_pairs * sizeof(PairPts) : 0) + 15) & ~(size_t)15;
  lc.nrm_off = staged ? (int32_t)noff2 : -1;
  size_t loff = (noff2 + (staged ? (size_t)sc.n_pairs * 2 * sizeof(TriNrm) : 0) + 15) & ~(size_t)15;
  const bool groups = (variant & F_SGROUP) && sc.n_sph_groups > 0;  // the grouped sphere scan shares its tests too
  const bool share = cull || groups;
  lc.list_off = share ? (int32_t)loff : -1;
  size_t coff = loff + (share ? (size_t)(threads / 64) * kListWaveWords(variant) * sizeof(int) : 0);
  lc.cand_off = groups ? (int32_t)coff : -1;
  *lds_bytes = coff + (groups ? (size_t)threads * (kSphCand * sizeof(uint16_t) + sizeof(int)) : 0);  // slots + a counter per lane
  // The grouped sphere scan is built for four waves per SIMD (108 VGPRs): four 256-lane workgroups per CU need 40 KiB
  // each at most.  A big material table (scenes/spheres.cu: one material per sphere, 15 KiB) that stands in the way
  // of the fourth workgroup stays in global memory -- measured on spheres 1024^2: 7.6 -> 8.5 Grays/s.
  if (groups && mats_in_lds && lc.lds_mats > 0 && *lds_bytes * (size_t)(1024 / threads) > 160 * 1024) {
    size_t without = 0;
    const LaunchCfg alt = make_cfg(variant, sc, fr, threads, &without, false);
    if (without * (size_t)(1024 / threads) <= 160 * 1024) {
      *lds_bytes = without;
      return alt;
    }
  }
  return lc;
}

template <uint32_t F>
static hipError_t launch_render_t(const SceneDev &sc, const FrameDev &fr, uint32_t *d_states, float *d_out,
                                  uint32_t *d_ray_counts, unsigned long long *d_counters, const SchedPlan &plan,
                                  bool probe, int blocks, int threads, const RenderTuning &tune, void *d_params,
                                  hipStream_t stream) {
  size_t lds = 0;
  LaunchCfg lc = make_cfg(F, sc, fr, threads, &lds);
  lc.tile_order = plan.tile_order;
  lc.visit_counts = plan.visit_counts;
  lc.sparse_items = plan.sparse_items;
  lc.head_list = plan.head_list;
  lc.probe_marks = plan.probe_marks;
  lc.sparse_stride = tune.sparse_stride;
  lc.exclusive = tune.exclusive;
  lc.probe_spp = plan.probe_spp;
  lc.promote = tune.promote;
  lc.lane_stride = tune.lane_stride > 0 ? tune.lane_stride : 1;
  lc.prio_tab = plan.prio_tab;  // (a first pass has one when it is long enough to gain from priorities: capi.hip)
  lc.tile_cost = plan.tile_cost;
  lc.rate_scale = 1.f / (64.f * (float)(plan.probe_spp > 0 ? plan.probe_spp : 1));
  lc.chain_next = (F & F_BVH) || lc.prio_tab == nullptr ? nullptr : plan.chain_next;
  lc.chain_fut = plan.chain_fut, lc.chain_first = plan.chain_first, lc.claims = plan.claims;
  lc.plan_simds = plan.plan_simds, lc.plan_rounds = plan.plan_rounds;
  lc.prio_every = tune.prio_every > 0 ? tune.prio_every : 16;
  {  // (render_body.h: the wave draws from the queue in batches; RTMI_FETCH_BATCH / RTMI_FETCH_BATCH_FIRST: A/B measurements)
    static const int batch_main = [] { const char *e = getenv("RTMI_FETCH_BATCH"); const int v = e ? atoi(e) : 16; return v < 1 ? 1 : v > 64 ? 64 : v; }();
    static const int batch_first = [] { const char *e = getenv("RTMI_FETCH_BATCH_FIRST"); const int v = e ? atoi(e) : 64; return v < 1 ? 1 : v > 64 ? 64 : v; }();
    // a first pass of a few samples: whole tiles; longest-first order: 16 (measured 4 / 16 / 64 on frames of 2.3 ...
    // 12.8 pixels per lane, NOTES.md); image order, or a first pass as long as a frame: the lanes that wait
    // (a pooled item waits for a lane of its wave: the longer a pixel takes, the fewer -- from 4,096 samples on, none)
    const int samples = fr.k_end - fr.k_begin > 0 ? fr.k_end - fr.k_begin : 1;
    const int by_length = 4096 / samples < 1 ? 1 : 4096 / samples;
    // image order (a frame too short to be scheduled, or a first pass as long as a frame): its last tiles weigh as much as
    // any, so batches only where the atomics would otherwise be the frame -- cornell 1024^2 x 16 spp: 8.0 ms a pixel at
    // a time, 5.2 ms four at a time; at 200 spp sixteen at a time cost 8 %
    // (below 32 spp -- where list frames are not scheduled -- 256 / samples: at 24 spp two pixels per atomic left a 2048^2
    // frame at the cursor's rate, 23.8 ms against 22.1 ms for 32 spp)
    const int per_atomic = samples < 32 ? 256 / samples : 64 / samples;
    const int image_batch = per_atomic < 1 ? 1 : per_atomic > 16 ? 16 : per_atomic;
    lc.fetch_batch = probe && samples <= 4 ? batch_first : plan.tile_order != nullptr ? (by_length < batch_main ? by_length : batch_main) : image_batch;
  }
  if (lds > 64 * 1024) {  // above the default dynamic-LDS limit: ask for it (160 KiB per CU on gfx950)
    hipError_t e = hipFuncSetAttribute(probe ? reinterpret_cast<const void *>(probe_kernel<F>)
                                             : reinterpret_cast<const void *>(render_kernel<F>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  {  // render_body.h reads LDS by byte offset (lds_byte): right only while the kernels declare no static LDS, in EVERY
     // build flavour (-DRTMI_STATS, -DRTMI_CHECK_MARGINS, A/B builds) -- asked of the code object once per variant
    static const hipError_t lds_ok = [] {
      hipFuncAttributes a{}, b{};
      hipError_t e = hipFuncGetAttributes(&a, reinterpret_cast<const void *>(render_kernel<F>));
      if (e == hipSuccess) e = hipFuncGetAttributes(&b, reinterpret_cast<const void *>(probe_kernel<F>));
      if (e != hipSuccess) return e;
      return a.sharedSizeBytes == 0 && b.sharedSizeBytes == 0 ? hipSuccess : hipErrorInvalidConfiguration;
    }();
    if (lds_ok != hipSuccess) return lds_ok;
  }
  RenderParams rp;
  rp.sc = sc, rp.fr = fr, rp.lc = lc;
  rp.states = d_states, rp.out = d_out, rp.ray_counts = d_ray_counts, rp.counters = d_counters;
  RenderParams *dp = reinterpret_cast<RenderParams *>(d_params);
  hipLaunchKernelGGL(params_write_kernel, dim3(1), dim3(64), 0, stream, rp, dp);
  if (probe) {
    hipLaunchKernelGGL(probe_kernel<F>, dim3(blocks), dim3(threads), lds, stream, (const RenderParams *)dp);
  } else {
    hipLaunchKernelGGL(render_kernel<F>, dim3(blocks), dim3(threads), lds, stream, (const RenderParams *)dp);
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
  X(F_SPHERE | F_SGROUP)                             \
  X(F_TRIS | F_SPHERE | F_SGROUP)                    \
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
                         int blocks, int threads, const RenderTuning &tune, void *d_params, hipStream_t stream) {
#define X(V) \
  if (variant == (uint32_t)(V)) \
    return launch_render_t<(V)>(sc, fr, d_states, d_out, d_ray_counts, d_counters, plan, probe, blocks, threads, tune, \
                                d_params, stream);
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
