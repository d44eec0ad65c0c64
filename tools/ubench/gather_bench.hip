// Micro-benchmark (GPU box): what does a per-lane gather cost on one MI355X CU?
//
// The mesh search of the trace kernel is a chain of per-lane record fetches (tree nodes, faces)
// at unrelated addresses.  This program prices the candidate record shapes so the node/face
// layout is chosen from measurements (DESIGN.md "Mesh queries"):
//   global: every lane reads K consecutive 16-byte words of a random STRIDE-aligned record of a
//           table (table size picks L1 / L2 / Infinity-Cache residency);
//   lds   : the same from a table in LDS (ds_read_b128 / b64 / b32 at random addresses).
// Each lane walks a dependent chain (next index derived from the loaded data), as a tree search
// does, so latency is exposed unless other waves cover it; waves per SIMD is swept.
// Output: ns per wave-step per CU and lane-records per ns per CU.
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/gather_bench.hip -o ray-tracing-cuda_amd/build/gather_bench
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <vector>

#define CK(x)                                                                     \
  do {                                                                            \
    hipError_t e_ = (x);                                                          \
    if (e_ != hipSuccess) {                                                       \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                     \
      exit(1);                                                                    \
    }                                                                             \
  } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x) {
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

// K 16-byte loads per step from record (idx & mask) * stride_words16; VAL extra dependent VALU per step
template <int K, int VALU>
__global__ __launch_bounds__(256) void gather_global(const uint4 *__restrict__ table, uint32_t mask, uint32_t stride16,
                                                     int steps, uint32_t *__restrict__ out) {
  uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x + 1u);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    const uint4 *rec = table + (size_t)(idx & mask) * stride16;
    uint4 q[K];
#pragma unroll
    for (int k = 0; k < K; k++) q[k] = rec[k];
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < K; k++) h += q[k].x ^ q[k].y ^ q[k].z ^ q[k].w;
    float f = __uint_as_float((h & 0x007fffffu) | 0x3f800000u);
#pragma unroll
    for (int v = 0; v < VALU; v++) f = f * 1.0000001f + 0.25f;
    acc += h + __float_as_uint(f);
    idx = mix(idx + h);
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// narrow loads: W = 1 (dword), 2 (dwordx2) per step, K of them at consecutive addresses
template <int K, typename T>
__global__ __launch_bounds__(256) void gather_narrow(const T *__restrict__ table, uint32_t mask, uint32_t stride,
                                                     int steps, uint32_t *__restrict__ out) {
  uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x + 1u);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    const T *rec = table + (size_t)(idx & mask) * stride;
    uint32_t h = 0;
#pragma unroll
    for (int k = 0; k < K; k++) {
      T v = rec[k];
      const uint32_t *w = reinterpret_cast<const uint32_t *>(&v);
#pragma unroll
      for (unsigned i = 0; i < sizeof(T) / 4; i++) h += w[i];
    }
    acc += h;
    idx = mix(idx + h);
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

// groups of G lanes read the SAME record, lane g of the group reading 16-byte word g % K16 (cooperative fetch)
template <int G>
__global__ __launch_bounds__(256) void gather_group(const uint4 *__restrict__ table, uint32_t mask, uint32_t stride16,
                                                    int steps, uint32_t *__restrict__ out) {
  const uint32_t lane = threadIdx.x & 63u, grp = lane / G, gl = lane % G;
  uint32_t idx = mix((blockIdx.x * 256u + (threadIdx.x & ~63u)) + grp + 1u);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    const uint4 *rec = table + (size_t)(idx & mask) * stride16;
    const uint4 q = rec[gl % stride16];
    uint32_t h = q.x ^ q.y ^ q.z ^ q.w;
    // share across the group so that the chain stays group-uniform
#pragma unroll
    for (int o = 1; o < G; o <<= 1) h += __shfl_xor(h, o);
    acc += h;
    idx = mix(idx + h);
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

template <int BYTES>
__global__ __launch_bounds__(256) void gather_lds(const uint32_t *__restrict__ src, int lds_words, int steps,
                                                  uint32_t *__restrict__ out) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  for (int i = threadIdx.x; i < lds_words; i += 256) lds[i] = src[i];
  __syncthreads();
  const uint32_t recs = (uint32_t)lds_words * 4u / BYTES;  // power of two
  uint32_t idx = mix(blockIdx.x * 256u + threadIdx.x + 1u);
  uint32_t acc = 0;
  for (int s = 0; s < steps; s++) {
    const uint32_t r = idx & (recs - 1u);
    uint32_t h;
    if (BYTES == 16) {
      const uint4 q = *reinterpret_cast<const uint4 *>(lds + r * 4u);
      h = q.x ^ q.y ^ q.z ^ q.w;
    } else if (BYTES == 8) {
      const uint2 q = *reinterpret_cast<const uint2 *>(lds + r * 2u);
      h = q.x ^ q.y;
    } else if (BYTES == 64) {
      const uint4 *p = reinterpret_cast<const uint4 *>(lds + r * 16u);
      const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
      h = a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w ^ c.x ^ c.y ^ c.z ^ c.w ^ d.x ^ d.y ^ d.z ^ d.w;
    } else {
      h = lds[r];
    }
    acc += h;
    idx = mix(idx + h);
  }
  out[blockIdx.x * 256u + threadIdx.x] = acc;
}

static int g_cus = 256;

template <typename F>
static double time_ms(F launch, int reps = 3) {
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch();
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int r = 0; r < reps; r++) {
    CK(hipEventRecord(a));
    launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    if (ms < best) best = ms;
  }
  return best;
}

static void report(const char *name, int waves_per_simd, int steps, double ms, int lanes_per_rec = 1) {
  // one block of 256 = 1 wave per SIMD of a CU; blocks = cus * waves_per_simd
  const double wave_steps_per_cu = (double)waves_per_simd * 4.0 * steps;
  const double ns_per_wave_step = ms * 1e6 / wave_steps_per_cu;
  printf("%-44s w/simd %d  %8.3f ms  %8.1f ns/wave-step/CU  %7.3f records/ns/CU\n", name, waves_per_simd, ms,
         ns_per_wave_step, 64.0 / lanes_per_rec / ns_per_wave_step);
  fflush(stdout);
}

int main(int argc, char **argv) {
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  g_cus = prop.multiProcessorCount;
  printf("device %s, %d CUs, clock %d kHz\n", prop.name, g_cus, prop.clockRate);
  const int steps = 2000;
  uint32_t *d_out;
  CK(hipMalloc(&d_out, (size_t)g_cus * 8 * 256 * 4));
  const size_t table_bytes_max = 512ull << 20;
  uint4 *d_table;
  CK(hipMalloc(&d_table, table_bytes_max));
  {
    std::vector<uint32_t> h(table_bytes_max / 4);
    uint32_t x = 12345u;
    for (size_t i = 0; i < h.size(); i++) {
      x = x * 1664525u + 1013904223u;
      h[i] = x;
    }
    CK(hipMemcpy(d_table, h.data(), table_bytes_max, hipMemcpyHostToDevice));
  }
  const int wps_list[] = {1, 2, 4, 8};
  struct Tab {
    const char *name;
    size_t bytes;
  } tabs[] = {{"16KB(L1)", 16u << 10}, {"2.5MB(L2)", 2u << 20}, {"24MB(MALL)", 32u << 20}, {"512MB(HBM)", 512u << 20}};
  for (const Tab &t : tabs) {
    for (int wps : wps_list) {
      const int blocks = g_cus * wps;
      char nm[128];
#define RUN_G(K, V, STRIDE16)                                                                                    \
  {                                                                                                              \
    const uint32_t recs = (uint32_t)(t.bytes / (16 * (STRIDE16)));                                               \
    snprintf(nm, sizeof nm, "global %s K=%d x16B stride %dB valu %d", t.name, K, 16 * (STRIDE16), V);            \
    double ms = time_ms([&] {                                                                                    \
      hipLaunchKernelGGL((gather_global<K, V>), dim3(blocks), dim3(256), 0, 0, d_table, recs - 1u, (uint32_t)(STRIDE16), \
                         steps, d_out);                                                                          \
    });                                                                                                          \
    report(nm, wps, steps, ms);                                                                                  \
  }
      RUN_G(1, 0, 1)
      RUN_G(2, 0, 2)
      RUN_G(3, 0, 4)
      RUN_G(4, 0, 4)
      RUN_G(8, 0, 8)
      if (t.bytes == (2u << 20)) {
        RUN_G(4, 100, 4)
        RUN_G(8, 100, 8)
        RUN_G(4, 300, 4)
        {
          const uint32_t recs = (uint32_t)(t.bytes / 4);
          snprintf(nm, sizeof nm, "global %s 1 x dword", t.name);
          double ms = time_ms([&] {
            hipLaunchKernelGGL((gather_narrow<1, uint32_t>), dim3(blocks), dim3(256), 0, 0,
                               reinterpret_cast<const uint32_t *>(d_table), recs - 1u, 1u, steps, d_out);
          });
          report(nm, wps, steps, ms);
        }
        {
          const uint32_t recs = (uint32_t)(t.bytes / 32);
          snprintf(nm, sizeof nm, "global %s 4 x dwordx2 (32B rec)", t.name);
          double ms = time_ms([&] {
            hipLaunchKernelGGL((gather_narrow<4, uint2>), dim3(blocks), dim3(256), 0, 0,
                               reinterpret_cast<const uint2 *>(d_table), recs - 1u, 4u, steps, d_out);
          });
          report(nm, wps, steps, ms);
        }
        {
          const uint32_t recs = (uint32_t)(t.bytes / 64);
          snprintf(nm, sizeof nm, "global %s group of 4 lanes x 16B of one 64B rec", t.name);
          double ms = time_ms([&] {
            hipLaunchKernelGGL((gather_group<4>), dim3(blocks), dim3(256), 0, 0, d_table, recs - 1u, 4u, steps, d_out);
          });
          report(nm, wps, steps, ms, 4);
        }
        {
          const uint32_t recs = (uint32_t)(t.bytes / 128);
          snprintf(nm, sizeof nm, "global %s group of 8 lanes x 16B of one 128B rec", t.name);
          double ms = time_ms([&] {
            hipLaunchKernelGGL((gather_group<8>), dim3(blocks), dim3(256), 0, 0, d_table, recs - 1u, 8u, steps, d_out);
          });
          report(nm, wps, steps, ms, 8);
        }
      }
    }
  }
  // LDS: 32 KiB table per workgroup
  for (int wps : wps_list) {
    if (wps > 4) continue;  // 32 KiB x 4 blocks = 128 KiB
    const int blocks = g_cus * wps, words = 8192;
    char nm[128];
#define RUN_L(B)                                                                                              \
  {                                                                                                           \
    snprintf(nm, sizeof nm, "lds 32KiB random %dB", B);                                                       \
    double ms = time_ms([&] {                                                                                 \
      hipLaunchKernelGGL((gather_lds<B>), dim3(blocks), dim3(256), words * 4, 0,                              \
                         reinterpret_cast<const uint32_t *>(d_table), words, steps, d_out);                   \
    });                                                                                                       \
    report(nm, wps, steps, ms);                                                                               \
  }
    RUN_L(4)
    RUN_L(8)
    RUN_L(16)
    RUN_L(64)
  }
  return 0;
}
