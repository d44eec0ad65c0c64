// Experiment (GPU box): which short instruction sequences reproduce IEEE sqrtf / division bit for bit?
//   sqrt candidates on ALL positive binary32 patterns; division candidates on the Lambertian sampler's own operands
//   (a = a coordinate of an accepted point, l = sqrtf(x*x+y*y+z*z)), 2^36 random triples.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/exp/arith_candidates.hip -o gpurun_out/arith_candidates
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ float rcp_rn(float x) {
  const float r = __builtin_amdgcn_rcpf(x);
  const float e = __builtin_fmaf(-x, r, 1.0f);
  return __builtin_fmaf(e, r, r);
}
// A: rsq, one coupled correction
__device__ __forceinline__ float sqrt_a(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  const float g = x * y, h = 0.5f * y;
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
// B: rsq, Newton on both then residual correction (7 instructions)
__device__ __forceinline__ float sqrt_b(float x) {
  const float y = __builtin_amdgcn_rsqf(x);
  float g = x * y, h = 0.5f * y;
  const float r = __builtin_fmaf(-h, g, 0.5f);
  g = __builtin_fmaf(g, r, g);
  h = __builtin_fmaf(h, r, h);
  const float d = __builtin_fmaf(-g, g, x);
  return __builtin_fmaf(d, h, g);
}
// C: hardware sqrt + residual correction through rsq
__device__ __forceinline__ float sqrt_c(float x) {
  const float s = __builtin_amdgcn_sqrtf(x);
  const float h = 0.5f * __builtin_amdgcn_rsqf(x);
  const float d = __builtin_fmaf(-s, s, x);
  return __builtin_fmaf(d, h, s);
}
__global__ void sqrt_all(unsigned long long *bad) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  unsigned long long a = 0, b = 0, c = 0, n = 0;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 31); i += stride) {
    const float x = __uint_as_float((uint32_t)i);
    if (!(x >= 0x1p-100f && x < 0x1p100f)) continue;  // the domain the kernels would use the short form in
    const float ref = sqrtf(x);
    n++;
    a += __float_as_uint(sqrt_a(x)) != __float_as_uint(ref);
    b += __float_as_uint(sqrt_b(x)) != __float_as_uint(ref);
    c += __float_as_uint(sqrt_c(x)) != __float_as_uint(ref);
  }
  atomicAdd(&bad[0], a), atomicAdd(&bad[1], b), atomicAdd(&bad[2], c), atomicAdd(&bad[3], n);
}
__device__ __forceinline__ uint32_t mix(uint64_t &s) {  // splitmix64
  s += 0x9e3779b97f4a7c15ull;
  uint64_t z = s;
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return (uint32_t)((z ^ (z >> 31)) >> 16);
}
__device__ __forceinline__ float pm1(uint32_t x) { return __builtin_fmaf((float)x, 0x1p-31f, 0x1p-32f) - 1.0f; }
__global__ void div_domain(unsigned long long *bad, uint64_t per_thread, uint64_t seed) {
  uint64_t s = seed + ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) * 0x632be59bd9b4e019ull;
  unsigned long long one = 0, two = 0, n = 0, small = 0;
  for (uint64_t i = 0; i < per_thread; i++) {
    uint32_t rx = mix(s), ry = mix(s), rz = mix(s);
    if ((i & 7) == 0) rx >>= (mix(s) & 31), ry >>= (mix(s) & 31), rz >>= (mix(s) & 31);  // small draws: points near (-1,-1,-1)
    if ((i & 7) == 1) rx = 0x80000000u + (rx >> (8 + (mix(s) & 15))), ry = 0x80000000u - (ry >> (8 + (mix(s) & 15))),
                      rz = 0x80000000u + (rz >> (8 + (mix(s) & 15)));                      // points near the origin: tiny l
    const float x = pm1(rx), y = pm1(ry), z = pm1(rz);
    const float sum = x * x + y * y + z * z;
    if (sum > 1.00000011920928955078125f) continue;
    const float l = sqrtf(sum);
    if (!(l >= 0x1p-126f)) continue;
    if (l < 0x1p-10f) small++;
    const float r = rcp_rn(l);
    const float v[3] = {x, y, z};
    for (int k = 0; k < 3; k++) {
      const float ref = v[k] / l;
      const float q0 = v[k] * r;
      const float q1 = __builtin_fmaf(__builtin_fmaf(-l, q0, v[k]), r, q0);
      const float q2 = __builtin_fmaf(__builtin_fmaf(-l, q1, v[k]), r, q1);
      n++;
      one += __float_as_uint(q1) != __float_as_uint(ref);
      two += __float_as_uint(q2) != __float_as_uint(ref);
    }
  }
  atomicAdd(&bad[4], one), atomicAdd(&bad[5], two), atomicAdd(&bad[6], n), atomicAdd(&bad[7], small);
}
int main(int argc, char **argv) {
  const uint64_t per_thread = argc > 1 ? strtoull(argv[1], nullptr, 10) : (1ull << 16);
  unsigned long long *d, h[8];
  hipMalloc(&d, sizeof(h));
  hipMemset(d, 0, sizeof(h));
  hipLaunchKernelGGL(sqrt_all, dim3(4096), dim3(256), 0, 0, d);
  hipLaunchKernelGGL(div_domain, dim3(4096), dim3(256), 0, 0, d, per_thread, 0x1234567ull);
  hipDeviceSynchronize();
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("sqrt over %llu inputs in [2^-100, 2^100): mismatches A(rsq+1) %llu  B(rsq+newton+1) %llu  C(sqrt+rsq corr) %llu\n", h[3], h[0], h[1], h[2]);
  printf("a / l over %llu sampler operands (%llu with l < 2^-10): mismatches one correction %llu, two corrections %llu\n", h[6], h[7], h[4], h[5]);
  return 0;
}
