// Exhaustive check (all 2^32 binary32 patterns) of a short correctly-rounded reciprocal against
// the compiler's IEEE division 1.0f / x on the GPU it runs on.  Prints, per binary exponent of x,
// how many inputs differ.  Build: hipcc -O2 --offload-arch=gfx950 -ffp-contract=off rcp_check.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstring>

__device__ __forceinline__ float fast_rcp(float x) {
  float r = __builtin_amdgcn_rcpf(x);
  float e = __builtin_fmaf(-x, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  e = __builtin_fmaf(-x, r, 1.0f);
  r = __builtin_fmaf(e, r, r);
  return r;
}

__global__ void check(unsigned long long *bad_by_exp, unsigned long long *bad1_by_exp) {
  const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
  for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < (1ull << 32); i += stride) {
    const uint32_t bits = (uint32_t)i;
    const float x = __uint_as_float(bits);
    const float ref = 1.0f / x;
    const float f = fast_rcp(x);
    // one-step variant
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    const float f1 = __builtin_fmaf(e, r, r);
    const uint32_t ex = (bits >> 23) & 0xffu;
    if (__float_as_uint(ref) != __float_as_uint(f) && !(ref != ref && f != f)) atomicAdd(&bad_by_exp[ex], 1ull);
    if (__float_as_uint(ref) != __float_as_uint(f1) && !(ref != ref && f1 != f1)) atomicAdd(&bad1_by_exp[ex], 1ull);
  }
}

int main() {
  unsigned long long *d, *d1, h[256], h1[256];
  hipMalloc(&d, sizeof(h)); hipMalloc(&d1, sizeof(h));
  hipMemset(d, 0, sizeof(h)); hipMemset(d1, 0, sizeof(h));
  hipLaunchKernelGGL(check, dim3(4096), dim3(256), 0, 0, d, d1);
  if (hipDeviceSynchronize() != hipSuccess) { printf("kernel failed\n"); return 1; }
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  hipMemcpy(h1, d1, sizeof(h1), hipMemcpyDeviceToHost);
  int lo = -1, hi = -1;
  for (int e = 0; e < 256; e++) {
    if (h[e] || h1[e]) printf("biased exponent %3d (|x| in [2^%d, 2^%d)): two-step differs on %llu inputs, one-step on %llu\n", e, e - 127, e - 126, h[e], h1[e]);
    if (!h[e]) { if (lo < 0) lo = e; hi = e; }
  }
  // longest clean run for the two-step variant
  int best_lo = 0, best_len = 0, cur_lo = 0, cur_len = 0;
  for (int e = 0; e < 256; e++) {
    if (!h[e]) { if (!cur_len) cur_lo = e; cur_len++; if (cur_len > best_len) best_len = cur_len, best_lo = cur_lo; } else cur_len = 0;
  }
  printf("two-step: exact for every x with biased exponent in [%d, %d], i.e. 2^%d <= |x| < 2^%d\n", best_lo, best_lo + best_len - 1, best_lo - 127, best_lo + best_len - 127);
  return 0;
}
