// Microbenchmark (GPU box): issue cost of v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 at 1, 2, 4, 6, 8 waves per SIMD.
// build: hipcc -O3 --offload-arch=gfx950 tools/exp/valu_rate.hip -o tools/exp/valu_rate ; run: tools/exp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
constexpr int kIters = 4096;

__global__ void k_fma(float *out, float a, float b) {
  float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  for (int i = 0; i < kIters; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                   "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a), "v"(b));
    }
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
__global__ void k_pk(float *out, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 av = {a, a}, bv = {b, b};
  for (int i = 0; i < kIters; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                   "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(av), "v"(bv));
    }
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
__global__ void k_pkmul(float *out, float a, float b) {
  f2 x0 = {(float)threadIdx.x, 1}, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
  f2 av = {a, a};
  for (int i = 0; i < kIters; i++) {
#pragma unroll
    for (int j = 0; j < 4; j++) {
      asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                   "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8"
                   : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(av));
    }
  }
  f2 s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <typename K>
static double run(K kern, int waves_per_simd, float *d_out) {
  int cus = 0;
  hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
  const int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
  hipEvent_t e0, e1;
  hipEventCreate(&e0), hipEventCreate(&e1);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001f, 1e-9f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(kern, dim3(blocks), dim3(256), 0, 0, d_out, 1.0000001f, 1e-9f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double insts_per_wave = (double)kIters * 32;
  // cycles per wave-instruction per SIMD at 2.4 GHz (all SIMDs busy alike): time * clock / (instructions per SIMD)
  return ms * 1e-3 * 2.4e9 / (insts_per_wave * waves_per_simd);
}
int main() {
  float *d_out;
  hipMalloc(&d_out, sizeof(float) * 256 * 256 * 8 * 4);
  printf("cycles per wave64 instruction per SIMD (at 2.4 GHz nominal; lower = faster)\n waves/SIMD   v_fma_f32   v_pk_fma_f32   v_pk_mul_f32\n");
  for (int w : {1, 2, 4, 6, 8})
    printf("   %d        %7.2f     %7.2f       %7.2f\n", w, run(k_fma, w, d_out), run(k_pk, w, d_out), run(k_pkmul, w, d_out));
  return 0;
}
