// The unchanged scene sources call the CUDA runtime by name (scenes/*.cu: cudaMalloc,
// cudaMemcpy, cudaGetDeviceCount, ...).  This header spells those names over the HIP
// runtime so they compile; it is used by the scene programs only — librtmi.so and the
// kernels are written against HIP directly.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef hipError_t cudaError_t;
typedef hipError_t cudaError;
#define cudaSuccess hipSuccess
#define cudaMemcpyHostToDevice hipMemcpyHostToDevice
#define cudaMemcpyDeviceToHost hipMemcpyDeviceToHost
#define cudaMemcpyDeviceToDevice hipMemcpyDeviceToDevice
typedef unsigned long long cudaTextureObject_t;  // handle made by ImageTexture::CreateCudaTextureObj

template <class T>
inline cudaError_t cudaMalloc(T **p, size_t n) { return hipMalloc((void **)p, n); }
inline cudaError_t cudaFree(void *p) { return hipFree(p); }
inline cudaError_t cudaMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { return hipMemcpy(d, s, n, k); }
template <class T>
inline cudaError_t cudaMallocPitch(T **p, size_t *pitch, size_t w, size_t h) { return hipMallocPitch((void **)p, pitch, w, h); }
inline cudaError_t cudaMemcpy2D(void *d, size_t dp, const void *s, size_t sp, size_t w, size_t h, hipMemcpyKind k) {
  return hipMemcpy2D(d, dp, s, sp, w, h, k);
}
inline cudaError_t cudaDeviceSynchronize() { return hipDeviceSynchronize(); }
inline cudaError_t cudaGetLastError() { return hipGetLastError(); }
inline const char *cudaGetErrorString(cudaError_t e) { return hipGetErrorString(e); }
inline cudaError_t cudaGetDeviceCount(int *n) { return hipGetDeviceCount(n); }
inline cudaError_t cudaSetDevice(int d) { return hipSetDevice(d); }
