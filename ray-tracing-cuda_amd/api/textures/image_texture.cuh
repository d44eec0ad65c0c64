#pragma once
#include <cuda_runtime.h>
#include <stdint.h>
#include "texture.cuh"

// What a cudaTextureObject_t handle of this build points at (device memory): the pitched
// RGBA8 image the scene uploaded.  Sampling (point filter, wrap, normalised coordinates —
// textures/image_texture.cu:9-38) happens inside librtmi.so.
struct RtImageDesc {
  const uint8_t *pixels;
  int height, width;
  uint64_t pitch;
};

class ImageTexture : public Texture {
 public:
  cudaTextureObject_t image_texture_;
  RT_API ImageTexture(cudaTextureObject_t image_texture) : Texture(rtapi::T_IMAGE), image_texture_(image_texture) {}

  // textures/image_texture.cu:17-38
  static cudaTextureObject_t CreateCudaTextureObj(uint8_t *dev_buffer, int height, int width, uint64_t pitch_in_bytes) {
    RtImageDesc h{dev_buffer, height, width, pitch_in_bytes};
    RtImageDesc *d = nullptr;
    if (hipMalloc((void **)&d, sizeof(h)) != hipSuccess) return 0;
    if (hipMemcpy(d, &h, sizeof(h), hipMemcpyHostToDevice) != hipSuccess) return 0;
    return (cudaTextureObject_t)(uintptr_t)d;
  }
};
