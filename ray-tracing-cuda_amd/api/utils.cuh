#pragma once
// Host driver interface of the reference (utils.cuh:18-57): the two entry points the scene
// programs call, plus the helpers they use directly.  Main / DistributedMain keep the
// reference's signature (note the argument order height, width) and ownership rules — the four
// device pointers are caller-owned globals that the callee allocates and never frees — and drive
// librtmi.so through the C ABI of include/rtmi.h (src/rt_api.hip).
//
// Run-time overrides (the reference bakes these in at compile time): RT_WIDTH, RT_HEIGHT,
// RT_SPP, RT_MAX_DEPTH (default 10 = TRACE_DEPTH_LIMIT, ray_tracing.cu:10), RT_SEED,
// RT_OUTPUT (default image.jpeg), RT_DUMP (also write the float32 H*W*3 frame to this path),
// RT_DIST_MODE=spp (DistributedMain only: the reference's sample split + sum-reduce instead of
// pixel-tile shards).
#include <cuda_runtime.h>
#include <curand_kernel.h>
#include <mpi.h>
#include <stdint.h>

#include <glm/glm.hpp>
#include <nvfunctional>
#include <string>
#include <tuple>
#include <vector>

#include "camera.cuh"
#include "hitable_list.cuh"
#include "ray.cuh"

std::string BaseName(const std::string &path);
std::string ParentPath(const std::string &path);

// CudaRandomFloat(min, max, state): curand_uniform(state) * (max - min) + min, range (min, max]
// (utils.cuh:22-27).  XORWOW step + Weyl counter on the array-of-structures state the scene sees.
__inline__ __device__ float CudaRandomFloat(float min, float max, curandState *state) {
  uint32_t t = state->v[0] ^ (state->v[0] >> 2);
  state->v[0] = state->v[1];
  state->v[1] = state->v[2];
  state->v[2] = state->v[3];
  state->v[3] = state->v[4];
  state->v[4] = (state->v[4] ^ (state->v[4] << 4)) ^ (t ^ (t << 1));
  state->d += 362437u;
  float u = (float)(state->v[4] + state->d) * 2.3283064e-10f + 1.16415322e-10f;
  return u * (max - min) + min;
}

void WriteImage(const std::vector<glm::vec3> &pixels, int height, int width, const std::string &path);

__host__ __device__ inline int GetWorkload(int rank, int world_size, int spp) {
  return spp / world_size + (int)(rank < (spp % world_size));
}

__host__ void Main(curandState **d_states, Camera **d_camera, HitableList **d_world, glm::vec3 **d_image,
                   nvstd::function<void(HitableList *world, Camera *camera)> init_world, int height, int width,
                   int spp);

__host__ void DistributedMain(curandState **d_states, Camera **d_camera, HitableList **d_world, glm::vec3 **d_image,
                              nvstd::function<void(HitableList *world, Camera *camera)> init_world, int height,
                              int width, int spp);
