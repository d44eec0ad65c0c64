#pragma once
class Material;
#include <curand_kernel.h>
#include <glm/glm.hpp>
#include "cuda_copyable.cuh"
#include "rt_kinds.cuh"

// Recorded material: a tag; subclasses add the constructor arguments (material.cuh:13-20).
class Material : public CudaCopyable {
 public:
  int rt_kind_;
  RT_API explicit Material(int kind) : rt_kind_(kind) {}
};
