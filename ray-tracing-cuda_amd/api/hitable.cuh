#pragma once
struct HitRecord;
#include <cuda_runtime.h>
#include "cuda_copyable.cuh"
#include "material.cuh"
#include "ray.cuh"

// Closest-hit payload of the reference (hitable.cuh:13-17); kept for source compatibility.
struct HitRecord {
  double t, u, v;
  glm::vec3 normal;
  Material *material_ptr;
};

// Recorded hitable: a tag; intersection itself runs inside librtmi.so.
class Hitable {
 public:
  int rt_kind_;
  RT_API explicit Hitable(int kind) : rt_kind_(kind) {}
};
