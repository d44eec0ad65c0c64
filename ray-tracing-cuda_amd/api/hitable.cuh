#pragma once
// Base of every recorded scene object.  In the reference a Hitable answers ray queries through
// a virtual Hit() on the device; here intersection lives in librtmi.so (kernels.hip), and a
// Hitable is only what the flatten kernel needs to recognise an object the user's InitWorld
// kernel created with `new`: a type tag (rt_kinds.cuh).  No virtual table is involved, so the
// objects can be read back by plain layout.
#include <cuda_runtime.h>

#include "cuda_copyable.cuh"
#include "rt_kinds.cuh"

class Material;

class Hitable {
 public:
  int rt_kind_;
  RT_API explicit Hitable(int kind) : rt_kind_(kind) {}
};

#include "material.cuh"
#include "ray.cuh"

// Source-compatibility only: the closest-hit payload type of the reference (t, u, v are
// binary64 there; the kernel keeps t in binary32 unless the scene has spheres).
struct HitRecord {
  double t, u, v;
  glm::vec3 normal;
  Material *material_ptr;
};
