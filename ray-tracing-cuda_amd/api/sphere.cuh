#pragma once
#include "hitable.cuh"
#include "material.cuh"

// Sphere(position, radius, material): records centre (binary32), radius (binary64, as the
// reference keeps it) and the material.  The quadratic of sphere.cu:13-21 — float length/dot,
// double discriminant — is evaluated in librtmi.so with radius*radius precomputed on the host.
class Sphere : public Hitable {
  glm::vec3 centre_;
  double r_;
  Material *mat_;

 public:
  RT_API Sphere(glm::vec3 position, double radius, Material *material_ptr)
      : Hitable(rtapi::H_SPHERE), centre_(position), r_(radius), mat_(material_ptr) {}
  RT_API glm::vec3 position() const { return centre_; }
  RT_API double radius() const { return r_; }
  RT_API Material *material_ptr() const { return mat_; }
};
