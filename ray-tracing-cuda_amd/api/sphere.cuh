#pragma once
#include "hitable.cuh"
#include "material.cuh"

// Sphere(position, radius, material) — sphere.cu:7-9
class Sphere : public Hitable {
 public:
  RT_API Sphere(glm::vec3 position, double radius, Material *material_ptr)
      : Hitable(rtapi::H_SPHERE), radius_(radius), position_(position), material_ptr_(material_ptr) {}
  RT_API double radius() const { return radius_; }
  RT_API glm::vec3 position() const { return position_; }
  RT_API Material *material_ptr() const { return material_ptr_; }

 private:
  double radius_;
  glm::vec3 position_;
  Material *material_ptr_;
};
