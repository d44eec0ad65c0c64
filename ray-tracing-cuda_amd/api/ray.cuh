#pragma once
// Ray keeps its origin and a direction that the constructor normalises (the reference's
// Camera::RayAt normalises before constructing, so camera rays are normalised twice — the
// kernel reproduces that, quirk g2).  Scene programs do not build rays themselves; the class is
// here because the reference's headers expose it.
#include <glm/glm.hpp>
#include "rt_kinds.cuh"

class Ray {
 public:
  RT_API Ray() {}
  RT_API Ray(glm::vec3 position, glm::vec3 direction) : o_(position), d_(glm::normalize(direction)) {}
  RT_API const glm::vec3 position() const { return o_; }
  RT_API const glm::vec3 direction() const { return d_; }

 private:
  glm::vec3 o_, d_;
};
