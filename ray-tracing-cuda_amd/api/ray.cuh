// Ray(position, direction): stores the origin and the NORMALISED direction (ray.cu:8-11).
#pragma once
#include <glm/glm.hpp>
#include "rt_kinds.cuh"

class Ray {
  glm::vec3 origin_, dir_;

 public:
  RT_API Ray() {}
  RT_API Ray(glm::vec3 position, glm::vec3 direction) : origin_(position), dir_(glm::normalize(direction)) {}
  RT_API const glm::vec3 position() const { return origin_; }
  RT_API const glm::vec3 direction() const { return dir_; }
};
