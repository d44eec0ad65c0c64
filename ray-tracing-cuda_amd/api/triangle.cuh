#pragma once
#include <glm/glm.hpp>
#include "hitable.cuh"

// Triangle(p[3], material) — triangle.cu:6-9
class Triangle : public Hitable {
 public:
  glm::vec3 p_[3];
  Material *material_ptr_;
  RT_API Triangle(glm::vec3 p[], Material *material_ptr) : Hitable(rtapi::H_TRIANGLE), material_ptr_(material_ptr) {
    for (int i = 0; i < 3; i++) p_[i] = p[i];
  }
};
