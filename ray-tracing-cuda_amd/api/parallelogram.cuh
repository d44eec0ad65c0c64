#pragma once
#include <glm/glm.hpp>
#include "hitable_list.cuh"

// Parallelogram(p[3], material): p3 = p1 + p2 - p0 (parallelogram.cu:10-15).  Only the three
// given corners are recorded; librtmi.so derives the fourth with the same operation.
class Parallelogram : public Hitable {
 public:
  glm::vec3 p_[3];
  Material *material_ptr_;
  RT_API Parallelogram(glm::vec3 p[3], Material *material_ptr)
      : Hitable(rtapi::H_PARALLELOGRAM), material_ptr_(material_ptr) {
    for (int i = 0; i < 3; i++) p_[i] = p[i];
  }
};
