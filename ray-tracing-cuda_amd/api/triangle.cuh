#pragma once
#include <glm/glm.hpp>
#include "hitable.cuh"

// Triangle(p[3], material): a lone triangle in the world list.  Edges and the unit normal that
// TriangleHit (utils.cu:54-55,79) derives per ray are precomputed once by librtmi.so.
class Triangle : public Hitable {
 public:
  Material *material_ptr_;
  glm::vec3 p_[3];
  RT_API Triangle(glm::vec3 p[], Material *material_ptr) : Hitable(rtapi::H_TRIANGLE), material_ptr_(material_ptr) {
    p_[0] = p[0], p_[1] = p[1], p_[2] = p[2];
  }
};
