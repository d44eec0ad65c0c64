#pragma once
#include "material.cuh"

// Metal(albedo[, fuzz]) — metal.cu:7-10 (fuzz is clamped to at most 1)
class Metal : public Material {
 public:
  glm::vec3 albedo_;
  float fuzz_;
  RT_API Metal(glm::vec3 albedo) : Metal(albedo, 0) {}
  RT_API Metal(glm::vec3 albedo, float fuzz) : Material(rtapi::M_METAL), albedo_(albedo), fuzz_(fuzz < 1 ? fuzz : 1) {}
};
