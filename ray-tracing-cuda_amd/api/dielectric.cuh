#pragma once
#include "material.cuh"

// Dielectric(attenuation, refractive_index) — dielectric.cu:10-14
class Dielectric : public Material {
 public:
  glm::vec3 attenuation_;
  double refractive_index_;
  RT_API Dielectric(glm::vec3 attenuation, double refractive_index)
      : Material(rtapi::M_DIELECTRIC), attenuation_(attenuation), refractive_index_(refractive_index) {}
};
