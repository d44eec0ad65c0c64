#pragma once
#include "material.cuh"
#include "textures/constant_texture.cuh"
#include "textures/texture.cuh"

// Lambertian(color) / Lambertian(texture) — lambertian.cu:9-17
class Lambertian : public Material {
 public:
  Texture *texture_ptr_ = nullptr;
  ConstantTexture color_;
  bool use_constant_tex_ = false;
  RT_API Lambertian(glm::vec3 color) : Material(rtapi::M_LAMBERTIAN), color_(color), use_constant_tex_(true) {}
  RT_API Lambertian(Texture *texture_ptr) : Material(rtapi::M_LAMBERTIAN), texture_ptr_(texture_ptr) {}
  RT_API const Texture *texture_ptr() const { return use_constant_tex_ ? &color_ : texture_ptr_; }
};
