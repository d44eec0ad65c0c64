#pragma once
#include "material.cuh"
#include "textures/texture.cuh"

// DiffuseLight(texture) — diffuse_light.cu:15-17
class DiffuseLight : public Material {
 public:
  Texture *texture_ptr_;
  RT_API DiffuseLight(Texture *texture_ptr) : Material(rtapi::M_DIFFUSE_LIGHT), texture_ptr_(texture_ptr) {}
};
