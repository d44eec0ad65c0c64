#pragma once
#include "texture.cuh"

// ConstantTexture(color) — textures/constant_texture.cu:7-9
class ConstantTexture : public Texture {
 public:
  glm::vec3 color_;
  RT_API ConstantTexture() : Texture(rtapi::T_CONSTANT) {}
  RT_API ConstantTexture(glm::vec3 color) : Texture(rtapi::T_CONSTANT), color_(color) {}
};
