#pragma once
#include <glm/glm.hpp>
#include "../cuda_copyable.cuh"
#include "../rt_kinds.cuh"

// Recorded texture: a tag; ConstantTexture / ImageTexture add their payload.
class Texture : public CudaCopyable {
 public:
  int rt_kind_;
  RT_API explicit Texture(int kind) : rt_kind_(kind) {}
};
