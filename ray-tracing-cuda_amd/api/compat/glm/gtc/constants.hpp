// glm::pi<T>() (see glm/glm.hpp in this directory for why this header exists).
#pragma once
#include "../glm.hpp"
namespace glm {
template <class T>
RTGLM_FN constexpr T pi() {
  return static_cast<T>(3.14159265358979323846264338327950288);
}
}  // namespace glm
