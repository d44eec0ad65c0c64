// glm::rotateX / rotateY / rotateZ for vec3 (glm/gtx/rotate_vector.inl closed forms).
#pragma once
#include "../glm.hpp"
namespace glm {
RTGLM_FN vec3 rotateX(const vec3 &v, float angle) {
  vec3 r = v;
  const float c = cosf(angle), s = sinf(angle);
  r.y = v.y * c - v.z * s;
  r.z = v.y * s + v.z * c;
  return r;
}
RTGLM_FN vec3 rotateY(const vec3 &v, float angle) {
  vec3 r = v;
  const float c = cosf(angle), s = sinf(angle);
  r.x = v.x * c + v.z * s;
  r.z = -v.x * s + v.z * c;
  return r;
}
RTGLM_FN vec3 rotateZ(const vec3 &v, float angle) {
  vec3 r = v;
  const float c = cosf(angle), s = sinf(angle);
  r.x = v.x * c - v.y * s;
  r.y = v.x * s + v.y * c;
  return r;
}
}  // namespace glm
