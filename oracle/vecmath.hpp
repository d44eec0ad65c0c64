// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// Minimal float vector math restating the GLM closed forms the reference
// calls (GLM itself is an un-vendored submodule of the reference,
// /root/reference/.gitmodules:7-9, commit unrecorded => "parity unpinned"
// for GLM; the forms below are GLM's published scalar implementations).
//
// Everything here must be compiled with -ffp-contract=off so every
// operator is one IEEE-754 binary32 rounding, in the written order.
#pragma once
#include <cmath>

namespace orc {

struct vec3 {
  float x, y, z;
  vec3() : x(0), y(0), z(0) {}
  vec3(float a, float b, float c) : x(a), y(b), z(c) {}
  explicit vec3(float s) : x(s), y(s), z(s) {}
  float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
};

struct vec2 {
  float x, y;
  vec2() : x(0), y(0) {}
  vec2(float a, float b) : x(a), y(b) {}
};

inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
inline vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }

// glm::dot(vec3): tmp = a*b; (tmp.x + tmp.y) + tmp.z
inline float dot(vec3 a, vec3 b) {
  vec3 t = a * b;
  return (t.x + t.y) + t.z;
}
// glm::cross
inline vec3 cross(vec3 a, vec3 b) {
  return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// glm::length = sqrt(dot(v,v))
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
// glm::inversesqrt = 1/sqrt(x); glm::normalize = v * inversesqrt(dot(v,v))
inline float inversesqrt(float x) { return 1.0f / std::sqrt(x); }
inline vec3 normalize(vec3 v) { return v * inversesqrt(dot(v, v)); }
// glm::reflect = I - N * dot(N, I) * 2
inline vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
// glm::refract
inline vec3 refract(vec3 I, vec3 N, float eta) {
  float d = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if (k >= 0.0f) return eta * I - (eta * d + std::sqrt(k)) * N;
  return vec3(0.0f);
}
// glm::clamp(x, lo, hi) = min(max(x, lo), hi) with glm::max(a,b) = (a < b) ? b : a
// and glm::min(a,b) = (b < a) ? b : a (NaN propagates from x).
inline float clampf(float x, float lo, float hi) {
  float m = (x < lo) ? lo : x;
  return (hi < m) ? hi : m;
}

}  // namespace orc
