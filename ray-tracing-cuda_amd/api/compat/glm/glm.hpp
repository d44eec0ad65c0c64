// Minimal GLM work-alike: only what the reference's scene sources and this API layer use
// (vec2/vec3/vec4/mat4, dot/cross/normalize/length/reflect/refract/clamp/sqrt/min/max).
// Exists because GLM is an un-fetched submodule of the reference (.gitmodules:7-9) and the
// scene programs `#include <glm/glm.hpp>`; formulas are GLM's published scalar forms.
// Not part of the rendering product (librtmi.so does not include it).
#pragma once
#include <hip/hip_runtime.h>
#include <cmath>

#define RTGLM_FN __host__ __device__ inline

namespace glm {

struct vec2 {
  float x, y;
  RTGLM_FN vec2() : x(0), y(0) {}
  template <class A, class B>
  RTGLM_FN vec2(A a, B b) : x((float)a), y((float)b) {}
  template <class A>
  RTGLM_FN explicit vec2(A a) : x((float)a), y((float)a) {}
  RTGLM_FN float &operator[](int i) { return i == 0 ? x : y; }
  RTGLM_FN float operator[](int i) const { return i == 0 ? x : y; }
  RTGLM_FN static constexpr int length() { return 2; }
};

struct vec4;

struct vec3 {
  float x, y, z;
  RTGLM_FN vec3() : x(0), y(0), z(0) {}
  template <class A, class B, class C>
  RTGLM_FN vec3(A a, B b, C c) : x((float)a), y((float)b), z((float)c) {}
  template <class A>
  RTGLM_FN explicit vec3(A a) : x((float)a), y((float)a), z((float)a) {}
  RTGLM_FN explicit vec3(const vec4 &v);
  RTGLM_FN float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : z); }
  RTGLM_FN float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : z); }
  // GLM's member length() is the COMPONENT COUNT (scenes/spheres.cu:60 relies on it)
  RTGLM_FN static constexpr int length() { return 3; }
  RTGLM_FN vec3 &operator+=(const vec3 &o) { x += o.x, y += o.y, z += o.z; return *this; }
  RTGLM_FN vec3 &operator-=(const vec3 &o) { x -= o.x, y -= o.y, z -= o.z; return *this; }
  RTGLM_FN vec3 &operator*=(float s) { x *= s, y *= s, z *= s; return *this; }
  RTGLM_FN vec3 &operator/=(float s) { x /= s, y /= s, z /= s; return *this; }
};

struct vec4 {
  float x, y, z, w;
  RTGLM_FN vec4() : x(0), y(0), z(0), w(0) {}
  template <class A, class B, class C, class D>
  RTGLM_FN vec4(A a, B b, C c, D d) : x((float)a), y((float)b), z((float)c), w((float)d) {}
  template <class D>
  RTGLM_FN vec4(const vec3 &v, D d) : x(v.x), y(v.y), z(v.z), w((float)d) {}
  RTGLM_FN float &operator[](int i) { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
  RTGLM_FN float operator[](int i) const { return i == 0 ? x : (i == 1 ? y : (i == 2 ? z : w)); }
  RTGLM_FN static constexpr int length() { return 4; }
};
RTGLM_FN vec3::vec3(const vec4 &v) : x(v.x), y(v.y), z(v.z) {}

RTGLM_FN vec2 operator+(vec2 a, vec2 b) { return vec2(a.x + b.x, a.y + b.y); }
RTGLM_FN vec2 operator-(vec2 a, vec2 b) { return vec2(a.x - b.x, a.y - b.y); }
RTGLM_FN vec2 operator*(vec2 a, float s) { return vec2(a.x * s, a.y * s); }
RTGLM_FN vec2 operator*(float s, vec2 a) { return vec2(s * a.x, s * a.y); }

RTGLM_FN vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
RTGLM_FN vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
RTGLM_FN vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
RTGLM_FN vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
RTGLM_FN vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
RTGLM_FN vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
RTGLM_FN vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
RTGLM_FN vec4 operator+(vec4 a, vec4 b) { return vec4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
RTGLM_FN vec4 operator*(vec4 a, float s) { return vec4(a.x * s, a.y * s, a.z * s, a.w * s); }

RTGLM_FN float dot(vec3 a, vec3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RTGLM_FN vec3 cross(vec3 a, vec3 b) {
  return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
RTGLM_FN float length(vec3 v) { return sqrtf(dot(v, v)); }
RTGLM_FN float inversesqrt(float x) { return 1.0f / sqrtf(x); }
RTGLM_FN vec3 normalize(vec3 v) { return v * inversesqrt(dot(v, v)); }
RTGLM_FN vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
RTGLM_FN vec3 refract(vec3 I, vec3 N, float eta) {
  float d = dot(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  return k >= 0.0f ? (eta * I - (eta * d + sqrtf(k)) * N) : vec3(0.0f);
}
RTGLM_FN float min(float a, float b) { return (b < a) ? b : a; }
RTGLM_FN float max(float a, float b) { return (a < b) ? b : a; }
RTGLM_FN float clamp(float x, float lo, float hi) { return min(max(x, lo), hi); }
RTGLM_FN vec3 clamp(vec3 v, float lo, float hi) { return vec3(clamp(v.x, lo, hi), clamp(v.y, lo, hi), clamp(v.z, lo, hi)); }
RTGLM_FN vec3 sqrt(vec3 v) { return vec3(sqrtf(v.x), sqrtf(v.y), sqrtf(v.z)); }

// column-major 4x4, m[c] is a column (as in GLM)
struct mat4 {
  vec4 c[4];
  RTGLM_FN mat4() {}
  template <class A>
  RTGLM_FN explicit mat4(A d) {
    c[0] = vec4(d, 0, 0, 0), c[1] = vec4(0, d, 0, 0), c[2] = vec4(0, 0, d, 0), c[3] = vec4(0, 0, 0, d);
  }
  RTGLM_FN vec4 &operator[](int i) { return c[i]; }
  RTGLM_FN const vec4 &operator[](int i) const { return c[i]; }
};
RTGLM_FN vec4 operator*(const mat4 &m, const vec4 &v) { return m[0] * v.x + m[1] * v.y + m[2] * v.z + m[3] * v.w; }
RTGLM_FN mat4 operator*(const mat4 &a, const mat4 &b) {
  mat4 r;
  for (int j = 0; j < 4; j++) r[j] = a * b[j];
  return r;
}

}  // namespace glm
