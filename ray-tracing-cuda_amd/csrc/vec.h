// Host+device binary32 vector helpers for the trace path.
//
// The translation units including this header are compiled with
// -ffp-contract=off: each operator below is exactly one IEEE-754 rounding, in
// the order GLM's scalar code performs them, which is what makes the frame
// reproducible against the CPU checker bit for bit.  Closed forms follow the
// GLM functions the reference calls (normalize/dot/cross/length/reflect/refract,
// e.g. /root/reference/ray-tracing-cuda/ray.cu:10, metal.cu:18, dielectric.cu:28).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

#define RT_HD __host__ __device__ __forceinline__

namespace rtmi {

struct V3 {
  float x, y, z;
};

RT_HD V3 mk(float x, float y, float z) { return V3{x, y, z}; }
RT_HD V3 splat(float s) { return V3{s, s, s}; }
RT_HD V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
RT_HD V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
RT_HD V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }
RT_HD V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }
RT_HD V3 operator*(V3 a, float s) { return V3{a.x * s, a.y * s, a.z * s}; }
RT_HD V3 operator*(float s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }
RT_HD V3 operator/(V3 a, float s) { return V3{a.x / s, a.y / s, a.z / s}; }

// (x*x' + y*y') + z*z'
RT_HD float dot3(V3 a, V3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }
RT_HD V3 cross3(V3 a, V3 b) {
  return V3{a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y};
}
RT_HD float len3(V3 v) { return sqrtf(dot3(v, v)); }
// v * (1 / sqrt(v.v))
RT_HD V3 unit3(V3 v) { return v * (1.0f / sqrtf(dot3(v, v))); }
// I - N * dot(N, I) * 2
RT_HD V3 reflect3(V3 I, V3 N) { return I - N * dot3(N, I) * 2.0f; }
// k = 1 - eta^2 (1 - (N.I)^2); k < 0 -> 0 ; else eta*I - (eta*(N.I) + sqrt(k))*N
RT_HD V3 refract3(V3 I, V3 N, float eta) {
  float d = dot3(N, I);
  float k = 1.0f - eta * eta * (1.0f - d * d);
  if (k >= 0.0f) return eta * I - (eta * d + sqrtf(k)) * N;
  return splat(0.0f);
}
// min(max(x, lo), hi) with the "(a < b) ? b : a" selection GLM uses
RT_HD float clamp1(float x, float lo, float hi) {
  float m = (x < lo) ? lo : x;
  return (hi < m) ? hi : m;
}

}  // namespace rtmi
