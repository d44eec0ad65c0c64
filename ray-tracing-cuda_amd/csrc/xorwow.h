// Seed-compatible XORWOW generator (the generator behind the reference's
// curandState; call sites /root/reference/ray-tracing-cuda/utils.cu:43-47 and
// utils.cuh:22-27).  State lives in six registers per lane; the 48-byte
// curandState of the reference carries Box-Muller fields this path never reads.
#pragma once
#include <stdint.h>

#include <vector>

#include "vec.h"

namespace rtmi {

struct Rng {
  uint32_t d, v0, v1, v2, v3, v4;
};

// xorshift on the 160-bit vector + Weyl counter; returns v4 + d.
RT_HD uint32_t rng_next(Rng &s) {
  uint32_t t = s.v0 ^ (s.v0 >> 2);
  s.v0 = s.v1;
  s.v1 = s.v2;
  s.v2 = s.v3;
  s.v3 = s.v4;
#if defined(__HIP_DEVICE_COMPILE__) && !defined(RTMI_NO_XOR3)
  // the same four-term exclusive or with three of the terms in one v_bitop3_b32 (truth table 0x96 = a ^ b ^ c): the
  // compiler chains two-input xors, and a draw is ten instructions of which this saves one -- C2 -1.7 %, C5 shard -0.7 %
  s.v4 = __builtin_amdgcn_bitop3_b32(s.v4, s.v4 << 4, t, 0x96) ^ (t << 1);
#else
  s.v4 = (s.v4 ^ (s.v4 << 4)) ^ (t ^ (t << 1));
#endif
  s.d += 362437u;
  return s.v4 + s.d;
}

// curand_uniform: x * 2^-32 + 2^-33, in (0, 1].
RT_HD float rng_uniform(Rng &s) {
  uint32_t x = rng_next(s);
  return (float)x * 2.3283064e-10f + 1.16415322e-10f;
}

// CudaRandomFloat(min, max, state): (min, max].
RT_HD float rng_range(float mn, float mx, Rng &s) {
  float t = rng_uniform(s);
  return t * (mx - mn) + mn;
}

// ---------------------------------------------------------------- host side
// Sequence jump: subsequence n starts n * 2^67 draws into the stream.  The
// xorshift part is linear over GF(2); kJumpBits matrices A^(2^(67+k)) are
// derived once on the host by repeated squaring and uploaded for the init
// kernel.  A matrix is stored as the image of each of the 160 basis vectors
// (5 words each).
constexpr int kJumpBits = 40;
constexpr int kJumpWords = 160 * 5;

struct HostJump {
  std::vector<uint32_t> m;  // kJumpBits * 160 * 5
};

inline void host_step_v(uint32_t v[5]) {
  Rng s{0, v[0], v[1], v[2], v[3], v[4]};
  rng_next(s);
  v[0] = s.v0, v[1] = s.v1, v[2] = s.v2, v[3] = s.v3, v[4] = s.v4;
}

inline void host_matvec(const uint32_t *M, uint32_t v[5]) {
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int b = 0; b < 160; b++)
    if ((v[b >> 5] >> (b & 31)) & 1u)
      for (int k = 0; k < 5; k++) r[k] ^= M[b * 5 + k];
  for (int k = 0; k < 5; k++) v[k] = r[k];
}

inline const HostJump &host_jump_tables() {
  static HostJump J = [] {
    HostJump out;
    out.m.resize((size_t)kJumpBits * kJumpWords);
    std::vector<uint32_t> A(kJumpWords), B(kJumpWords);
    for (int b = 0; b < 160; b++) {
      uint32_t e[5] = {0, 0, 0, 0, 0};
      e[b >> 5] = 1u << (b & 31);
      host_step_v(e);
      for (int k = 0; k < 5; k++) A[b * 5 + k] = e[k];
    }
    auto square = [&]() {
      for (int b = 0; b < 160; b++) {
        uint32_t col[5];
        for (int k = 0; k < 5; k++) col[k] = A[b * 5 + k];
        host_matvec(A.data(), col);
        for (int k = 0; k < 5; k++) B[b * 5 + k] = col[k];
      }
      A.swap(B);
    };
    for (int i = 0; i < 67; i++) square();
    for (int j = 0; j < kJumpBits; j++) {
      for (int w = 0; w < kJumpWords; w++) out.m[(size_t)j * kJumpWords + w] = A[w];
      square();
    }
    return out;
  }();
  return J;
}

// Seeding of curand_init(seed, ., 0): salts, odd multipliers, Marsaglia's constants.
RT_HD Rng rng_seed(uint64_t seed) {
  uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
  uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  Rng s;
  s.d = 6615241u + t1 + t0;
  s.v0 = 123456789u + t0;
  s.v1 = 362436069u ^ t0;
  s.v2 = 521288629u + t1;
  s.v3 = 88675123u ^ t1;
  s.v4 = 5783321u + t0;
  return s;
}

inline Rng host_rng_init(uint64_t seed, uint64_t subsequence) {
  Rng s = rng_seed(seed);
  const HostJump &J = host_jump_tables();
  uint32_t v[5] = {s.v0, s.v1, s.v2, s.v3, s.v4};
  for (int k = 0; k < kJumpBits && subsequence; k++, subsequence >>= 1)
    if (subsequence & 1) host_matvec(&J.m[(size_t)k * kJumpWords], v);
  s.v0 = v[0], s.v1 = v[1], s.v2 = v[2], s.v3 = v[3], s.v4 = v[4];
  return s;
}

}  // namespace rtmi
