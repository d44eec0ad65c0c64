// ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
//
// cuRAND-compatible XORWOW restated from the published curand_kernel.h
// algorithm.  cuRAND is a third-party dependency of the reference that is
// absent from /root/reference and from this image (CUDA toolkit, version
// unpinned by the reference's CMake); the reference holds no RNG golden
// vector, so the seed-scrambling constants are "parity unpinned".  What IS
// pinned (tests/test_oracle_rng.py): the recurrence and the 2^67 sequence
// jump agree with rocRAND's independent implementation of the same generator
// (/opt/rocm/include/rocrand/rocrand_xorwow.h, *_precomputed.h), which shares
// the recurrence but not the seed salts.
//
// Call sites restated:
//   curand_init(seed, idx, 0, &state[idx])   /root/reference/ray-tracing-cuda/utils.cu:43-47
//   curand_uniform(state)*(max-min)+min      /root/reference/ray-tracing-cuda/utils.cuh:22-27
#pragma once
#include <cstdint>
#include <cstring>
#include <vector>

namespace orc {

// Compact state: the 24 live bytes of the 48-byte curandState (Box-Muller
// fields are never read on this path).
struct Xorwow {
  uint32_t d;
  uint32_t v[5];
};

// One step of the xorshift part on v only (linear over GF(2)).
inline void xorwow_step_v(uint32_t v[5]) {
  uint32_t t = v[0] ^ (v[0] >> 2);
  v[0] = v[1];
  v[1] = v[2];
  v[2] = v[3];
  v[3] = v[4];
  v[4] = (v[4] ^ (v[4] << 4)) ^ (t ^ (t << 1));
}

// curand(state): xorshift step + Weyl sequence.
inline uint32_t xorwow_next(Xorwow *s) {
  xorwow_step_v(s->v);
  s->d += 362437u;
  return s->v[4] + s->d;
}

// curand_uniform: (0, 1].  2.3283064e-10f is exactly 2^-32 in binary32, so
// the multiply is exact and fusing it with the add cannot change the result.
inline float xorwow_uniform(Xorwow *s) {
  uint32_t x = xorwow_next(s);
  return (float)x * 2.3283064e-10f + (2.3283064e-10f / 2.0f);
}

// CudaRandomFloat(min, max, state)  (utils.cuh:22-27): range (min, max].
inline float random_float(float mn, float mx, Xorwow *s) {
  float t = xorwow_uniform(s);
  return t * (mx - mn) + mn;
}

// 160x160 GF(2) matrix stored as the image of each basis vector:
// m[bit][0..4] = step^k applied to e_bit  (same layout cuRAND/rocRAND use).
struct JumpMatrix {
  uint32_t m[160][5];
};

inline void jump_apply(const JumpMatrix &J, uint32_t v[5]) {
  uint32_t r[5] = {0, 0, 0, 0, 0};
  for (int i = 0; i < 5; i++)
    for (int j = 0; j < 32; j++)
      if (v[i] & (1u << j))
        for (int k = 0; k < 5; k++) r[k] ^= J.m[i * 32 + j][k];
  std::memcpy(v, r, sizeof(r));
}

inline JumpMatrix jump_one_step() {
  JumpMatrix J;
  for (int b = 0; b < 160; b++) {
    uint32_t v[5] = {0, 0, 0, 0, 0};
    v[b / 32] = 1u << (b % 32);
    xorwow_step_v(v);
    std::memcpy(J.m[b], v, sizeof(v));
  }
  return J;
}

inline JumpMatrix jump_square(const JumpMatrix &A) {
  JumpMatrix R;
  for (int b = 0; b < 160; b++) {
    uint32_t v[5];
    std::memcpy(v, A.m[b], sizeof(v));
    jump_apply(A, v);
    std::memcpy(R.m[b], v, sizeof(v));
  }
  return R;
}

// seq[k] = A^(2^(67+k)) : skipping 2^k subsequences of 2^67 draws each.
inline const std::vector<JumpMatrix> &sequence_jump_matrices() {
  static std::vector<JumpMatrix> seq = [] {
    std::vector<JumpMatrix> out;
    JumpMatrix J = jump_one_step();
    for (int i = 0; i < 67; i++) J = jump_square(J);
    for (int k = 0; k < 40; k++) {
      out.push_back(J);
      J = jump_square(J);
    }
    return out;
  }();
  return seq;
}

// curand_init(seed, subsequence, offset=0, state)
inline Xorwow xorwow_init(uint64_t seed, uint64_t subsequence) {
  Xorwow s;
  uint32_t s0 = (uint32_t)seed ^ 0xaad26b49u;
  uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xf7dcefddu;
  uint32_t t0 = 1099087573u * s0;
  uint32_t t1 = 2591861531u * s1;
  s.d = 6615241u + t1 + t0;
  s.v[0] = 123456789u + t0;
  s.v[1] = 362436069u ^ t0;
  s.v[2] = 521288629u + t1;
  s.v[3] = 88675123u ^ t1;
  s.v[4] = 5783321u + t0;
  // skipahead_sequence: v <- A^(subsequence * 2^67) v ; d is unchanged
  // because 2^67 * 362437 == 0 mod 2^32.
  const auto &seq = sequence_jump_matrices();
  for (int k = 0; subsequence != 0 && k < (int)seq.size(); k++, subsequence >>= 1)
    if (subsequence & 1) jump_apply(seq[k], s.v);
  return s;
}

}  // namespace orc
