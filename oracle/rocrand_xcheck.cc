// ORACLE — TEST INFRASTRUCTURE ONLY.
// Cross-check of oracle/xorwow.hpp against rocRAND's independent XORWOW
// (same recurrence and 2^67 sequence jump as cuRAND; different seed salts).
// Host-only: nothing here touches a GPU.  Prints "OK" and exits 0 on success.
#include <cstdio>
#include <cstdint>
#include <rocrand/rocrand_xorwow.h>
#include "xorwow.hpp"

// xorwow.hpp's init with rocRAND's salts instead of cuRAND's.
static orc::Xorwow init_rocrand_salts(uint64_t seed, uint64_t subsequence) {
  orc::Xorwow s;
  uint32_t s0 = (uint32_t)seed ^ 0x2c7f967fU;
  uint32_t s1 = (uint32_t)(seed >> 32) ^ 0xa03697cbU;
  uint32_t t0 = 1228688033U * s0;
  uint32_t t1 = 2073658381U * s1;
  s.d = 6615241u + t1 + t0;
  s.v[0] = 123456789u + t0;
  s.v[1] = 362436069u ^ t0;
  s.v[2] = 521288629u + t1;
  s.v[3] = 88675123u ^ t1;
  s.v[4] = 5783321u + t0;
  const auto &seq = orc::sequence_jump_matrices();
  for (int k = 0; subsequence != 0; k++, subsequence >>= 1)
    if (subsequence & 1) orc::jump_apply(seq[k], s.v);
  return s;
}

int main() {
  const uint64_t seeds[] = {0ull, 1024ull, 10086ull, 0x123456789abcdef0ull};
  const uint64_t subs[] = {0, 1, 2, 3, 4, 5, 63, 64, 1000, 65535, 65536, 1048575, 1048581, 16777215, (1ull << 32) + 7};
  int bad = 0, n = 0;
  for (uint64_t seed : seeds)
    for (uint64_t sub : subs) {
      rocrand_state_xorwow st;
      rocrand_init(seed, sub, 0, &st);
      orc::Xorwow mine = init_rocrand_salts(seed, sub);
      for (int i = 0; i < 32; i++) {
        unsigned a = rocrand(&st);
        unsigned b = orc::xorwow_next(&mine);
        n++;
        if (a != b) {
          if (bad < 5) std::printf("MISMATCH seed=%llu sub=%llu i=%d roc=%08x mine=%08x\n", (unsigned long long)seed, (unsigned long long)sub, i, a, b);
          bad++;
        }
      }
    }
  std::printf("%s %d/%d draws equal\n", bad ? "FAIL" : "OK", n - bad, n);
  return bad ? 1 : 0;
}
