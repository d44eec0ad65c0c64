// curandState as this build keeps it: the six live words of cuRAND's XORWOW state
// (d, v[0..4]).  The scene sources only name the type (`curandState *d_states`) and pass
// pointers to CudaRandomFloat (utils.cuh); generation itself lives in csrc/xorwow.h.
#pragma once
#include <stdint.h>
struct curandStateXORWOW {
  uint32_t d;
  uint32_t v[5];
};
typedef curandStateXORWOW curandState;
typedef curandStateXORWOW curandState_t;
