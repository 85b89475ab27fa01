// Philox4x32-10 counter-based generator for the on-device rollout
// (rng_mode='device').  The CPU statement of the same stream is
// oracle/philox_np.py; both must agree word for word.
//   counter = (agent, step, episode, stream)   key = (seed_lo, seed_hi)
//   stream 0: action draws   stream 1: environment reset draws
#pragma once
#include <stdint.h>

struct RcPhilox { uint32_t r0, r1, r2, r3; };

__host__ __device__ static inline RcPhilox rc_philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                                            uint32_t k0, uint32_t k1) {
  const uint64_t M0 = 0xD2511F53ull, M1 = 0xCD9E8D57ull;
  for (int i = 0; i < 10; ++i) {
    const uint64_t p0 = M0 * c0, p1 = M1 * c2;
    const uint32_t hi0 = (uint32_t)(p0 >> 32), lo0 = (uint32_t)p0;
    const uint32_t hi1 = (uint32_t)(p1 >> 32), lo1 = (uint32_t)p1;
    const uint32_t n0 = hi1 ^ c1 ^ k0, n1 = lo1, n2 = hi0 ^ c3 ^ k1, n3 = lo0;
    c0 = n0; c1 = n1; c2 = n2; c3 = n3;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  RcPhilox o; o.r0 = c0; o.r1 = c1; o.r2 = c2; o.r3 = c3;
  return o;
}

__host__ __device__ static inline int rc_mulhi_range(uint32_t r, int n) { return (int)(((uint64_t)r * (uint64_t)n) >> 32); }
__host__ __device__ static inline float rc_u01(uint32_t r) { return (float)(r >> 8) * (1.0f / 16777216.0f); }
