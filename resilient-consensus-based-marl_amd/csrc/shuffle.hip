// Shuffle permutations of the adversaries' mini-batch fits, generated on the device.
//
// Keras' fit(batch_size, epochs) reshuffles the rows every epoch with TensorFlow's RNG
// (agents/adversarial_CAC_agents.py:38-41,131-135,163-165,237-253); that stream cannot be reproduced
// outside TensorFlow, so this implementation (and its oracle, oracle/rpbcac_oracle.py::ShuffleStream)
// DEFINES the shuffle, counter-based so that it needs no serial state:
//     the call-th mini-batch fit of seed s, epoch e, visits the rows in the order that sorts
//     key(p) = Philox4x32-10(counter = (p, e, call, stream 2), key = seed).word0     (ties -> lower p first)
// One workgroup sorts one permutation: 64-bit (key << 32 | p) bitonic sort in LDS.
#include "rcmarl_common.h"
#include "rcmarl_rng.h"

namespace {

__global__ __launch_bounds__(256) void k_shuffle_perms(const unsigned long long* __restrict__ seeds,
                                                       const int* __restrict__ calls, int n, int epochs, int B,
                                                       int np2, int* __restrict__ perm) {
  RCMARL_DYN_SMEM(unsigned long long, keys);
  const int e = blockIdx.x, q = blockIdx.y, s = blockIdx.z;
  const unsigned long long seed = seeds[s];
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t call = (uint32_t)calls[q];
  for (int i = threadIdx.x; i < np2; i += blockDim.x) {
    unsigned long long v = ~0ull;                       // padding sorts to the end
    if (i < B) v = ((unsigned long long)rc_philox4x32_10((uint32_t)i, (uint32_t)e, call, 2u, k0, k1).r0 << 32) | (uint32_t)i;
    keys[i] = v;
  }
  __syncthreads();
  for (int k = 2; k <= np2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = threadIdx.x; i < np2; i += blockDim.x) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = keys[i], b = keys[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { keys[i] = b; keys[ixj] = a; }
        }
      }
      __syncthreads();
    }
  int* out = perm + (((long)s * n + q) * epochs + e) * B;
  for (int i = threadIdx.x; i < B; i += blockDim.x) out[i] = (int)(uint32_t)keys[i];
}

}  // namespace

RCMARL_EXPORT int rcmarl_shuffle_perms(const void* seeds, const int* calls, int n, int epochs, int B, int* perm, int S,
                                       void* stream) {
  if (!seeds || !calls || !perm || n <= 0 || epochs <= 0 || B <= 0 || S <= 0) return RCMARL_ERR_ARG;
  int np2 = 2;
  while (np2 < B) np2 <<= 1;
  if (np2 > 8192) return RCMARL_ERR_UNSUPPORTED;        // 64 KiB of LDS per permutation
  const dim3 grid(epochs, n, S), block(256);
  RCMARL_LAUNCH(k_shuffle_perms, grid, block, (size_t)np2 * 8, stream, (const unsigned long long*)seeds, calls, n, epochs,
                B, np2, perm);
  return rcmarl_check_launch();
}
