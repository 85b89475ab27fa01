// Shuffle permutations of the adversaries' mini-batch fits, generated on the device.
//
// Keras' fit(batch_size, epochs) reshuffles the rows every epoch with TensorFlow's RNG
// (agents/adversarial_CAC_agents.py:38-41,131-135,163-165,237-253); that stream cannot be reproduced
// outside TensorFlow, so this implementation (and its oracle, oracle/rpbcac_oracle.py::ShuffleStream)
// DEFINES the shuffle, counter-based so that it needs no serial state:
//     the call-th mini-batch fit of seed s, epoch e, visits the rows in the order that sorts
//     key(p) = Philox4x32-10(counter = (p, e, call, stream 2), key = seed).word0     (ties -> lower p first)
// One workgroup sorts one permutation of 64-bit items (key << 32 | p).  The keys are uniform 32-bit hashes, so a BUCKET sort does
// it in O(B): histogram of the top 8 bits (LDS atomics), exclusive scan, scatter into the bucket ranges (arrival order arbitrary),
// then one thread per bucket insertion-sorts its ~B/256 items -- the result is the unique ascending order, bit-identical to the
// bitonic sort of rounds 1-3, which moved every item 78 times through LDS (0.86 ms per call at 512 seeds x 10 epochs x 3000 rows:
// 26 ms per block of BASELINE configs[1] batched, on the adversaries' critical path).
#include "rcmarl_common.h"
#include "rcmarl_rng.h"

namespace {

constexpr int SH_BUCKETS = 256;

__global__ __launch_bounds__(256) void k_shuffle_perms(const unsigned long long* __restrict__ seeds,
                                                       const int* __restrict__ calls, int n, int epochs, int B,
                                                       int* __restrict__ perm) {
  RCMARL_DYN_SMEM(unsigned long long, items);                    // [B] bucketed items, then sorted in place
  __shared__ int hist[SH_BUCKETS], start[SH_BUCKETS + 1], cursor[SH_BUCKETS];
  const int e = blockIdx.x, q = blockIdx.y, s = blockIdx.z, t = threadIdx.x;
  const unsigned long long seed = seeds[s];
  const uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
  const uint32_t call = (uint32_t)calls[q];
  hist[t] = 0;
  cursor[t] = 0;
  __syncthreads();
  for (int i = t; i < B; i += 256) atomicAdd(&hist[rc_philox4x32_10((uint32_t)i, (uint32_t)e, call, 2u, k0, k1).r0 >> 24], 1);
  __syncthreads();
  // exclusive scan of the 256 counts (Hillis-Steele in LDS; start[256] = B)
  start[t + 1] = hist[t];
  if (t == 0) start[0] = 0;
  __syncthreads();
  for (int d = 1; d < SH_BUCKETS; d <<= 1) {
    const int v = t + 1 > d ? start[t + 1 - d] : 0;
    __syncthreads();
    if (t + 1 > d) start[t + 1] += v;
    __syncthreads();
  }
  for (int i = t; i < B; i += 256) {
    const uint32_t key = rc_philox4x32_10((uint32_t)i, (uint32_t)e, call, 2u, k0, k1).r0;
    const int b = key >> 24;
    items[start[b] + atomicAdd(&cursor[b], 1)] = ((unsigned long long)key << 32) | (uint32_t)i;
  }
  __syncthreads();
  {                                                               // bucket t: insertion sort of items[start[t] .. start[t+1])
    const int lo = start[t], hi = start[t + 1];
    for (int i = lo + 1; i < hi; ++i) {
      const unsigned long long v = items[i];
      int j = i - 1;
      while (j >= lo && items[j] > v) { items[j + 1] = items[j]; --j; }
      items[j + 1] = v;
    }
  }
  __syncthreads();
  int* out = perm + (((long)s * n + q) * epochs + e) * B;
  for (int i = t; i < B; i += 256) out[i] = (int)(uint32_t)items[i];
}

}  // namespace

RCMARL_EXPORT int rcmarl_shuffle_perms(const void* seeds, const int* calls, int n, int epochs, int B, int* perm, int S,
                                       void* stream) {
  if (!seeds || !calls || !perm || n <= 0 || epochs <= 0 || B <= 0 || S <= 0) return RCMARL_ERR_ARG;
  if (B > 8192) return RCMARL_ERR_UNSUPPORTED;          // 64 KiB of LDS per permutation
  const dim3 grid(epochs, n, S), block(256);
  const size_t smem = (size_t)B * 8;
  static const bool ok = rc_want_lds(k_shuffle_perms, (size_t)8192 * 8, 48 * 1024);
  if (!ok) return RCMARL_ERR_LAUNCH;
  RCMARL_LAUNCH(k_shuffle_perms, grid, block, smem, stream, (const unsigned long long*)seeds, calls, n, epochs, B, perm);
  return rcmarl_check_launch();
}
