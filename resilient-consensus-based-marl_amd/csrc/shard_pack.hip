// Pack / unpack pass of the sharded instance's all-to-all transposes (parallel.ShardedConsensus).
//
// The reference gathers neighbour messages by Python list indexing inside one process
// (training/train_agents.py:129-130: `[critic_weights[i] for i in in_nodes[node]]`).  With ONE instance sharded
// over several GPUs that gather becomes an all-to-all of the message matrix [N agents][P_hid columns]: every rank
// sends (its agents) x (the peer's columns).  A column block of an agent-major matrix is a STRIDED box, RCCL wants
// contiguous buffers: this kernel is the one copy pass per direction that makes it so (round 2 did the same with
// .contiguous() + torch.cat + slice-assign = three passes).
//
//     dst[b][r][c] = src[b][r][c]      b < batches, r < rows, c < cols     (rows with row_mask[r] == 0 are skipped)
// 16-byte accesses when every stride and both bases allow it, 4-byte otherwise.  HBM-bound: 8 B per element.
#include "rcmarl_common.h"

namespace {

template <int VEC>
__global__ __launch_bounds__(256) void k_copy3d(const float* __restrict__ src, long src_batch, long ld_src,
                                                float* __restrict__ dst, long dst_batch, long ld_dst, int batches, int rows, int cols,
                                                const int* __restrict__ row_mask) {
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * VEC;
  if (c >= cols) return;
  for (int b = blockIdx.z; b < batches; b += gridDim.z)          // (any number of batches: the grid's z extent stops at 65535)
    for (int r = blockIdx.y; r < rows; r += gridDim.y) {
      if (row_mask && row_mask[r] == 0) continue;
      const float* s = src + b * src_batch + r * ld_src + c;
      float* d = dst + b * dst_batch + r * ld_dst + c;
      if (VEC == 4) {
        *reinterpret_cast<float4*>(d) = *reinterpret_cast<const float4*>(s);
      } else {
        *d = *s;
      }
    }
}

}  // namespace

RCMARL_EXPORT int rcmarl_copy3d(const float* src, long src_batch, long ld_src, float* dst, long dst_batch, long ld_dst,
                                int batches, int rows, int cols, const int* row_mask, void* stream) {
  if (!src || !dst || batches < 0 || rows < 0 || cols < 0 || ld_src < cols || ld_dst < cols) return RCMARL_ERR_ARG;
  if (batches == 0 || rows == 0 || cols == 0) return RCMARL_OK;
  const bool vec = cols % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && src_batch % 4 == 0 && dst_batch % 4 == 0 &&
                   ((uintptr_t)src & 15) == 0 && ((uintptr_t)dst & 15) == 0;
  const int per = vec ? 4 : 1;
  const dim3 grid((cols + 256 * per - 1) / (256 * per), rows < 65535 ? rows : 65535, batches < 65535 ? batches : 65535), block(256);
  if (vec) {
    RCMARL_LAUNCH(k_copy3d<4>, grid, block, 0, stream, src, src_batch, ld_src, dst, dst_batch, ld_dst, batches, rows, cols, row_mask);
  } else {
    RCMARL_LAUNCH(k_copy3d<1>, grid, block, 0, stream, src, src_batch, ld_src, dst, dst_batch, ld_dst, batches, rows, cols, row_mask);
  }
  return rcmarl_check_launch();
}
