// What does ds_read_b64_tr_b16 deliver?  LDS holds u16 value = its own element index; every lane reads 8 bytes at a lane-chosen
// address; print, per lane, the four u16 it received.  Two address patterns:
//   P0: lane l reads at element 4*l            (64 consecutive 8-byte chunks)
//   P1: lane l reads row (l>>2)&3 (+4 per 16-lane group), cols 4*(l&3): a [4][16] row-major block per 16-lane group, row stride 16
// hipcc --offload-arch=gfx950 -O3 -o tr_b16_probe tr_b16_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned short* out, int pattern, int stride) {
  __shared__ unsigned short lds[4096];
  for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = (unsigned short)i;
  __syncthreads();
  const int l = threadIdx.x;
  int elem;
  if (pattern == 0) elem = 4 * l;
  else elem = ((l >> 4) * 4 + ((l >> 2) & 3)) * stride + 4 * (l & 3);
  s4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4*)(lds + elem));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)v[j];
}
int main() {
  unsigned short* d; hipMalloc(&d, 64 * 4 * 2);
  unsigned short h[256];
  for (int pat = 0; pat < 3; ++pat) {
    const int stride = pat == 2 ? 40 : 16;
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, pat == 0 ? 0 : 1, stride);
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("pattern %d (row stride %d elements): lane -> 4 x u16 element indices received\n", pat, stride);
    for (int l = 0; l < 64; ++l) printf("  lane %2d: %4d %4d %4d %4d%s", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3], (l & 3) == 3 ? "\n" : "   ");
  }
  return 0;
}
