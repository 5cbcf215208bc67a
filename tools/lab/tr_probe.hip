// tr_probe.hip -- what does ds_read_b64_tr_b16 return?  LDS holds its own element index (16-bit); every lane supplies the address of
// 4 contiguous elements: lane t (= lane & 15) of 16-lane group g points at row (t >> 2), columns 4 (t & 3) .. +3 of the group's
// [4][RS] block (row stride RS elements).  Prints, per lane, the four 16-bit values it received as (row, col) of its group's block.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out, int rs) {
  __shared__ short lds[8192];
  for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (short)i;
  __syncthreads();
  const int lane = threadIdx.x, t = lane & 15, g = lane >> 4;
  const short* a = lds + g * 1024 + (t >> 2) * rs + 4 * (t & 3);
  s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)a);
  for (int j = 0; j < 4; ++j) out[lane * 4 + j] = v[j];
}
int main() {
  short* d;
  hipMalloc(&d, 64 * 4 * 2);
  for (int rs : {16, 64}) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, rs);
    short h[256];
    hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    printf("row stride %d elements\n", rs);
    for (int l = 0; l < 64; ++l) {
      printf("lane %2d:", l);
      for (int j = 0; j < 4; ++j) {
        const int idx = h[l * 4 + j] - (l >> 4) * 1024;
        printf("  (r%d,c%2d)", idx / rs, idx % rs);
      }
      printf("\n");
    }
  }
  return 0;
}
