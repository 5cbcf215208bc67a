// shadow_lab.hip -- how many other instructions of the SAME wave fit behind a v_mfma_f32_32x32x2_f32 (64 matrix-pipe cycles) for free?
// One wave per SIMD (256 workgroups x 4 waves), four independent accumulator chains, K pinned instructions of one kind between two
// consecutive MFMAs (asm volatile: the order in the binary is the order written here).  Prints shader cycles per MFMA for each K.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o shadow_lab shadow_lab.hip ; run: ./shadow_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int KIND> __device__ __forceinline__ void other(float& x, float& y, const float* lds) {
  if (KIND == 0) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(x) : "v"(y));
  if (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
  if (KIND == 2) asm volatile("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x) : "v"(y));
  if (KIND == 3) asm volatile("ds_read_b32 %0, %1" : "=v"(x) : "v"((unsigned)(size_t)lds) : "memory");
  if (KIND == 4) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(y));
  if (KIND == 5) asm volatile("s_nop 0");
  if (KIND == 6) { typedef float f2 __attribute__((ext_vector_type(2))); f2 v = {x, y}; asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(v)); x = v.x; }
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
template <int KIND, int K, int MF = 0>      // MF 0: v_mfma_f32_32x32x2_f32 (64 cycles)   1: v_mfma_f32_32x32x16_bf16 (32 cycles)
__global__ __launch_bounds__(256, 1) void shadow(const float* src, float* out, long long* clk, int iters) {
  __shared__ float lds[1024];
  lds[threadIdx.x] = src[threadIdx.x];
  __syncthreads();
  float a[4], b[4], x[16], y = src[threadIdx.x & 63];
  for (int i = 0; i < 4; ++i) { a[i] = src[(threadIdx.x + 256 * i) & 4095]; b[i] = src[(threadIdx.x * 7 + 256 * i + 13) & 4095]; }
  for (int i = 0; i < 16; ++i) x[i] = src[(threadIdx.x + i) & 4095];
  f32x16 c[4];
  for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
  bf16x8 xa[2];
  for (int q = 0; q < 2; ++q) for (int i = 0; i < 8; ++i) xa[q][i] = (__bf16)a[(i + q) & 3];
  const long long c0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        if (MF == 0) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(c[n]) : "v"(a[t]), "v"(b[(t + n) & 3]));
        else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(c[n]) : "v"(xa[t & 1]), "v"(xa[(t + n) & 1]));
#pragma unroll
        for (int k = 0; k < K; ++k) other<KIND>(x[(4 * t + n + k) & 15], y, lds + (threadIdx.x & 255));
      }
    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  const long long c1 = __builtin_readcyclecounter();
  float s = y;
  for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
  for (int i = 0; i < 16; ++i) s += x[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) clk[blockIdx.x] = c1 - c0;
}

template <int KIND, int K, int MF = 0> void run(const char* name, const float* src, float* out, long long* clk) {
  const int wgs = 256, iters = 4000;
  hipLaunchKernelGGL((shadow<KIND, K, MF>), dim3(wgs), dim3(256), 0, 0, src, out, clk, iters);
  CK(hipDeviceSynchronize());
  hipLaunchKernelGGL((shadow<KIND, K, MF>), dim3(wgs), dim3(256), 0, 0, src, out, clk, iters);
  CK(hipDeviceSynchronize());
  std::vector<long long> hc(wgs);
  CK(hipMemcpy(hc.data(), clk, wgs * 8, hipMemcpyDeviceToHost));
  double cyc = 0;
  for (auto v : hc) cyc += v;
  printf("%-22s %-14s K = %2d : %6.1f cycles per MFMA\n", MF ? "32x32x16_bf16" : "32x32x2_f32", name, K, cyc / wgs / (iters * 16.0));
}
template <int KIND> void sweep(const char* name, const float* src, float* out, long long* clk) {
  run<KIND, 0>(name, src, out, clk); run<KIND, 1>(name, src, out, clk); run<KIND, 2>(name, src, out, clk); run<KIND, 4>(name, src, out, clk);
  run<KIND, 6>(name, src, out, clk); run<KIND, 8>(name, src, out, clk); run<KIND, 12>(name, src, out, clk); run<KIND, 16>(name, src, out, clk);
}
int main() {
  std::vector<float> h(4096);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0f; }
  float *src, *out; long long* clk;
  CK(hipMalloc(&src, 4096 * 4)); CK(hipMalloc(&out, 256 * 256 * 4)); CK(hipMalloc(&clk, 256 * 8));
  CK(hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  sweep<0>("v_fma_f32", src, out, clk);
  sweep<1>("v_exp_f32", src, out, clk);
  sweep<2>("v_mov dpp", src, out, clk);
  sweep<3>("ds_read_b32", src, out, clk);
  sweep<4>("v_cndmask", src, out, clk);
  sweep<5>("s_nop", src, out, clk);
  sweep<6>("v_pk_fma_f32", src, out, clk);
  run<0, 0, 1>("v_fma_f32", src, out, clk); run<0, 1, 1>("v_fma_f32", src, out, clk); run<0, 2, 1>("v_fma_f32", src, out, clk);
  run<0, 4, 1>("v_fma_f32", src, out, clk); run<0, 8, 1>("v_fma_f32", src, out, clk); run<0, 16, 1>("v_fma_f32", src, out, clk);
  run<1, 2, 1>("v_exp_f32", src, out, clk); run<1, 4, 1>("v_exp_f32", src, out, clk); run<1, 8, 1>("v_exp_f32", src, out, clk);
  run<3, 2, 1>("ds_read_b32", src, out, clk); run<3, 4, 1>("ds_read_b32", src, out, clk); run<3, 8, 1>("ds_read_b32", src, out, clk);
  return 0;
}
