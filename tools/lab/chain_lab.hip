// chain_lab.hip -- how many INDEPENDENT accumulator chains does v_mfma_f32_16x16x4_f32 need per SIMD to run at rate, and does a
// ds_read_b128 operand stream (the row-resident kernels' inner loop: 2 reads per 8 MFMAs) cost matrix-pipe time?
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o chain_lab chain_lab.hip ; run: ./chain_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

template <int NACC, int LDS>
__global__ __launch_bounds__(256) void chain(const float* src, float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float tile[6144];
  for (int i = threadIdx.x; i < 6144; i += 256) tile[i] = src[i & 4095];
  __syncthreads();
  float b[8];
  for (int i = 0; i < 8; ++i) b[i] = src[(threadIdx.x * 7 + 256 * i + 13) & 4095];
  f32x4 c[NACC];
  for (int n = 0; n < NACC; ++n) c[n] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int lane = threadIdx.x & 63;
  const float* base = tile + (lane & 15) * 192 + (lane >> 4) * 4;
  float4 a = *reinterpret_cast<const float4*>(base);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      float4 nx = a;
      if (LDS) nx = *reinterpret_cast<const float4*>(base + (((t + 1) % 12) * 16));
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, b[(n) & 7], c[n], 0, 0, 0);
        c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, b[(n + 1) & 7], c[n], 0, 0, 0);
        c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, b[(n + 2) & 7], c[n], 0, 0, 0);
        c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, b[(n + 3) & 7], c[n], 0, 0, 0);
      }
      a = nx;
    }
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n) for (int i = 0; i < 4; ++i) s += c[n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NACC, int LDS>
void run(const float* src, float* out, int wps) {
  const int wgs = 256 * wps, iters = 24000 / (12 * 4 * NACC) * 4;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((chain<NACC, LDS>), dim3(wgs), dim3(256), 0, 0, src, out, iters);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((chain<NACC, LDS>), dim3(wgs), dim3(256), 0, 0, src, out, iters);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double n_mfma = (double)wgs * 4 * iters * 12 * 4 * NACC;
  printf("16x16x4 f32, %d chain(s) per wave, %d wave(s) per SIMD, %s: %7.1f TFLOP/s (%.2f of 157.3)\n", NACC, wps,
         LDS ? "A operand from ds_read_b128 (1 read per 4*chains MFMAs)" : "operands in registers", n_mfma * 2048.0 / (ms * 1e-3 / 5) * 1e-12,
         n_mfma * 2048.0 / (ms * 1e-3 / 5) * 1e-12 / 157.3);
}

int main() {
  float *src, *out;
  CK(hipMalloc(&src, 4096 * 4)); CK(hipMalloc(&out, (size_t)2048 * 256 * 4));
  float h[4096]; unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  CK(hipMemcpy(src, h, sizeof(h), hipMemcpyHostToDevice));
  for (int wps = 1; wps <= 4; ++wps) {
    run<1, 0>(src, out, wps); run<2, 0>(src, out, wps); run<4, 0>(src, out, wps); run<8, 0>(src, out, wps);
    run<2, 1>(src, out, wps); run<4, 1>(src, out, wps);
  }
  return 0;
}
