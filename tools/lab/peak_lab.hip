// peak_lab.hip -- sustained matrix-pipe ceiling of one MI355X under its power limit (standalone, no torch).
// Every SIMD of the chip issues back-to-back MFMAs on NACC independent accumulators; operands are random (or zero) data.
// Prints TFLOP/s and the shader clock the run sustained: wave 0 of every workgroup reads s_memtime (shader cycles) and the
// constant-rate 100 MHz wall clock around its loop.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o peak_lab peak_lab.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

// KIND 0: v_mfma_f32_32x32x2_f32 (4 acc x 16 regs)   1: v_mfma_f32_16x16x4_f32 (8 acc x 4 regs)   2: v_mfma_f32_32x32x16_bf16
template <int KIND>
__global__ __launch_bounds__(256) void peak(const float* src, float* out, long long* clk, int iters) {
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + 256 * i) & 4095]; b[i] = src[(threadIdx.x * 7 + 256 * i + 13) & 4095]; }
  float s = 0.f;
  const long long w0 = wall_clock64(), c0 = __builtin_readcyclecounter();
  if (KIND == 0) {
    f32x16 c[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], b[(t + n) & 7], c[n], 0, 0, 0);
    }
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
  } else if (KIND == 1) {
    f32x4 c[8];
    for (int n = 0; n < 8; ++n) for (int i = 0; i < 4; ++i) c[n][i] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int n = 0; n < 8; ++n) c[n] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[t], b[(t + n) & 7], c[n], 0, 0, 0);
    }
    for (int n = 0; n < 8; ++n) for (int i = 0; i < 4; ++i) s += c[n][i];
  } else {
    f32x16 c[4];
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
    bf16x8 x[2], y[2];
    for (int q = 0; q < 2; ++q) for (int i = 0; i < 8; ++i) { x[q][i] = (__bf16)a[(i + q) & 7]; y[q][i] = (__bf16)b[(i + 3 * q) & 7]; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int n = 0; n < 4; ++n) c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x[t & 1], y[(t + n) & 1], c[n], 0, 0, 0);
    }
    for (int n = 0; n < 4; ++n) for (int i = 0; i < 16; ++i) s += c[n][i];
  }
  const long long c1 = __builtin_readcyclecounter(), w1 = wall_clock64();
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main(int argc, char** argv) {
  const int kind = argc > 1 ? atoi(argv[1]) : 0;
  const int wgs = argc > 2 ? atoi(argv[2]) : 2048;
  const int iters = argc > 3 ? atoi(argv[3]) : 2000;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const int zero = argc > 5 ? atoi(argv[5]) : 0;
  std::vector<float> h(4096);
  unsigned s = 12345u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = zero ? 0.f : (((s >> 8) & 0xffff) / 32768.0f - 1.0f); }
  float *src, *out; long long* clk;
  CK(hipMalloc(&src, 4096 * 4)); CK(hipMalloc(&out, (size_t)wgs * 256 * 4)); CK(hipMalloc(&clk, (size_t)wgs * 16));
  CK(hipMemcpy(src, h.data(), 4096 * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto launch = [&]() {
    if (kind == 0) hipLaunchKernelGGL(peak<0>, dim3(wgs), dim3(256), 0, 0, src, out, clk, iters);
    else if (kind == 1) hipLaunchKernelGGL(peak<1>, dim3(wgs), dim3(256), 0, 0, src, out, clk, iters);
    else hipLaunchKernelGGL(peak<2>, dim3(wgs), dim3(256), 0, 0, src, out, clk, iters);
  };
  launch(); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch();
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  const double flop_per_mfma = kind == 0 ? 4096.0 : kind == 1 ? 2048.0 : 32768.0;
  const double n_mfma = (double)wgs * 4 * iters * (kind == 1 ? 64 : 32);
  std::vector<long long> hc(2 * wgs);
  CK(hipMemcpy(hc.data(), clk, (size_t)wgs * 16, hipMemcpyDeviceToHost));
  double cyc = 0, wall = 0;
  for (int i = 0; i < wgs; ++i) { cyc += hc[2 * i]; wall += hc[2 * i + 1]; }
  const char* names[] = {"v_mfma_f32_32x32x2_f32", "v_mfma_f32_16x16x4_f32", "v_mfma_f32_32x32x16_bf16"};
  printf("%-26s %5d WGs x 4 waves, %5d iters, %s data: %9.1f us/launch  %8.1f TFLOP/s   shader clock %.3f GHz (s_memtime / 100 MHz wall clock)\n",
         names[kind], wgs, iters, zero ? "zero  " : "random", ms * 1e3 / reps, n_mfma * flop_per_mfma / (ms * 1e-3 / reps) * 1e-12, cyc / wall * 0.1);
  return 0;
}
