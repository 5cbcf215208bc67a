// micro-benchmarks for gfx950 (one wave per SIMD unless stated):
//   (1) dependent-issue latency of v_mfma_f32_32x32x16_bf16 / v_mfma_f32_32x32x2_f32: NACC independent accumulators, round-robin
//   (2) VALU instructions interleaved between the MFMAs of the same wave
//   (3) two waves on one SIMD (512-thread workgroup): MFMA-only wave next to a VALU-only wave
// build: hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o mfma_valu mfma_valu.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int NACC, int NV, bool F32, int SPLIT>   // SPLIT: 0 = all waves same; 1 = waves 0-3 MFMA only, 4-7 VALU only
__global__ __launch_bounds__(512) void k(float* out, long long* cyc, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool do_m = SPLIT == 0 || wave < 4, do_v = SPLIT == 0 || wave >= 4;
  f32x16 c[NACC];
  for (int n = 0; n < NACC; ++n) c[n] = f32x16{};
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(threadIdx.x * 0.001f + i); b[i] = (__bf16)(i * 0.5f); }
  float fa = threadIdx.x * 0.001f, fb = 0.5f;
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
  if (do_m && do_v) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if (F32) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c[n], 0, 0, 0);
        else c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[n], 0, 0, 0);
#pragma unroll
        for (int r = 0; r < NV; ++r) v[r & 7] = __builtin_fmaf(v[r & 7], 1.0001f, 0.5f);
      }
    }
  } else if (do_m) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < NACC; ++n) {
        if (F32) c[n] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, c[n], 0, 0, 0);
        else c[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[n], 0, 0, 0);
      }
    }
  } else {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int n = 0; n < NACC; ++n)
#pragma unroll
        for (int r = 0; r < NV; ++r) v[r & 7] = __builtin_fmaf(v[r & 7], 1.0001f, 0.5f);
    }
  }
  long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int n = 0; n < NACC; ++n)
    for (int i = 0; i < 16; ++i) s += c[n][i];
  for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int NACC, int NV, bool F32, int SPLIT>
void run(const char* name, int threads) {
  float* out; long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8 * 8);
  const int iters = 20000;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, NV, F32, SPLIT>), dim3(256), dim3(threads), 0, 0, out, cyc, 100);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, NV, F32, SPLIT>), dim3(256), dim3(threads), 0, 0, out, cyc, iters);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  long long h[8]; hipMemcpy(h, cyc, 64, hipMemcpyDeviceToHost);
  printf("%-44s NACC=%d NV=%2d | wall/iter %.1f ns | cyc per MFMA-slot: wave0 %.1f  wave4 %.1f\n", name, NACC, NV, ms * 1e6 / iters,
         (double)h[0] / iters / NACC, threads > 256 ? (double)h[4] / iters / NACC : 0.0);
  hipFree(out); hipFree(cyc);
}

int main() {
  printf("-- bf16 32x32x16 (32 cycles of matrix pipe), NACC independent accumulators, no VALU\n");
  run<1, 0, false, 0>("bf16 MFMA chain", 256);
  run<2, 0, false, 0>("bf16 MFMA chain", 256);
  run<3, 0, false, 0>("bf16 MFMA chain", 256);
  run<4, 0, false, 0>("bf16 MFMA chain", 256);
  run<6, 0, false, 0>("bf16 MFMA chain", 256);
  printf("-- fp32 32x32x2 (64 cycles)\n");
  run<1, 0, true, 0>("fp32 MFMA chain", 256);
  run<2, 0, true, 0>("fp32 MFMA chain", 256);
  run<4, 0, true, 0>("fp32 MFMA chain", 256);
  printf("-- VALU between MFMAs of the same wave (4 accumulators)\n");
  run<4, 4, false, 0>("bf16 MFMA + NV fma each", 256);
  run<4, 8, false, 0>("bf16 MFMA + NV fma each", 256);
  run<4, 12, false, 0>("bf16 MFMA + NV fma each", 256);
  run<4, 16, false, 0>("bf16 MFMA + NV fma each", 256);
  run<4, 24, false, 0>("bf16 MFMA + NV fma each", 256);
  printf("-- two waves per SIMD\n");
  run<4, 0, false, 0>("both waves: bf16 MFMA only", 512);
  run<4, 8, false, 0>("both waves: MFMA + 8 fma interleaved", 512);
  run<4, 16, false, 1>("wave A MFMA only | wave B 16 fma per slot", 512);
  run<4, 8, false, 1>("wave A MFMA only | wave B 8 fma per slot", 512);
  run<2, 16, false, 1>("wave A MFMA only (2 acc) | wave B 16 fma", 512);
  return 0;
}
