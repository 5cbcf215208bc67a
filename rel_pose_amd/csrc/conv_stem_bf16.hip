// conv_stem_bf16.hip -- the ResNet stem convolution (7x7, stride 2, pad 3, 3 -> 64; torchvision resnet.conv1 as driven by reference
// src/model.py:127) of the bf16 configuration (BASELINE.json configs[4]): bf16 operands on v_mfma_f32_16x16x32_bf16, fp32 accumulate, bf16
// output.  Same row-resident formulation as the exact-fp32 kernel (conv_stem.hip): a wave owns 16 consecutive output pixels of one
// output row; their 7 x 7 x 3 patches are the B operand, gathered straight from the fp32 channels-last image inside its 3-pixel zero
// frame (rp_preprocess_padded's output: no bf16 copy of the image is ever made) and rounded to bf16 in registers; the filter bank is
// rounded to bf16 once per workgroup into LDS ([64][7 x 24 + 24] taps, each filter row padded from 21 to 24 so that eight consecutive k
// never straddle two image rows; 512-byte row pitch, 16-byte chunk index XOR (row & 15): conflict-free ds_read_b128).  K = 192 -> 6 MFMAs
// per 16 pixels x 16 channels instead of the fp32 kernel's 44: the launch is bound by its 411 MB of output (at 256 images), which is
// why MIOpen's implicit GEMM (348 us, K = 147 fits none of its tiles) loses to it.  In training the epilogue also emits the BatchNorm
// batch statistics of the stored (bf16-rounded) values, like conv_stem.hip.
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef unsigned short bf16_t;
constexpr int CO = 64, KH = 7, KW = 7, CI = 3, KROW = 24, KT = 6;     // 7 filter rows x 24 = 168 taps, padded to 6 k-steps of 32
constexpr int WPITCH = 256;                                            // LDS row pitch in bf16 elements (512 B = 32 chunks)
constexpr int SNW = 8, SNT = SNW * 64;

struct StemBP {
  const float *x, *w;     // x [N,H,W,3] fp32 = the image inside a 3-pixel zero frame; w [64][7][7][3] fp32 (channels-last nn.Conv2d weight)
  bf16_t* y;              // [N,OH,OW,64] bf16
  double* stats;          // optional [gridDim.x][2][64]
  int N, H, W, OH, OW, tiles_x, tiles;
};
struct f4u { float v[4]; } __attribute__((packed, aligned(4)));

template <bool STATS>
__global__ __launch_bounds__(SNT, 4) void conv_stem_bf16_kernel(StemBP p) {
  __shared__ __attribute__((aligned(16))) bf16_t wl[CO * WPITCH];        // 32 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, q = lane >> 4;
  // filter bank -> LDS (bf16): row = output channel, k = 24 ky + r, 8 taps per 16-byte chunk, chunk index XOR (row & 15)
  for (int i = tid; i < CO * (KT * 4); i += SNT) {
    const int row = i / (KT * 4), ch = i % (KT * 4);
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int k = 8 * ch + e, ky = k / KROW, r = k - ky * KROW;
      v[e] = (ky < KH && r < KW * CI) ? p.w[row * (KH * KW * CI) + ky * (KW * CI) + r] : 0.f;
    }
    *reinterpret_cast<bf16x8*>(wl + row * WPITCH + ((ch ^ (row & 15)) << 3)) = pack8(v);
  }
  __syncthreads();
  // this lane's gather offsets: k-step t, k = 32 t + 8 q .. + 7 = filter row ky, taps r0 .. r0 + 7 (never across a filter row: 24 = 3 x 8)
  int goff[KT];
#pragma unroll
  for (int t = 0; t < KT; ++t) {
    const int k0 = 32 * t + 8 * q, ky = k0 / KROW, r0 = k0 - ky * KROW;
    goff[t] = ky < KH ? ky * p.W * CI + r0 : -1;
  }
  const int nwaves = gridDim.x * SNW;
  float s1[16], s2[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  for (int tile = blockIdx.x * SNW + wave; tile < p.tiles; tile += nwaves) {
    const int tx = tile % p.tiles_x, rr = tile / p.tiles_x, oy = rr % p.OH, n = rr / p.OH, ox0 = tx * 16;
    const int ox = min(ox0 + j, p.OW - 1);
    const float* win = p.x + (((long long)n * p.H + 2 * oy) * p.W + 2 * ox) * CI;
    bf16x8 pb[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t) {
      float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
      if (goff[t] >= 0) {
        const f4u u0 = *reinterpret_cast<const f4u*>(win + goff[t]), u1 = *reinterpret_cast<const f4u*>(win + goff[t] + 4);
#pragma unroll
        for (int e = 0; e < 4; ++e) { v[e] = u0.v[e]; v[4 + e] = u1.v[e]; }
      }
      pb[t] = pack8(v);
    }
    f32x4v acc[4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) acc[hb] = f32x4v{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) {
        const bf16x8 a = *reinterpret_cast<const bf16x8*>(wl + (16 * hb + j) * WPITCH + (((4 * t + q) ^ j) << 3));
        acc[hb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, pb[t], acc[hb], 0, 0, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);          // one filter fragment in flight per MFMA (else all 24 are hoisted: 96 registers)
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      }
    // lane (j, q): pixel j, channels 16 hb + 4 q .. + 3.  Pairs of 16-channel blocks go out as ONE 16-byte store per lane after a
    // half swap with lane q ^ 1 (common.h: st_bf16x8): even q writes channels 32 h2 + 4 q .. + 7, odd q 32 h2 + 16 + 4 (q - 1) .. + 7
    // -- 64 contiguous bytes per pixel and instruction instead of 32 (the 8-byte form was store-issue-bound: 261 us per launch)
    {
      const bool live = ox0 + j < p.OW;
      bf16_t* yr = p.y + (((long long)n * p.OH + oy) * p.OW + min(ox0 + j, p.OW - 1)) * CO + ((q & 1) ? 16 + 4 * (q - 1) : 4 * q);
#pragma unroll
      for (int h2 = 0; h2 < 2; ++h2) {
        const float4 v0 = make_float4(acc[2 * h2][0], acc[2 * h2][1], acc[2 * h2][2], acc[2 * h2][3]);
        const float4 v1 = make_float4(acc[2 * h2 + 1][0], acc[2 * h2 + 1][1], acc[2 * h2 + 1][2], acc[2 * h2 + 1][3]);
        if (live) st_bf16x8(yr + 32 * h2, v0, v1, q);          // (lanes q and q ^ 1 share the pixel: both live or both not)
      }
      if (STATS && live) {
#pragma unroll
        for (int hb = 0; hb < 4; ++hb) {
          const unsigned w0 = pk_bf16(acc[hb][0], acc[hb][1]), w1 = pk_bf16(acc[hb][2], acc[hb][3]);
          const float v0 = __uint_as_float(w0 << 16), v1 = __uint_as_float(w0 & 0xffff0000u), v2 = __uint_as_float(w1 << 16),
                      v3 = __uint_as_float(w1 & 0xffff0000u);
          s1[4 * hb] += v0; s1[4 * hb + 1] += v1; s1[4 * hb + 2] += v2; s1[4 * hb + 3] += v3;
          s2[4 * hb] = fmaf(v0, v0, s2[4 * hb]); s2[4 * hb + 1] = fmaf(v1, v1, s2[4 * hb + 1]);
          s2[4 * hb + 2] = fmaf(v2, v2, s2[4 * hb + 2]); s2[4 * hb + 3] = fmaf(v3, v3, s2[4 * hb + 3]);
        }
      }
    }
  }
  if (STATS) {     // fixed-order combine: 16 pixels of a wave (DPP row sum), then the 8 waves in double (the filter bank is dead by now)
    __syncthreads();
    float* red = reinterpret_cast<float*>(wl);            // [SNW][2][64]
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float a = row16_sum(s1[i]), b = row16_sum(s2[i]);
      if (j == 0) {
        const int c = 16 * (i >> 2) + 4 * q + (i & 3);
        red[(wave * 2 + 0) * CO + c] = a;
        red[(wave * 2 + 1) * CO + c] = b;
      }
    }
    __syncthreads();
    if (tid < 2 * CO) {
      double acc = 0.0;
      for (int w = 0; w < SNW; ++w) acc += (double)red[(w * 2 + tid / CO) * CO + tid % CO];
      p.stats[(long long)blockIdx.x * 2 * CO + tid] = acc;
    }
  }
}

int stem_bf16_slots() {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_stem_bf16_kernel<true>, SNT, 0);
    slots = cus * (per_cu > 0 ? per_cu : 1);
  }
  return slots;
}

}  // namespace

extern "C" int rp_conv_stem_bf16_blocks(int N, int H, int W) {
  if (N <= 0 || H < 1 || W < 1) return 0;
  const long long tiles = (long long)N * ((H - 1) / 2 + 1) * (((W - 1) / 2 + 1 + 15) / 16);
  const long long wgs = (tiles + SNW - 1) / SNW;
  return (int)(wgs < stem_bf16_slots() ? wgs : stem_bf16_slots());
}

/* x_padded [N, H+6, W+6, 3] fp32 (3-pixel zero frame), w [64,7,7,3] fp32 -> y [N, OH, OW, 64] bf16; stats (optional) as rp_conv_stem_fwd */
extern "C" int rp_conv_stem_fwd_bf16(const float* x_padded, const float* w, void* y, double* stats, int N, int H, int W, void* stream) {
  if (N <= 0 || H < 1 || W < 1 || !x_padded || !w || !y) return RP_EBADSHAPE;
  StemBP p{x_padded, w, (bf16_t*)y, stats, N, H + 6, W + 6, (H + 2 * 3 - KH) / 2 + 1, (W + 2 * 3 - KW) / 2 + 1, 0, 0};
  if (2 * (p.OW - 1) + KROW / CI > p.W) return RP_EBADSHAPE;              // the last window's pad taps must stay inside its row
  p.tiles_x = (p.OW + 15) / 16;
  const long long tiles = (long long)N * p.OH * p.tiles_x;
  if (tiles >= (1LL << 31)) return RP_EBADSHAPE;
  p.tiles = (int)tiles;
  const int slots = stem_bf16_slots();
  const int grid = (int)((tiles + SNW - 1) / SNW < slots ? (tiles + SNW - 1) / SNW : slots);
  if (stats) hipLaunchKernelGGL(conv_stem_bf16_kernel<true>, dim3(grid), dim3(SNT), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(conv_stem_bf16_kernel<false>, dim3(grid), dim3(SNT), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
