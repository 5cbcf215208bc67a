// conv_stem.hip -- the ResNet stem convolution (7x7, stride 2, pad 3, 3 -> 64 channels; torchvision resnet.conv1 as driven by
// reference src/model.py:127), forward, as a hand-written implicit GEMM.  It is the one convolution of the front-end where MIOpen
// is far from the matrix peak (K = 7*7*3 = 147 is awkward for its tiles: 452 us = 67 TF at 128 images, profiles/r2_conv_probe.txt;
// the other layers run at 100-135 TF there and stay on MIOpen).
//
// Row-resident formulation, as csrc/linear_rows.hip: a wave owns 16 consecutive output pixels of one output row; their 7x7x3 input
// patches are the B operand of v_mfma_f32_16x16x4_f32, gathered straight from the channels-last image into 44 VGPRs (lane (j, q)
// holds patch[j][16t + 4q + 0..3]); the whole filter bank sits in LDS once per workgroup ([64][7 x 24] with each filter row padded
// from 21 to 24 taps so that four consecutive k never straddle two image rows; zero weights on the pads; 48 KB, XOR-swizzled
// 16-byte chunks), so the main loop has no barrier and no global weight traffic: 44 ds_read_b128 + 176 MFMAs per 16 pixels.
// Eight waves share one copy of the filters and run 4 per SIMD, which covers the gather latency.  Output: lane (j, q) holds y[pixel j][16b + 4q + 0..3]:
// 16-byte stores, 256 contiguous bytes per pixel.
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
RP_DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

constexpr int CO = 64, KH = 7, KW = 7, CI = 3, KROW = 24, KP = 176, NT_ = 11;     // 7 rows x 24 = 168, padded to 11 groups of 16
constexpr int WROW = 192;                                                          // LDS row pitch (floats): 48 chunks of 16 bytes

struct StemP {
  const float *x, *w;     // x [N,H,W,3] = the image inside a 3-pixel zero frame; w [64][7][7][3] (a channels-last nn.Conv2d weight)
  float* y;               // [N,OH,OW,64]
  double* stats;          // optional [gridDim.x][2][64]: per-workgroup sums of y and y*y over its pixels (BatchNorm statistics partials)
  int N, H, W, OH, OW;
  int tiles_x, tiles;     // 16-pixel tiles per output row; total
};

struct f4u { float v[4]; } __attribute__((packed, aligned(4)));

constexpr int SNW = 8, SNT = SNW * 64;        // 8 waves share one copy of the filter bank; <= 128 VGPRs: 4 waves per SIMD hide the gathers

__global__ __launch_bounds__(SNT, 4) void conv_stem_fwd_kernel(StemP p) {
  __shared__ __attribute__((aligned(16))) float wl[CO * WROW];          // 48 KB
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, q = lane >> 4;
  // filter bank -> LDS: row = output channel, k = 24 ky + r; chunk index XOR (row & 15)
  for (int i = tid; i < CO * (WROW / 4); i += SNT) {
    const int row = i / (WROW / 4), ch = i % (WROW / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    float* o = &v.x;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int k = 4 * ch + e, ky = k / KROW, r = k - ky * KROW;
      if (ky < KH && r < KW * CI) o[e] = p.w[row * (KH * KW * CI) + ky * (KW * CI) + r];
    }
    st4(wl + row * WROW + ((ch ^ (row & 15)) * 4), v);
  }
  __syncthreads();
  const int nwaves = gridDim.x * SNW;
  int tile = blockIdx.x * SNW + wave;
  auto geom = [&](int t, int& n, int& oy, int& ox0) {
    const int tx = t % p.tiles_x;
    const int r = t / p.tiles_x;
    oy = r % p.OH;
    n = r / p.OH;
    ox0 = tx * 16;
  };
  // x is the image with a 3-pixel zero frame (H, W are the PADDED extents): every tap of every window is in bounds, and the three
  // pad taps read past each 21-tap filter row (zero weights) stay inside the row.  Window of output (oy, ox) starts at (2 oy, 2 ox).
  auto load = [&](int t, float (&pr)[4 * NT_]) {
    int n, oy, ox0;
    geom(t, n, oy, ox0);
    const int ox = min(ox0 + j, p.OW - 1);
    const float* win = p.x + (((long long)n * p.H + 2 * oy) * p.W + 2 * ox) * CI;
#pragma unroll
    for (int t4 = 0; t4 < NT_; ++t4) {
      const int k0 = 16 * t4 + 4 * q, ky = k0 / KROW, r0 = k0 - ky * KROW;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ky < KH) {
        const f4u u = *reinterpret_cast<const f4u*>(win + (long long)ky * p.W * CI + r0);
        v = make_float4(u.v[0], u.v[1], u.v[2], u.v[3]);
      }
      pr[4 * t4] = v.x; pr[4 * t4 + 1] = v.y; pr[4 * t4 + 2] = v.z; pr[4 * t4 + 3] = v.w;
    }
  };
  float s1[16], s2[16];                                   // per-lane sums over this wave's tiles: channel 16 hb + 4 q + r
#pragma unroll
  for (int i = 0; i < 16; ++i) { s1[i] = 0.f; s2[i] = 0.f; }
  for (; tile < p.tiles; tile += nwaves) {
    float cur[4 * NT_];
    load(tile, cur);                                      // (the other waves of the SIMD cover this gather's latency)
    f32x4v acc[4];
#pragma unroll
    for (int hb = 0; hb < 4; ++hb) acc[hb] = f32x4v{0.f, 0.f, 0.f, 0.f};
    auto wfrag = [&](int it) {                          // it = 4 * t4 + hb
      const int t4 = it >> 2, hb = it & 3;
      return ld4(wl + (16 * hb + j) * WROW + (((4 * t4 + q) ^ j) * 4));
    };
    float4 a = wfrag(0);
    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
#pragma unroll
    for (int it = 0; it < 4 * NT_; ++it) {
      const int t4 = it >> 2, hb = it & 3;
      float4 nx = a;
      if (it + 1 < 4 * NT_) nx = wfrag(it + 1);        // one fragment in flight under the four MFMAs of the current one
      acc[hb] = mfma16(a.x, cur[4 * t4], acc[hb]);
      acc[hb] = mfma16(a.y, cur[4 * t4 + 1], acc[hb]);
      acc[hb] = mfma16(a.z, cur[4 * t4 + 2], acc[hb]);
      acc[hb] = mfma16(a.w, cur[4 * t4 + 3], acc[hb]);
      a = nx;
      __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    }
    int n, oy, ox0;
    geom(tile, n, oy, ox0);
    if (ox0 + j < p.OW) {
      float* yr = p.y + (((long long)n * p.OH + oy) * p.OW + ox0 + j) * CO + 4 * q;
#pragma unroll
      for (int hb = 0; hb < 4; ++hb) st4(yr + 16 * hb, make_float4(acc[hb][0], acc[hb][1], acc[hb][2], acc[hb][3]));
      if (p.stats) {
#pragma unroll
        for (int hb = 0; hb < 4; ++hb)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            s1[4 * hb + r] += acc[hb][r];
            s2[4 * hb + r] = fmaf(acc[hb][r], acc[hb][r], s2[4 * hb + r]);
          }
      }
    }
  }
  if (p.stats) {
    // fixed-order combine: 16 pixels of a wave (DPP row sum), then the 8 waves in double (the filter bank in LDS is dead by now)
    __syncthreads();
    float* red = wl;                                      // [SNW][2][64]
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

}  // namespace

// x_padded [N, H+6, W+6, 3]: the channels-last image inside a 3-pixel zero frame (rp_preprocess can write it directly); w [64,7,7,3];
// y [N, OH, OW, 64] with OH = (H - 1) / 2 + 1; stats (optional): [rp_conv_stem_blocks(N,H,W)][2][64] doubles = per-workgroup sums of y and y^2
// per channel (the BatchNorm batch statistics come out of the convolution's epilogue: rp_bn_stats_from_partials finishes them)
static int stem_slots() {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, conv_stem_fwd_kernel, SNT, 0);
    slots = cus * (per_cu > 0 ? per_cu : 1);
  }
  return slots;
}

// number of workgroups rp_conv_stem_fwd launches = rows of its `stats` output
extern "C" int rp_conv_stem_blocks(int N, int H, int W) {
  if (N <= 0 || H < 1 || W < 1) return 0;
  const long long tiles = (long long)N * ((H - 1) / 2 + 1) * (((W - 1) / 2 + 1 + 15) / 16);
  const long long wgs = (tiles + SNW - 1) / SNW;
  return (int)(wgs < stem_slots() ? wgs : stem_slots());
}

extern "C" int rp_conv_stem_fwd(const float* x_padded, const float* w, float* y, double* stats, int N, int H, int W, void* stream) {
  if (N <= 0 || H < 1 || W < 1 || !x_padded || !w || !y) return RP_EBADSHAPE;
  StemP p{x_padded, w, y, stats, N, H + 6, W + 6, (H + 2 * 3 - KH) / 2 + 1, (W + 2 * 3 - KW) / 2 + 1, 0, 0};
  if (2 * (p.OW - 1) + KROW / CI > p.W) return RP_EBADSHAPE;              // the last window's pad taps must stay inside its row
  p.tiles_x = (p.OW + 15) / 16;
  const long long tiles = (long long)N * p.OH * p.tiles_x;
  if (tiles >= (1LL << 31)) return RP_EBADSHAPE;
  p.tiles = (int)tiles;
  const int slots = stem_slots();
  const int grid = (int)((tiles + SNW - 1) / SNW < slots ? (tiles + SNW - 1) / SNW : slots);
  hipLaunchKernelGGL(conv_stem_fwd_kernel, dim3(grid), dim3(SNT), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
