// conv_stem_wgrad_f32.hip -- weight gradient of the ResNet stem convolution (7x7 / 2, pad 3, 3 -> 64; resnet.conv1, reference
// src/model.py:127) in EXACT fp32 (the headline configuration):
//
//     dW[co][ky][kx][ci] = sum over (image, oy, ox) of dY[n, oy, ox, co] * Xp[n, 2 oy + ky, 2 ox + kx, ci]      (Xp = image in its zero frame)
//
// MIOpen's backward-weights runs it at 0.43 of the fp32 MFMA peak (450 us at 128 images: K = 147 fits none of its tiles).  Two steps:
//   1. SPACE TO DEPTH: the framed fp32 image [N,230,230,3] becomes P [N,115,115,12] with channel (ky & 1, kx & 1, ci), which turns the
//      stride-2 7x7 convolution into a stride-1 4x4 convolution over 12 channels (taps (u, v) = (ky >> 1, kx >> 1); the 8th row / column
//      of the 8x8 support has zero weight and is dropped at the end): 192 columns = six N-tiles of 32 instead of 147;
//   2. conv3x3_wgrad_f32.hip's output-stationary stream on that: the [64 co] x [192] result lives in 12 accumulator tiles of 32 x 32
//      (wave (coh, nq): three N-tiles), the workgroup walks output ROWS (112 pixels = 56 k-steps of 2); the four P rows of a tile are one
//      contiguous 22 KB run and the dY row 28 KB, both copied as they lie by LDS-DMA into two buffers, one piece per four k-steps behind
//      the MFMAs; an MFMA operand is one conflict-free ds_read_b32 with an immediate offset (lane = output channel / column, half-wave =
//      which pixel of the k-step; a column's (tap, channel) offset sits in the lane's base address); per-workgroup partials, fixed-order
//      reduce straight into the fp32 [64][7][7][3] gradient.
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

constexpr int CO = 64, OW = 112, OH = 112, PW = 115, PH = 115, PC = 12;
constexpr int PROWB = PW * PC * 4;                 // 5520 B per P row
constexpr int NXP = (4 * PROWB + 1023) / 1024;     // 22 DMA pieces for the four rows of a tile (the last one overruns by 448 B)
constexpr int XB = NXP * 1024;                     // 22 528 B
constexpr int YROWB = OW * CO * 4;                 // 28 672 B of dY per output row
constexpr int NYP = YROWB / 1024;                  // 28
constexpr int KS = OW / 2;                         // 56 k-steps of 2 pixels
constexpr int NCOL = 16 * PC;                      // 192 columns: tap x channel
static_assert(NYP % 4 == 0 && 6 + NYP / 4 <= KS / 4 && (NXP + 3) / 4 <= 6, "DMA pieces of a tile must fit the k-step slots");

struct SwF {
  const float* p;       // [N,115,115,12] (+ 1 KB of slack behind it)
  const float* dy;      // [N,112,112,64]
  float* ws;            // [gridDim.x][64][192]
  int ntiles;           // N * 112 output rows
};

RP_DEV void glds16s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
RP_DEV const void* uniform_ptr_s(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
template <int OFF> RP_DEV float rd32s(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N, class F> RP_DEV void sfor_s(F&& f) {
  if constexpr (N > 0) {
    sfor_s<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__global__ __launch_bounds__(256, 1) void conv_stem_wgrad_f32_kernel(SwF p) {
  __shared__ __attribute__((aligned(16))) unsigned char Xs[2][XB];        // 45 056 B
  __shared__ __attribute__((aligned(16))) unsigned char Ys[2][YROWB];     // 57 344 B
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int coh = wave >> 1, nq = wave & 1;
  const int G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((long long)p.ntiles * b / G), t1 = (int)((long long)p.ntiles * (b + 1) / G);
  const unsigned xs0 = (unsigned)(size_t)(rp_lds_ptr_t)(&Xs[0][0]), ys0 = (unsigned)(size_t)(rp_lds_ptr_t)(&Ys[0][0]);

  f32x16 acc[3] = {zero16(), zero16(), zero16()};
  // column c = 32 (3 nq + i) + l31 = 12 tap + ch, tap = 4 u + v: P position (oy + u, ox + v), channel ch
  unsigned boff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int c = 32 * (3 * nq + i) + l31, tap = c / PC, ch = c - tap * PC;
    boff[i] = (unsigned)((tap >> 2) * PROWB + (tap & 3) * (PC * 4) + ch * 4 + hi * (PC * 4));
  }
  const unsigned aoff = (unsigned)((32 * coh + l31) * 4 + hi * (CO * 4));

  auto dma_x = [&](int t, int buf, int q) {   // piece q of the four P rows of tile t = (image t / 112, output row t % 112): one contiguous run
    const int img = t / OH, oy = t - img * OH;
    glds16s(uniform_ptr_s(reinterpret_cast<const unsigned char*>(p.p) + ((long long)img * PH + oy) * PROWB), (unsigned)(q * 1024 + lane * 16),
            xs0 + buf * XB + q * 1024);
  };
  auto dma_y = [&](int t, int buf, int q) {
    glds16s(uniform_ptr_s(reinterpret_cast<const unsigned char*>(p.dy) + (long long)t * YROWB), (unsigned)(q * 1024 + lane * 16),
            ys0 + buf * YROWB + q * 1024);
  };
  if (t0 < t1) {
    for (int q = wave; q < NXP; q += 4) dma_x(t0, t0 & 1, q);
    for (int q = wave; q < NYP; q += 4) dma_y(t0, t0 & 1, q);
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int t = t0; t < t1; ++t) {
    // the next tile goes into the other buffers; the run's last tile fetches ITSELF there again (nobody reads it: no branch in the loop)
    const int tn = min(t + 1, t1 - 1), bn = (t & 1) ^ 1;
    const unsigned ya = ys0 + (t & 1) * YROWB + aoff;
    const unsigned x0 = xs0 + (t & 1) * XB;
    const unsigned xa0 = x0 + boff[0], xa1 = x0 + boff[1], xa2 = x0 + boff[2];
    float a, b0, b1, b2;
    a = rd32s<0>(ya);
    b0 = rd32s<0>(xa0); b1 = rd32s<0>(xa1); b2 = rd32s<0>(xa2);
    sfor_s<KS>([&](auto kc) {
      constexpr int k = kc;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b0), "+v"(b1), "+v"(b2));
      float an = 0.f, n0 = 0.f, n1 = 0.f, n2 = 0.f;
      if constexpr (k + 1 < KS) {
        an = rd32s<2 * (k + 1) * CO * 4>(ya);
        n0 = rd32s<2 * (k + 1) * PC * 4>(xa0);
        n1 = rd32s<2 * (k + 1) * PC * 4>(xa1);
        n2 = rd32s<2 * (k + 1) * PC * 4>(xa2);
      }
      if constexpr (k % 4 == 0) {                      // the next tile: one DMA piece per four k-steps, behind the MFMAs
        constexpr int j = k / 4;                       // 0 .. 13
        if constexpr (j < 6) {
          if (wave + 4 * j < NXP) dma_x(tn, bn, wave + 4 * j);
        } else if constexpr (j < 6 + NYP / 4) {
          dma_y(tn, bn, wave + 4 * (j - 6));           // j = 6 .. 12: pieces wave, wave + 4, .. wave + 24 (<= 27)
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      acc[0] = mfma32(a, b0, acc[0]);
      acc[1] = mfma32(a, b1, acc[1]);
      acc[2] = mfma32(a, b2, acc[2]);
      if constexpr (k + 1 < KS) { a = an; b0 = n0; b1 = n1; b2 = n2; }
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // partial [64 co][192]: rows co = 32 coh + acc_row(r, hi), columns 32 (3 nq + i) + l31
  float* o = p.ws + (long long)b * (CO * NCOL) + (32 * coh) * NCOL + 96 * nq + l31;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[acc_row(r, hi) * NCOL + 32 * i] = acc[i][r];
}

// dW[co][ky][kx][ci] fp32 <- sum over workgroups of ws[b][co][12 (4 u + v) + 3 (2 (ky & 1) + (kx & 1)) + ci], u = ky >> 1, v = kx >> 1.
// 64 outputs x 4 slices of the partials per workgroup; fixed order: a slice front to back, then the four slices.
__global__ __launch_bounds__(256) void conv_stem_wgrad_f32_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nblk) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int idx = min(blockIdx.x * 64 + o, CO * 147 - 1);
  const int co = idx / 147, r = idx - co * 147, ky = r / 21, r2 = r - ky * 21, kx = r2 / 3, ci = r2 - kx * 3;
  const int col = PC * (4 * (ky >> 1) + (kx >> 1)) + 3 * (2 * (ky & 1) + (kx & 1)) + ci;
  const float* src = ws + co * NCOL + col;
  const int k0 = (int)((long long)nblk * sl / 4), k1 = (int)((long long)nblk * (sl + 1) / 4);
  float s = 0.f;
#pragma unroll 8
  for (int k = k0; k < k1; ++k) s += src[(long long)k * (CO * NCOL)];
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0 && blockIdx.x * 64 + o < CO * 147) dw[idx] = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
}

// P[n][a][b][(pa, pb, ci)] = Xp[n][2a + pa][2b + pb][ci]
__global__ __launch_bounds__(256) void stem_s2d_f32_kernel(const float* __restrict__ xp, float* __restrict__ out, int N, int Hp, int Wp) {
  const long long total = (long long)N * (Hp / 2) * (Wp / 2);
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int bq = (int)(i % (Wp / 2));
  const long long r = i / (Wp / 2);
  const int a = (int)(r % (Hp / 2)), n = (int)(r / (Hp / 2));
  const float* r0 = xp + (((long long)n * Hp + 2 * a) * Wp + 2 * bq) * 3;
  const float* r1 = r0 + (long long)Wp * 3;
  float v[12];
#pragma unroll
  for (int e = 0; e < 6; ++e) { v[e] = r0[e]; v[6 + e] = r1[e]; }
  float4* o = reinterpret_cast<float4*>(out + i * PC);
  o[0] = make_float4(v[0], v[1], v[2], v[3]);
  o[1] = make_float4(v[4], v[5], v[6], v[7]);
  o[2] = make_float4(v[8], v[9], v[10], v[11]);
}

size_t p_bytes(int N) { return (((size_t)N * PH * PW * PC * 4 + 1024) + 255) & ~(size_t)255; }      // + slack for the last run's overrun
int blocks_of(int N) {
  const long long tiles = (long long)N * OH;
  return (int)(tiles < 256 ? tiles : 256);
}

}  // namespace

extern "C" size_t rp_conv_stem_wgrad_f32_workspace_bytes(int N) {
  if (N <= 0) return 0;
  return p_bytes(N) + (size_t)blocks_of(N) * CO * NCOL * sizeof(float);
}

/* dw [64][7][7][3] fp32 from the framed fp32 image x_padded [N,230,230,3] and dY [N,112,112,64] fp32 (224 x 224 images only) */
extern "C" int rp_conv_stem_wgrad_f32(const float* x_padded, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H,
                                      int W, void* stream) {
  if (!x_padded || !dy || !dw || !workspace || N <= 0) return RP_EBADSHAPE;
  if (H != 224 || W != 224) return RP_EUNSUPPORTED;
  if (((uintptr_t)x_padded | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) return RP_EALIGN;
  if (workspace_bytes < rp_conv_stem_wgrad_f32_workspace_bytes(N)) return RP_EWORKSPACE;
  hipStream_t st = (hipStream_t)stream;
  float* P = (float*)workspace;
  float* ws = (float*)((char*)workspace + p_bytes(N));
  const long long npos = (long long)N * PH * PW;
  hipLaunchKernelGGL(stem_s2d_f32_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, st, x_padded, P, N, 2 * PH, 2 * PW);
  RP_CHECK_LAUNCH();
  const int nblk = blocks_of(N);
  SwF p{P, dy, ws, N * OH};
  hipLaunchKernelGGL(conv_stem_wgrad_f32_kernel, dim3(nblk), dim3(256), 0, st, p);
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv_stem_wgrad_f32_reduce_kernel, dim3((CO * 147 + 63) / 64), dim3(256), 0, st, (const float*)ws, dw, nblk);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
