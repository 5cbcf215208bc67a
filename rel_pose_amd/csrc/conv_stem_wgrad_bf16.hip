// conv_stem_wgrad_bf16.hip -- weight gradient of the ResNet stem convolution (7x7 / 2, pad 3, 3 -> 64; resnet.conv1, reference
// src/model.py:127) in the bf16 configuration (BASELINE.json configs[4]):
//
//     dW[co][ky][kx][ci] = sum over (image, oy, ox) of dY[n, oy, ox, co] * Xp[n, 2 oy + ky, 2 ox + kx, ci]      (Xp = image in its zero frame)
//
// MIOpen's backward-weights needs 350 us for it at 256 images (K = 147 fits none of its tiles) plus a bf16 copy of the image; the launch
// moves 411 MB of dY, so ~100 us is the floor.  Two steps:
//   1. SPACE TO DEPTH (rp_stem_s2d_bf16): the framed fp32 image [N,230,230,3] becomes P [N,115,115,16] bf16 with channel
//      (ky & 1, kx & 1, ci) = 12 real + 4 zero, which turns the stride-2 7x7 convolution into a stride-1 4x4 convolution over 16 channels
//      (taps (u, v) = (ky >> 1, kx >> 1); the 8th row / column of the 8x8 support has zero weight and is dropped at the end);
//   2. the output-stationary stream of conv3x3_wgrad_bf16.hip on that: the [64 co] x [16 taps x 16 ch] result lives in 16 accumulator
//      tiles of 32 x 32 (wave (coh, nq): 4 tiles), the workgroup walks (image, 2-row strip) tiles of 224 output pixels = 14 k-steps of
//      16 pixels, both MFMA operands come from LDS by transpose reads (dY from 16-byte channel planes; P rows are copied as they lie:
//      a position is 32 bytes, so the 4 x 16 block of a read is 128 contiguous bytes), next tile requested a tile ahead, one barrier per
//      tile, fixed-order reduce of the per-workgroup partials (second launch) straight into the fp32 [64][7][7][3] gradient.
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef unsigned short bf16_t;
constexpr int CO = 64, OW = 112, OH = 112, TH = 2, TPI = OH / TH;        // 56 strips of 2 output rows per image
constexpr int PW = 115, PH = 115, PCH = 16;                               // space-to-depth image
constexpr int XROWS = TH + 3, XPOS = XROWS * PW;                          // 5 rows x 115 positions of 32 B
constexpr int XBYTES = XPOS * 32;                                         // 18 400 B
constexpr int XBUF = 18432;
constexpr int YPOS = TH * OW;                                             // 224 pixels of dY per tile
constexpr int YPLANE = 228 * 16, YBUF = 8 * YPLANE;                       // as conv3x3_wgrad_bf16.hip
constexpr int YBASE = 2 * XBUF;
constexpr int XV = (XBYTES / 16 + 255) / 256;                             // 5 vectors per thread
constexpr int YV = YPOS * 8 / 256;                                        // 7
constexpr int KS = YPOS / 16, NSTEP = KS * 4;                             // 14 k-steps x 4 N-tiles per wave
constexpr int TILE_BYTES = YPOS * CO * 2;
constexpr int NCOL = 16 * PCH;                                            // 256 columns: tap x channel

struct SwP {
  const bf16_t* p;      // [N,115,115,16]
  const bf16_t* dy;     // [N,112,112,64]
  float* ws;            // [gridDim.x][64][256]
  int ntiles, N;
};

template <int I, int N, class F>
RP_DEV void static_for_s(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_s<I + 1, N>(f);
  }
}
template <int IMM>
RP_DEV void tr_read_s(unsigned long long& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int N>
RP_DEV void lds_wait_s(unsigned long long& a, unsigned long long& b, unsigned long long& c, unsigned long long& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
RP_DEV bf16x8 op8s(unsigned long long lo, unsigned long long hi) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 v;
  v[0] = lo; v[1] = hi;
  return __builtin_bit_cast(bf16x8, v);
}
constexpr int reads_of_s(int step) { return step < 0 || step >= NSTEP ? 0 : 2 + (step % 4 == 0 ? 2 : 0); }

__global__ __launch_bounds__(256, 1) void conv_stem_wgrad_kernel(SwP p) {
  __shared__ __attribute__((aligned(256))) unsigned char lds[2 * XBUF + 2 * YBUF];      // 95 232 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t16 = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5, coh = wave >> 1, nq = wave & 1;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // transpose reads: the lane supplies pixel row 8 hi + (t16 >> 2) (+ 4 half + 16 kk by immediate) and 4 channels; see conv3x3_wgrad_bf16.hip
  const unsigned aA = lds_base + YBASE + (4 * coh + 2 * g + ((t16 & 3) >> 1)) * YPLANE + ((t16 & 3) & 1) * 8 + (8 * hi + (t16 >> 2)) * 16;
  // P operand: N-tile nt = taps 2 nt (lanes g = 0) and 2 nt + 1 (g = 1: one position to the right), channels 4 (t16 & 3) .. + 3 of 16
  const unsigned aB = lds_base + (8 * hi + (t16 >> 2) + g) * 32 + (t16 & 3) * 8;

  f32x16 acc[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = zero16();

  const int G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((long long)p.ntiles * b / G), t1 = (int)((long long)p.ntiles * (b + 1) / G);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.p, 0, p.N * (PH * PW * 32), 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.ntiles * TILE_BYTES, 0x00020000);
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t RX[XV], RY[YV];

  auto fetch_x = [&](int i, int t, bool enable) {          // the strip's 5 rows of P are one contiguous run of 18 400 bytes
    const int img = t / TPI, ti = t - img * TPI;
    const int v = tid + 256 * i;
    const bool ok = enable && v * 16 < XBYTES;
    RX[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? (unsigned)((img * PH + TH * ti) * (PW * 32) + v * 16) : 0x80000000u, 0, 0);
  };
  auto fetch_y = [&](int i, int t, bool enable) {
    RY[i] = __builtin_amdgcn_raw_buffer_load_b128(yr, enable ? (unsigned)(t * TILE_BYTES + (tid + 256 * i) * 16) : 0x80000000u, 0, 0);
  };
  auto stash_x = [&](int i, int bufi) {
    const int v = min(tid + 256 * i, XBUF / 16 - 1);
    *reinterpret_cast<u32x4_t*>(lds + bufi * XBUF + v * 16) = RX[i];
  };
  auto stash_y = [&](int i, int bufi) {
    *reinterpret_cast<u32x4_t*>(lds + YBASE + bufi * YBUF + (tid & 7) * YPLANE + ((tid >> 3) + 32 * i) * 16) = RY[i];
  };

  auto tile = [&](auto bufc, int t) {
    constexpr int BI = decltype(bufc)::value;
    unsigned long long A[2][2], B[3][2];
    const unsigned aBq = aB + nq * (2 * PW * 32);                       // this wave's taps: rows u = 2 nq, 2 nq + 1
    auto issue = [&](auto stc) {
      constexpr int st = decltype(stc)::value;
      constexpr int kk = st / 4, i = st % 4, u = i >> 1, v = 2 * (i & 1), ty = (16 * kk) / OW;
      if constexpr (i == 0) {
        tr_read_s<BI * YBUF + (16 * kk) * 16>(A[kk & 1][0], aA);
        tr_read_s<BI * YBUF + (16 * kk + 4) * 16>(A[kk & 1][1], aA);
      }
      constexpr int imm = BI * XBUF + (16 * kk + 3 * ty + u * PW + v) * 32;
      tr_read_s<imm>(B[st % 3][0], aBq);
      tr_read_s<imm + 4 * 32>(B[st % 3][1], aBq);
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    const int tn2 = min(t + 2, t1 - 1);
    const bool more2 = t + 2 < t1;
    static_for_s<0, NSTEP>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (st + 2 < NSTEP) issue(std::integral_constant<int, st + 2>{});
      constexpr int kk = st / 4, i = st % 4;
      lds_wait_s<reads_of_s(st + 1) + reads_of_s(st + 2)>(A[kk & 1][0], A[kk & 1][1], B[st % 3][0], B[st % 3][1]);
      acc[i] = mfma_bf(op8s(A[kk & 1][0], A[kk & 1][1]), op8s(B[st % 3][0], B[st % 3][1]), acc[i]);
      if constexpr (st % 4 == 2) {
        constexpr int pc = st / 4;
        if constexpr (pc < XV) {
          stash_x(pc, BI ^ 1);
          fetch_x(pc, tn2, more2);
        } else if constexpr (pc < XV + YV) {
          stash_y(pc - XV, BI ^ 1);
          fetch_y(pc - XV, tn2, more2);
        }
      }
    });
    __syncthreads();
  };

  if (t0 < t1) {
#pragma unroll
    for (int i = 0; i < XV; ++i) fetch_x(i, t0, true);
#pragma unroll
    for (int i = 0; i < YV; ++i) fetch_y(i, t0, true);
#pragma unroll
    for (int i = 0; i < XV; ++i) stash_x(i, 0);
#pragma unroll
    for (int i = 0; i < YV; ++i) stash_y(i, 0);
#pragma unroll
    for (int i = 0; i < XV; ++i) fetch_x(i, min(t0 + 1, t1 - 1), t0 + 1 < t1);
#pragma unroll
    for (int i = 0; i < YV; ++i) fetch_y(i, min(t0 + 1, t1 - 1), t0 + 1 < t1);
  }
  __syncthreads();
  for (int t = t0; t < t1; t += 2) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < t1) tile(std::integral_constant<int, 1>{}, t + 1);
  }
  // partial [64 co][256]: column = 16 tap + ch, tap = 4 u + v; this wave: N-tiles nt = 4 nq + i = columns 32 nt + (lane & 31)
  float* o = p.ws + (long long)b * (CO * NCOL) + (32 * coh) * NCOL + 128 * nq + (lane & 31);
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[acc_row(r, hi) * NCOL + 32 * i] = acc[i][r];
}

// dW[co][ky][kx][ci] fp32 <- sum over workgroups of ws[b][co][16 (4 u + v) + 3 (2 (ky & 1) + (kx & 1)) + ci], u = ky >> 1, v = kx >> 1.
// 64 outputs x 4 slices of the partials per workgroup; fixed order: a slice front to back, then the four slices.
__global__ __launch_bounds__(256) void conv_stem_wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nblk) {
  __shared__ double part[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int idx = min(blockIdx.x * 64 + o, CO * 147 - 1);
  const int co = idx / 147, r = idx - co * 147, ky = r / 21, r2 = r - ky * 21, kx = r2 / 3, ci = r2 - kx * 3;
  const int col = 16 * (4 * (ky >> 1) + (kx >> 1)) + 3 * (2 * (ky & 1) + (kx & 1)) + ci;
  const float* src = ws + co * NCOL + col;
  const int k0 = (int)((long long)nblk * sl / 4), k1 = (int)((long long)nblk * (sl + 1) / 4);
  double s = 0.0;
  for (int k = k0; k < k1; ++k) s += (double)src[(long long)k * (CO * NCOL)];
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0 && blockIdx.x * 64 + o < CO * 147) dw[idx] = (float)(((part[0][o] + part[1][o]) + part[2][o]) + part[3][o]);
}

// P[n][a][b][(pa, pb, ci)] = bf16(Xp[n][2a + pa][2b + pb][ci]); channels 12..15 zero
__global__ __launch_bounds__(256) void stem_s2d_kernel(const float* __restrict__ xp, bf16_t* __restrict__ out, int N, int Hp, int Wp) {
  const long long total = (long long)N * (Hp / 2) * (Wp / 2);
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  const int bq = (int)(i % (Wp / 2));
  const long long r = i / (Wp / 2);
  const int a = (int)(r % (Hp / 2)), n = (int)(r / (Hp / 2));
  const float* r0 = xp + (((long long)n * Hp + 2 * a) * Wp + 2 * bq) * 3;
  const float* r1 = r0 + (long long)Wp * 3;
  float v[16];
#pragma unroll
  for (int e = 0; e < 6; ++e) { v[e] = r0[e]; v[6 + e] = r1[e]; }
  v[12] = v[13] = v[14] = v[15] = 0.f;
  uint4* o = reinterpret_cast<uint4*>(out + i * 16);
  o[0] = make_uint4(pk_bf16(v[0], v[1]), pk_bf16(v[2], v[3]), pk_bf16(v[4], v[5]), pk_bf16(v[6], v[7]));
  o[1] = make_uint4(pk_bf16(v[8], v[9]), pk_bf16(v[10], v[11]), 0u, 0u);
}

}  // namespace

extern "C" size_t rp_conv_stem_wgrad_workspace_bytes(int N) {
  const long long tiles = (long long)N * TPI;
  const long long nblk = tiles < 256 ? tiles : 256;
  return (size_t)N * PH * PW * 32 + (size_t)nblk * CO * NCOL * sizeof(float);
}

/* dw [64][7][7][3] fp32 from the framed fp32 image x_padded [N,230,230,3] and dY [N,112,112,64] bf16 (224 x 224 images only) */
extern "C" int rp_conv_stem_wgrad_bf16(const float* x_padded, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H,
                                       int W, void* stream) {
  if (!x_padded || !dy || !dw || !workspace || N <= 0) return RP_EBADSHAPE;
  if (H != 224 || W != 224) return RP_EUNSUPPORTED;
  if (((uintptr_t)x_padded | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) return RP_EALIGN;
  if (workspace_bytes < rp_conv_stem_wgrad_workspace_bytes(N)) return RP_EWORKSPACE;
  if ((long long)N * PH * PW * 32 >= (1LL << 31) || (long long)N * TPI * TILE_BYTES >= (1LL << 32)) return RP_EBADSHAPE;
  hipStream_t st = (hipStream_t)stream;
  bf16_t* P = (bf16_t*)workspace;
  float* ws = (float*)((char*)workspace + (size_t)N * PH * PW * 32);
  const long long npos = (long long)N * PH * PW;
  hipLaunchKernelGGL(stem_s2d_kernel, dim3((unsigned)((npos + 255) / 256)), dim3(256), 0, st, x_padded, P, N, 2 * PH, 2 * PW);
  RP_CHECK_LAUNCH();
  const int tiles = N * TPI, nblk = tiles < 256 ? tiles : 256;
  SwP p{P, (const bf16_t*)dy, ws, tiles, N};
  hipLaunchKernelGGL(conv_stem_wgrad_kernel, dim3(nblk), dim3(256), 0, st, p);
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv_stem_wgrad_reduce_kernel, dim3((CO * 147 + 63) / 64), dim3(256), 0, st, (const float*)ws, dw, nblk);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
