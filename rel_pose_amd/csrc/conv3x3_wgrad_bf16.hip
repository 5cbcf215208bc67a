// conv3x3_wgrad_bf16.hip -- weight gradient of the 3x3 / stride 1 / pad 1, 64 -> 64 convolutions of resnet.layer1 in the bf16 configuration
// (BASELINE.json configs[4]; autograd of src/model.py:131's BasicBlock convolutions):
//
//     dW[co][r][s][ci] = sum over (image, y, x) of dY[n, y, x, co] * X[n, y + r - 1, x + s - 1, ci]          (X zero outside the image)
//
// OUTPUT-STATIONARY over the pixel stream: the whole 64 x 576 result (36 tiles of 32 x 32 fp32) lives in the accumulators of ONE
// workgroup -- wave (coh, cih) holds the nine taps of its (output-channel half, input-channel half): 144 accumulator registers, one
// wave per SIMD -- while the workgroup walks its run of (image, 4-row strip) tiles; at the end every workgroup leaves a [64][3][3][64]
// fp32 partial and rp_conv3x3_c64_wgrad's second launch sums the partials in a fixed order (deterministic, no atomics) into bf16.
//   * both MFMA operands contract over PIXELS but lie pixel-major in memory (NHWC), so both are read with the LDS transpose read
//     (ds_read_b64_tr_b16: 4 pixels x 16 channels per 16-lane group -> a lane holds 4 consecutive pixels of its own channel); two
//     reads = one bf16x8 operand of v_mfma_f32_32x32x16_bf16;
//   * LDS images in PLANES, one per 16-byte channel chunk ([chunk][position][16 B], plane stride = 64 B mod 256 B): the 32 lanes of a
//     read's service group touch 4 planes x 4 consecutive positions = 4 x 64 B that tile the 256-byte bank row, and the address is
//     ONE register per operand kind plus an immediate that carries the k-step and the filter tap (a 16-pixel k-step that straddles an
//     image row, k-steps 3 and 10, uses a second register: + 2 halo positions on the upper half-wave);
//   * the X halo (6 x 58 positions) and the dY strip (224 pixels) of the NEXT tile are requested a tile ahead (buffer loads: zeros
//     outside the image for free) and land in the other LDS buffers piece by piece between the MFMAs: one barrier per tile;
//   * per 16-pixel k-step a wave issues 2 + 9 x 2 transpose reads for 9 MFMAs, operands two MFMAs ahead (hand-placed reads and
//     waits, conv3x3_bf16.hip explains why).
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef unsigned short bf16_t;
constexpr int C = 64, IW = 56, IH = 56, TH = 4, TPI = IH / TH, HC = IW + 2, HR = TH + 2;
constexpr int NPOS = HC * HR;                  // 348 halo positions
constexpr int XPLANE = 356 * 16;               // 5696 B = 64 (mod 256)
constexpr int XBUF = 8 * XPLANE;               // 45 568 B
constexpr int YPOS = TH * IW;                  // 224 pixels of dY per tile
constexpr int YPLANE = 228 * 16;               // 3648 B = 64 (mod 256)
constexpr int YBUF = 8 * YPLANE;               // 29 184 B
constexpr int YBASE = 2 * XBUF;                // dY buffers behind the two halo buffers
constexpr int XV = (NPOS * 8 + 255) / 256;     // 11 halo vectors per thread
constexpr int YV = YPOS * 8 / 256;             // 7 dY vectors per thread
constexpr int KS = YPOS / 16;                  // 14 k-steps of 16 pixels per tile
constexpr int NSTEP = KS * 9;                  // 126 MFMAs per wave and tile
constexpr int TILE_BYTES = YPOS * C * 2;       // 28 672 B of x or dY per tile in memory

struct WgP {
  const bf16_t* x;      // [N,56,56,64]
  const bf16_t* dy;     // [N,56,56,64]
  float* ws;            // [gridDim.x][64][9][64] partials
  int ntiles;
};

template <int I, int N, class F>
RP_DEV void static_for_w(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for_w<I + 1, N>(f);
  }
}
template <int IMM>
RP_DEV void tr_read(unsigned long long& d, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int N>
RP_DEV void lds_wait2(unsigned long long& a, unsigned long long& b, unsigned long long& c, unsigned long long& d) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N));
}
RP_DEV bf16x8 op8(unsigned long long lo, unsigned long long hi) {
  typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
  u64x2 v;
  v[0] = lo; v[1] = hi;
  return __builtin_bit_cast(bf16x8, v);
}
constexpr int reads_of(int step) { return step < 0 || step >= NSTEP ? 0 : 2 + (step % 9 == 0 ? 2 : 0); }

__global__ __launch_bounds__(256, 1) void conv3x3_c64_wgrad_kernel(WgP p) {
  __shared__ __attribute__((aligned(256))) unsigned char lds[2 * XBUF + 2 * YBUF];      // 149 504 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int t16 = lane & 15, g = (lane >> 4) & 1, hi = lane >> 5, coh = wave >> 1, cih = wave & 1;
  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  // transpose-read addresses of this lane: it SUPPLIES row 8 hi + (t16 >> 2) (+ 4 half, + 16 kk by immediate) and channels
  // 32 half-of-64 + 16 g + 4 (t16 & 3) .. + 3 of it, and RECEIVES channel 16 g + t16 of the four rows
  const int sub = ((t16 & 3) & 1) * 8, rowl = (8 * hi + (t16 >> 2)) * 16;
  const unsigned aA = lds_base + YBASE + (4 * coh + 2 * g + ((t16 & 3) >> 1)) * YPLANE + sub + rowl;
  const unsigned aB = lds_base + (4 * cih + 2 * g + ((t16 & 3) >> 1)) * XPLANE + sub + rowl;
  const unsigned aB2 = aB + hi * 32;                                     // k-steps 3 and 10: the upper 8 pixels are in the next image row

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero16();

  const int G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((long long)p.ntiles * b / G), t1 = (int)((long long)p.ntiles * (b + 1) / G);
  const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.ntiles * TILE_BYTES, 0x00020000);
  const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc((void*)p.dy, 0, p.ntiles * TILE_BYTES, 0x00020000);
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t RX[XV], RY[YV];

  auto fetch_x = [&](int i, int t, bool enable) {
    const int pos = (tid >> 3) + 32 * i;
    const int row = (int)((unsigned)pos / (unsigned)HC), col = pos - row * HC;
    const int img = t / TPI, ti = t - img * TPI, gr = TH * ti - 1 + row;
    const bool ok = enable && (i < XV - 1 || pos < NPOS) && col >= 1 && col <= IW && gr >= 0 && gr < IH;
    const unsigned off = (unsigned)((((img * IH + gr) * IW + (col - 1)) * C + (tid & 7) * 8) * 2);
    RX[i] = __builtin_amdgcn_raw_buffer_load_b128(xr, ok ? off : 0x80000000u, 0, 0);
  };
  auto fetch_y = [&](int i, int t, bool enable) {
    RY[i] = __builtin_amdgcn_raw_buffer_load_b128(yr, enable ? (unsigned)(t * TILE_BYTES + (tid + 256 * i) * 16) : 0x80000000u, 0, 0);
  };
  auto stash_x = [&](int i, int bufi) {
    const int pos = min((tid >> 3) + 32 * i, NPOS);                      // (past the last position: the plane's pad slot)
    *reinterpret_cast<u32x4_t*>(lds + bufi * XBUF + (tid & 7) * XPLANE + pos * 16) = RX[i];
  };
  auto stash_y = [&](int i, int bufi) {
    *reinterpret_cast<u32x4_t*>(lds + YBASE + bufi * YBUF + (tid & 7) * YPLANE + ((tid >> 3) + 32 * i) * 16) = RY[i];
  };

  auto tile = [&](auto bufc, int t) {
    constexpr int BI = decltype(bufc)::value;
    unsigned long long A[2][2], B[3][2];
    // operand reads of MFMA step `st` = (k-step kk = st / 9, tap = st % 9): [A(kk) with the first tap,] B(kk, tap)
    auto issue = [&](auto stc) {
      constexpr int st = decltype(stc)::value;
      constexpr int kk = st / 9, tap = st % 9, r = tap / 3, s_ = tap % 3;
      constexpr int ty = (16 * kk) / IW;
      if constexpr (tap == 0) {
        tr_read<BI * YBUF + (16 * kk) * 16>(A[kk & 1][0], aA);
        tr_read<BI * YBUF + (16 * kk + 4) * 16>(A[kk & 1][1], aA);
      }
      constexpr int imm = BI * XBUF + (16 * kk + 2 * ty + r * HC + s_) * 16;
      if constexpr (kk == 3 || kk == 10) {
        tr_read<imm>(B[st % 3][0], aB2);
        tr_read<imm + 64>(B[st % 3][1], aB2);
      } else {
        tr_read<imm>(B[st % 3][0], aB);
        tr_read<imm + 64>(B[st % 3][1], aB);
      }
    };
    issue(std::integral_constant<int, 0>{});
    issue(std::integral_constant<int, 1>{});
    const int tn2 = min(t + 2, t1 - 1);
    const bool more2 = t + 2 < t1;
    static_for_w<0, NSTEP>([&](auto stc) {
      constexpr int st = decltype(stc)::value;
      if constexpr (st + 2 < NSTEP) issue(std::integral_constant<int, st + 2>{});
      constexpr int kk = st / 9, tap = st % 9;
      lds_wait2<reads_of(st + 1) + reads_of(st + 2)>(A[kk & 1][0], A[kk & 1][1], B[st % 3][0], B[st % 3][1]);
      acc[tap] = mfma_bf(op8(A[kk & 1][0], A[kk & 1][1]), op8(B[st % 3][0], B[st % 3][1]), acc[tap]);
      // between the MFMAs: the next tile's operands registers -> LDS (other buffers), the tile after that requested into the registers
      if constexpr (st % 7 == 3) {
        constexpr int i = st / 7;
        if constexpr (i < XV) {
          stash_x(i, BI ^ 1);
          fetch_x(i, tn2, more2);
        } else if constexpr (i < XV + YV) {
          stash_y(i - XV, BI ^ 1);
          fetch_y(i - XV, tn2, more2);
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
  // partial [64 co][9 taps][64 ci] of this workgroup: rows co = 32 coh + acc_row(r, hi), columns 32 cih + l31
  float* o = p.ws + (long long)b * (C * 9 * C) + (32 * coh) * (9 * C) + 32 * cih + (lane & 31);
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[acc_row(r, hi) * (9 * C) + tap * C] = acc[tap][r];
}

// 64 outputs x 4 slices of the partials per workgroup; fixed order: a slice front to back, then the four slices
__global__ __launch_bounds__(256) void conv3x3_c64_wgrad_reduce_kernel(const float* __restrict__ ws, bf16_t* __restrict__ dw, int nblk) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;                                    // < 36864
  const int k0 = (int)((long long)nblk * sl / 4), k1 = (int)((long long)nblk * (sl + 1) / 4);
  float s = 0.f;
  for (int k = k0; k < k1; ++k) s += ws[(long long)k * (C * 9 * C) + idx];
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0) {
    const float v = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
    dw[idx] = (bf16_t)(pk_bf16(v, 0.f) & 0xffffu);
  }
}

}  // namespace

extern "C" int rp_conv3x3_c64_wgrad_blocks(int N) {
  const int tiles = N * TPI;
  return tiles < 256 ? tiles : 256;
}
extern "C" size_t rp_conv3x3_c64_wgrad_workspace_bytes(int N) {
  return (size_t)rp_conv3x3_c64_wgrad_blocks(N) * C * 9 * C * sizeof(float);
}

/* dw [64][3][3][64] bf16 = sum over pixels of dY (x) shifted X (see the file header); workspace = per-workgroup fp32 partials */
extern "C" int rp_conv3x3_c64_wgrad_bf16(const void* x, const void* dy, void* dw, void* workspace, size_t workspace_bytes, int N, int H, int W,
                                         void* stream) {
  if (!x || !dy || !dw || !workspace || N <= 0) return RP_EBADSHAPE;
  if (H != IH || W != IW) return RP_EUNSUPPORTED;
  if ((size_t)N * IH * IW * C * 2 > 0x7fffffffull) return RP_EUNSUPPORTED;      // buffer-resource range (see rp_conv3x3_c64_bf16)
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) return RP_EALIGN;
  if (workspace_bytes < rp_conv3x3_c64_wgrad_workspace_bytes(N)) return RP_EWORKSPACE;
  const int nblk = rp_conv3x3_c64_wgrad_blocks(N);
  WgP p{(const bf16_t*)x, (const bf16_t*)dy, (float*)workspace, N * TPI};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3x3_c64_wgrad_kernel, dim3(nblk), dim3(256), 0, st, p);
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv3x3_c64_wgrad_reduce_kernel, dim3(C * 9 * C / 64), dim3(256), 0, st, (const float*)workspace, (bf16_t*)dw, nblk);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
