// conv3x3_wgrad_f32.hip -- weight gradient of the 3x3 / stride 1 / pad 1, 64 -> 64 convolutions of resnet.layer1 in EXACT fp32 (the headline
// configuration; autograd of src/model.py:131's BasicBlock convolutions, which the reference runs through cuDNN):
//
//     dW[co][r][s][ci] = sum over (image, y, x) of dY[n, y, x, co] * X[n, y + r - 1, x + s - 1, ci]          (X zero outside the image)
//
// MIOpen's backward-weights solver runs this shape at 0.55 of the fp32 MFMA peak inside the step (345 us at 128 images).  Here it is
// dw192_f32.hip's OUTPUT-STATIONARY stream with a nine-tap B operand: the whole 64 x 576 result (36 tiles of 32 x 32) lives in ONE
// workgroup's accumulators -- wave (coh, cih) holds the nine taps of its (output-channel half, input-channel half), 144 registers, one
// wave per SIMD -- while the workgroup walks its run of image ROWS; per-workgroup fp32 partials, fixed-order reduce (no atomics).
//   * both operands contract over pixels and lie pixel-major (NHWC): with v_mfma_f32_32x32x2_f32 an operand is ONE conflict-free
//     ds_read_b32 per lane (lane = channel, half-wave = which of the k-step's two pixels) -- rows are copied as they lie by LDS-DMA;
//   * X rows live in a RING of four 58-position slots (position 0 and 57 = the zero padding, written once): a tile (one image row)
//     needs rows y - 1, y, y + 1 and fetches only row y + 2; the taps of a row outside the image read a fifth, permanently zero slot
//     (an address select per tile, no branch in the loop);
//   * per 2-pixel k-step a wave issues 1 + 9 ds_read_b32 (immediate offsets: next to fp32 MFMAs LDS instructions are free, VALU
//     instructions are not -- profiles/r5_shadow_lab.txt) for 9 MFMAs, operands one step ahead, waits written out by hand; the next
//     tile's DMA pieces go out one per two k-steps behind the MFMAs.
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

constexpr int C = 64, IW = 56, IH = 56;
constexpr int XROW = (IW + 2) * C;             // floats of one X row slot (58 positions)
constexpr int YROW = IW * C;                   // floats of one dY row
constexpr int ROW_BYTES = IW * C * 4;          // 14 336 B of x or dY per image row in memory = 14 DMA pieces of 1 KB
constexpr int NPIECE = ROW_BYTES / 1024;
constexpr int KS = IW / 2;                     // 28 k-steps of 2 pixels per tile

struct WgF {
  const float* x;       // [N,56,56,64]
  const float* dy;      // [N,56,56,64]
  float* ws;            // [gridDim.x][64][9][64] partials
  int ntiles;           // N * 56 image rows
};

RP_DEV void glds16r(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
RP_DEV const void* uniform_ptr_r(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
template <int OFF> RP_DEV float rd32(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N, class F> RP_DEV void sfor(F&& f) {
  if constexpr (N > 0) {
    sfor<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__global__ __launch_bounds__(256, 1) void conv3x3_c64_wgrad_f32_kernel(WgF p) {
  __shared__ __attribute__((aligned(16))) float Xr[5][XROW];      // 74 240 B: ring slots 0 .. 3, slot 4 = a row of zeros
  __shared__ __attribute__((aligned(16))) float Ys[2][YROW];      // 28 672 B
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int coh = wave >> 1, cih = wave & 1;
  const int G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((long long)p.ntiles * b / G), t1 = (int)((long long)p.ntiles * (b + 1) / G);
  // the zero padding of the four ring slots (DMA only ever writes positions 1 .. 56) and the zero row
  for (int i = tid; i < 4 * 2 * C; i += 256) Xr[i >> 7][((i >> 6) & 1) * (IW + 1) * C + (i & 63)] = 0.f;
  for (int i = tid; i < XROW; i += 256) Xr[4][i] = 0.f;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = zero16();

  const unsigned xs0 = (unsigned)(size_t)(rp_lds_ptr_t)(&Xr[0][0]), ys0 = (unsigned)(size_t)(rp_lds_ptr_t)(&Ys[0][0]);
  // DMA piece q (0 .. 13) of image row `row` of x into its ring slot / of dY into its buffer; pieces of a row are spread over the waves
  auto dma_x = [&](int row, int q) {
    glds16r(uniform_ptr_r(p.x + (long long)row * YROW), (unsigned)(q * 1024 + lane * 16), xs0 + (row & 3) * (XROW * 4) + C * 4 + q * 1024);
  };
  auto dma_y = [&](int row, int q) {
    glds16r(uniform_ptr_r(p.dy + (long long)row * YROW), (unsigned)(q * 1024 + lane * 16), ys0 + (row & 1) * (YROW * 4) + q * 1024);
  };
  if (t0 < t1) {
    const int y0 = t0 % IH;
    for (int q = wave; q < NPIECE; q += 4) {
      if (y0 > 0) dma_x(t0 - 1, q);
      dma_x(t0, q);
      if (t0 + 1 < p.ntiles) dma_x(t0 + 1, q);    // (the bottom neighbour, or -- last row of an image -- the NEXT tile's centre row)
      dma_y(t0, q);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  for (int t = t0; t < t1; ++t) {
    const int y = t % IH;
    const bool top = y > 0, bot = y < IH - 1;          // rows y - 1 / y + 1 exist in this image
    const int xn = min(t + 2, p.ntiles - 1), yn = min(t + 1, p.ntiles - 1);      // next tile's new rows (re-fetches at the very end: harmless)
    // operand addresses: lane = channel 32 half + l31, half-wave = pixel 2 k + hi; the immediate carries the k-step (and the tap's column)
    const unsigned ya = ys0 + (t & 1) * (YROW * 4) + (32 * coh + l31) * 4 + hi * (C * 4);
    const unsigned xoff = (32 * cih + l31) * 4 + hi * (C * 4);
    // (a row outside the image reads the zero slot: its three MFMAs per k-step add zeros -- 2 of 56 tiles -- and the loop has no branch)
    const unsigned xa0 = xs0 + (top ? ((t - 1) & 3) : 4) * (XROW * 4) + xoff, xa1 = xs0 + (t & 3) * (XROW * 4) + xoff;
    const unsigned xa2 = xs0 + (bot ? ((t + 1) & 3) : 4) * (XROW * 4) + xoff;
    float a, bq[9];
    a = rd32<0>(ya);
    bq[0] = rd32<0>(xa0); bq[1] = rd32<C * 4>(xa0); bq[2] = rd32<2 * C * 4>(xa0);
    bq[3] = rd32<0>(xa1); bq[4] = rd32<C * 4>(xa1); bq[5] = rd32<2 * C * 4>(xa1);
    bq[6] = rd32<0>(xa2); bq[7] = rd32<C * 4>(xa2); bq[8] = rd32<2 * C * 4>(xa2);
    sfor<KS>([&](auto kc) {
      constexpr int k = kc;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]), "+v"(bq[4]), "+v"(bq[5]), "+v"(bq[6]),
                   "+v"(bq[7]), "+v"(bq[8]));
      float an = 0.f, bn[9] = {};
      if constexpr (k + 1 < KS) {
        constexpr int o = 2 * (k + 1) * C * 4;
        an = rd32<o>(ya);
        bn[0] = rd32<o>(xa0); bn[1] = rd32<o + C * 4>(xa0); bn[2] = rd32<o + 2 * C * 4>(xa0);
        bn[3] = rd32<o>(xa1); bn[4] = rd32<o + C * 4>(xa1); bn[5] = rd32<o + 2 * C * 4>(xa1);
        bn[6] = rd32<o>(xa2); bn[7] = rd32<o + C * 4>(xa2); bn[8] = rd32<o + 2 * C * 4>(xa2);
      }
      if constexpr (k % 2 == 0 && k < 16) {            // the next tile's rows: one DMA piece per two k-steps, behind the MFMAs
        constexpr int j = k / 2, q4 = 4 * (j & 3);
        if (wave + q4 < NPIECE) {
          if (j < 4) dma_x(xn, wave + q4);
          else dma_y(yn, wave + q4);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 9; ++i) acc[i] = mfma32(a, bq[i], acc[i]);
      if constexpr (k + 1 < KS) {
        a = an;
#pragma unroll
        for (int i = 0; i < 9; ++i) bq[i] = bn[i];
      }
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  // partial [64 co][9 taps][64 ci] of this workgroup: rows co = 32 coh + acc_row(r, hi), columns 32 cih + l31
  float* o = p.ws + (long long)b * (C * 9 * C) + (32 * coh) * (9 * C) + 32 * cih + l31;
#pragma unroll
  for (int tap = 0; tap < 9; ++tap)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[acc_row(r, hi) * (9 * C) + tap * C] = acc[tap][r];
}

// 64 outputs x 4 slices of the partials per workgroup; fixed order: a slice front to back, then the four slices
__global__ __launch_bounds__(256) void conv3x3_c64_wgrad_f32_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nblk) {
  __shared__ float part[4][64];
  const int o = threadIdx.x & 63, sl = threadIdx.x >> 6;
  const int idx = blockIdx.x * 64 + o;                                    // < 36864
  const int k0 = (int)((long long)nblk * sl / 4), k1 = (int)((long long)nblk * (sl + 1) / 4);
  float s = 0.f;
  for (int k = k0; k < k1; ++k) s += ws[(long long)k * (C * 9 * C) + idx];
  part[sl][o] = s;
  __syncthreads();
  if (sl == 0) dw[idx] = ((part[0][o] + part[1][o]) + part[2][o]) + part[3][o];
}

}  // namespace

extern "C" int rp_conv3x3_c64_wgrad_f32_blocks(int N) {
  const int tiles = N * IH;
  return tiles < 256 ? tiles : 256;
}
extern "C" size_t rp_conv3x3_c64_wgrad_f32_workspace_bytes(int N) {
  return (size_t)rp_conv3x3_c64_wgrad_f32_blocks(N) * C * 9 * C * sizeof(float);
}

/* dw [64][3][3][64] fp32 = sum over pixels of dY (x) shifted X (see the file header); workspace = per-workgroup fp32 partials */
extern "C" int rp_conv3x3_c64_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H, int W,
                                        void* stream) {
  if (!x || !dy || !dw || !workspace || N <= 0) return RP_EBADSHAPE;
  if (H != IH || W != IW) return RP_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)dy | (uintptr_t)dw | (uintptr_t)workspace) & 15) return RP_EALIGN;
  if (workspace_bytes < rp_conv3x3_c64_wgrad_f32_workspace_bytes(N)) return RP_EWORKSPACE;
  const int nblk = rp_conv3x3_c64_wgrad_f32_blocks(N);
  WgF p{x, dy, (float*)workspace, N * IH};
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(conv3x3_c64_wgrad_f32_kernel, dim3(nblk), dim3(256), 0, st, p);
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL(conv3x3_c64_wgrad_f32_reduce_kernel, dim3(C * 9 * C / 64), dim3(256), 0, st, (const float*)workspace, dw, nblk);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
