// conv3x3_bf16.hip -- the 3x3 / stride 1 / pad 1, 64 -> 64 convolutions of the CNN front-end's layer1 in the bf16 configuration
// (BASELINE.json configs[4]; reference src/model.py:131 `self.resnet.layer1(x)` = 2 BasicBlocks = 4 such convolutions on [2B,64,56,56],
// torchvision resnet.py BasicBlock.forward; and, with the rotated / transposed filter, their input gradients).
//
// NHWC bf16 in HBM (channels-last, what the front-end already keeps), fp32 accumulation on v_mfma_f32_32x32x16_bf16.
//   * implicit GEMM with the INPUT HALO resident in LDS: a workgroup owns a run of (image, 4-row strip) tiles; the strip's 6 x 58
//     halo positions x 64 channels (44.5 KB) are fetched once (global -> registers -> LDS, 16 B per lane) and read nine times, once
//     per filter tap -- MIOpen's / CK's implicit GEMMs re-gather the input per tap through L2 (9x the bytes through the TA path);
//   * the FILTER lives in registers: wave (mh, nh) computes output channels 32 nh .. 32 nh + 31 and keeps its 36 A-operand fragments
//     (9 taps x 4 k-steps x 8 bf16 per lane = 144 VGPRs) for the whole launch -- loaded from L2 once per workgroup, no LDS traffic;
//   * eight waves, two per SIMD (launch bounds 512 x 2, <= 256 registers): wave (mq, nh) owns M-tiles 2 mq, 2 mq + 1 of the strip's seven
//     32-pixel tiles (the eighth is a dummy) -- a wave parked on a memory or LDS wait leaves the matrix pipe to its SIMD partner --, D[out channel][pixel] so that a lane owns one
//     pixel and writes 4 consecutive channels (8 B) per store;
//   * the next tile's halo is requested before the current tile's 144 MFMAs and lands in the other LDS buffer after them: one
//     workgroup barrier per tile;
//   * optional on-load BatchNorm + ReLU of the INPUT (`scale`, `shift` per input channel: x <- max(0, x * scale + shift), applied
//     once per halo element between the global load and the LDS store; padding stays exactly 0) -- the previous layer's
//     BatchNorm-apply pass folded into this convolution's operand load;
//   * optional BatchNorm batch statistics of the OUTPUT from the epilogue: per-channel sums of y and y^2 (of the bf16-rounded values
//     that are stored), one [2][64] double partial per workgroup for rp_bn_stats_from_partials.
// LDS layout of a halo buffer: eight PLANES, one per 16-byte channel chunk, each [348 positions][16 B] with a plane stride of 349 x 16 B.
//   * operand read (ds_read_b128; lane = pixel, k-slots 8 hi .. 8 hi + 7 of k-step ks = chunk 2 ks + hi): the 16 lanes of a service
//     group are consecutive pixels of one row = 256 contiguous bytes of one plane: conflict-free, and the address is ONE register per
//     M-tile (position x 16 + hi x plane) plus an immediate (k-step plane pair, tap offset): no address arithmetic in the main loop;
//   * halo store (ds_write_b128; 8 consecutive lanes = the 8 chunks of one position): 349 mod 16 = 13 is odd, so the 8 planes of a
//     position land in 8 different 16-byte bank groups.
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef unsigned short bf16_t;
constexpr int C = 64;                    // channels in and out
constexpr int IW = 56, IH = 56;          // image size
constexpr int TH = 4;                    // output rows per tile
constexpr int TPI = IH / TH;             // 14 tiles per image
constexpr int HC = IW + 2, HR = TH + 2;  // halo: 58 columns x 6 rows
constexpr int NPOS = HC * HR;            // 348 positions of 128 B
constexpr int PLANE = (NPOS + 1) * 16;   // 5584 B: plane stride (349 positions: odd multiple of 16 B)
constexpr int BUF = 8 * PLANE;           // 44 672 B per buffer
constexpr int NVEC = NPOS * 8;           // 2784 16-byte vectors
#ifndef RP_CONV_NT
#define RP_CONV_NT 256
#endif
constexpr int NT = RP_CONV_NT;           // threads per workgroup: 256 = one wave per SIMD (4 M-tiles per wave), 512 = two (2 M-tiles)
constexpr int VPT = (NVEC + NT - 1) / NT;  // 6 halo vectors per thread
constexpr int STAGE = TH * IW * C * 2;   // 28 672 B: a tile's output, contiguous in y
constexpr int OVEC = STAGE / 16;         // 1792 16-byte output vectors per tile
constexpr int OPT = (OVEC + NT - 1) / NT;  // 4 per thread (the last one on half of the threads)

struct ConvP {
  const bf16_t* x;       // [N,56,56,64]
  const bf16_t* w;       // [64 out][3][3][64 in]
  bf16_t* y;             // [N,56,56,64]
  const float* scale;    // [64] or nullptr
  const float* shift;    // [64]
  double* stats;         // [gridDim.x][2][64] or nullptr
  int ntiles;            // N * 14
};

template <int I, int N, class F>
RP_DEV void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}
// operand read by hand: hipcc sinks every C++-level LDS read next to its MFMA (one register set, `s_waitcnt lgkmcnt(0)` in between:
// a full LDS round trip per MFMA group with one wave per SIMD), whatever the source order or the sched_group_barrier hints say.
// A volatile asm read keeps its place; the matching wait is an asm that passes the registers through ("+v"), so the MFMAs stay behind
// it.  The compiler's own waits (it does not see these reads) are only ever stricter than needed, never too lenient.
template <int IMM>
RP_DEV void lds_read128(bf16x8& d, unsigned addr) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(IMM));
}
template <int N>
RP_DEV void lds_wait(bf16x8& a0, bf16x8& a1, bf16x8& a2, bf16x8& a3) {
  asm volatile("s_waitcnt lgkmcnt(%4)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "n"(N));
}
template <int N>
RP_DEV void lds_wait(bf16x8& a0, bf16x8& a1) {
  asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a0), "+v"(a1) : "n"(N));
}

RP_DEV float bf_lo(unsigned w) { return __uint_as_float(w << 16); }
RP_DEV float bf_hi(unsigned w) { return __uint_as_float(w & 0xffff0000u); }

template <bool BN, bool STATS>
__global__ __launch_bounds__(NT, NT / 256) void conv3x3_c64_kernel(ConvP p) {
  __shared__ __attribute__((aligned(256))) unsigned char lds[2 * BUF + 2 * STAGE];     // 146 688 B
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5, mq = wave >> 1, nh = wave & 1;       // wave (mq, nh): M-tiles 2 mq, 2 mq + 1; channels 32 nh ..

  // ---- this wave's filter fragments: A operand, lane = (out channel 32 nh + l31, k-slots 8 hi .. 8 hi + 7 of each 16-channel step).
  // The 72 KB filter goes global -> LDS as whole lines (18 coalesced 16-byte loads per thread) and LDS -> registers from there: read
  // straight from global memory every wave instruction touches 64 different rows (8x the L2 requests).  Launch time is the same either
  // way (the launch-size-independent part of this kernel is ~3 us: profiles/r5_conv3x3.txt).
  bf16x8 wf[9][4];
  {
    constexpr int WV = 64 * 9 * C * 2 / 16;                      // 4608 vectors
#pragma unroll
    for (int i = 0; i < WV / NT; ++i) {
      const int v = tid + NT * i;
      *reinterpret_cast<uint4*>(lds + v * 16) = *reinterpret_cast<const uint4*>(p.w + v * 8);
    }
    __syncthreads();
    const unsigned char* wrow = lds + ((32 * nh + l31) * (9 * C) + 8 * hi) * 2;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wf[tap][ks] = *reinterpret_cast<const bf16x8*>(wrow + (tap * C + 16 * ks) * 2);
    __syncthreads();                                             // (the halo buffers reuse this memory)
  }

  // ---- halo vectors of this thread: vector v = tid + 256 i -> (position v >> 3, chunk v & 7 = tid & 7); the descriptors are recomputed
  // from (tid, i) where they are used (a handful of VALU per vector per tile) instead of living in 33 registers
  auto vec_desc = [&](int i, int& row, int& col, int& pos) {
    pos = (tid >> 3) + (NT / 8) * i;
    row = (int)((unsigned)pos / (unsigned)HC);
    col = pos - row * HC;
  };
  float sc[8], sh[8];
  if (BN) {
#pragma unroll
    for (int k = 0; k < 8; ++k) { sc[k] = p.scale[8 * (tid & 7) + k]; sh[k] = p.shift[8 * (tid & 7) + k]; }
  }

  // ---- pixel operand addresses: M-tile j of this wave = pixels 32 (4 mh + j) + l31 of the strip (clamped: the eighth tile is a dummy)
  constexpr int MT = 1024 / NT;            // M-tiles per wave (8 slots for the strip's 7)
  int e0[MT], ypix[MT], yswz[MT];
#pragma unroll
  for (int j = 0; j < MT; ++j) {
    const int m = min(32 * (MT * mq + j) + l31, TH * IW - 1);
    const int ty = m / IW, tx = m - ty * IW;
    // staged output: pixel m = 128 B, its 16-byte chunk c at slot c ^ ((m >> 1) & 7); this lane's pieces: chunks 4 nh + g, half hi
    ypix[j] = m * 128 + 8 * hi;
    yswz[j] = (m >> 1) & 7;
    e0[j] = (ty * HC + tx) * 16 + hi * PLANE;
  }

  const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)lds;
  float ssum[16], ssq[16];
  if (STATS) {
#pragma unroll
    for (int r = 0; r < 16; ++r) ssum[r] = ssq[r] = 0.f;
  }

  const int G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((long long)p.ntiles * b / G), t1 = (int)((long long)p.ntiles * (b + 1) / G);
  uint4 R[VPT];

  // BRANCH-FREE: a buffer load whose offset lies past `num_records` returns zeros, so the padding / predicated-off vectors cost no
  // branch and no select.  (With exec-masked global loads inside the unrolled tile body hipcc sinks each load into its own `if`
  // block and waits `vmcnt(0)` right behind it: eleven exposed memory round trips per tile.)
  const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, p.ntiles * (TH * IW * C * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc((void*)p.y, 0, p.ntiles * STAGE, 0x00020000);
  // tile t's staged output -> y: 7 whole-line 16-byte stores per thread (the 8-byte accumulator pieces stored straight to global memory
  // were store-ISSUE-bound: 16 dwordx2 per lane per tile cost ~9 k cycles, twice the tile's MFMA time)
  auto writeout1 = [&](int k, int t, bool enable, int stagebytes) {
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const int u = min(tid + NT * k, OVEC - 1), m = u >> 3, slot = u & 7;
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(lds + 2 * BUF + stagebytes + u * 16);
    const unsigned off = (unsigned)(t * STAGE + m * 128 + ((slot ^ ((m >> 1) & 7)) << 4));
    __builtin_amdgcn_raw_buffer_store_b128(v, yrsrc, (enable && tid + NT * k < OVEC) ? off : 0x80000000u, 0, 0);
  };
  auto fetch1 = [&](int i, int t, bool enable) {   // vector i of tile t: global -> register (zeros outside the image)
    int row, col, pos;
    vec_desc(i, row, col, pos);
    const int img = t / TPI, ti = t - img * TPI;
    const int gr = TH * ti - 1 + row;
    const bool ok = enable && (i < VPT - 1 || pos < NPOS) && col >= 1 && col <= IW && gr >= 0 && gr < IH;
    const unsigned off = (unsigned)((((img * IH + gr) * IW + (col - 1)) * C + (tid & 7) * 8) * 2);
    typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
    const u32x4_t v = __builtin_amdgcn_raw_buffer_load_b128(xrsrc, ok ? off : 0x80000000u, 0, 0);
    R[i] = make_uint4(v[0], v[1], v[2], v[3]);
  };
  auto stash1 = [&](int i, int t, int bufbytes) {   // register -> LDS (BatchNorm + ReLU on the way; padding stays 0)
    int row, col, pos;
    vec_desc(i, row, col, pos);
    uint4 v = R[i];
    if (BN) {
      const int img = t / TPI, ti = t - img * TPI;
      const int gr = TH * ti - 1 + row;
      const bool inside = col >= 1 && col <= IW && gr >= 0 && gr < IH;
      unsigned* wv = reinterpret_cast<unsigned*>(&v);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float a = fmaxf(fmaf(bf_lo(wv[k]), sc[2 * k], sh[2 * k]), 0.f);
        const float c = fmaxf(fmaf(bf_hi(wv[k]), sc[2 * k + 1], sh[2 * k + 1]), 0.f);
        wv[k] = inside ? pk_bf16(a, c) : 0u;
      }
    }
    // (the 64 vectors past the last position of the last piece land in the pad position of each plane / the gap behind plane 7)
    *reinterpret_cast<uint4*>(lds + bufbytes + (tid & 7) * PLANE + min(pos, NPOS) * 16) = v;
  };

  // one tile: 36 (tap, k-step) groups of 4 MFMAs, operands fetched one group ahead; between them, piece by piece, the NEXT tile's halo
  // goes registers -> LDS (other buffer) and the tile after that is requested into the freed registers
#ifdef RP_CONV_PROBE
  unsigned long long pr_loop = 0, pr_epi = 0, pr_bar = 0, pr_n = 0;
#define PROBE_T() __builtin_readcyclecounter()
#endif
  auto tile = [&](auto bufc, int t) {
    constexpr int BI = decltype(bufc)::value;
#ifdef RP_CONV_PROBE
    const unsigned long long c0 = PROBE_T();
#endif
    const unsigned char* lb = lds + BI * BUF;
    f32x16 acc[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) acc[j] = zero16();
    constexpr int DEPTH = 2;                     // operand groups in flight ahead of the MFMAs (LDS latency under load > one group)
    bf16x8 a[DEPTH + 1][MT];
    unsigned ea[MT];
#pragma unroll
    for (int j = 0; j < MT; ++j) ea[j] = (unsigned)(BI * BUF + e0[j]) + lds_base;
    auto lda = [&](auto stepc, bf16x8 (&d)[MT]) {
      constexpr int step = decltype(stepc)::value;
      constexpr int r = step / 12, s_ = (step / 4) % 3, ks = step & 3;
#pragma unroll
      for (int j = 0; j < MT; ++j) lds_read128<2 * ks * PLANE + (r * HC + s_) * 16>(d[j], ea[j]);
    };
    static_for<0, DEPTH>([&](auto d) { lda(d, a[decltype(d)::value]); });
    const int tn1 = min(t + 1, t1 - 1), tn2 = min(t + 2, t1 - 1);
    const bool more2 = t + 2 < t1;
    static_for<0, 36>([&](auto stepc) {
      constexpr int step = decltype(stepc)::value;
      if constexpr (step + DEPTH < 36) lda(std::integral_constant<int, step + DEPTH>{}, a[(step + DEPTH) % (DEPTH + 1)]);
      constexpr int ahead = (step + DEPTH < 36 ? DEPTH : 35 - step) * MT;      // reads issued after this group's
      bf16x8(&ac)[MT] = a[step % (DEPTH + 1)];
      if constexpr (MT == 4) lds_wait<ahead>(ac[0], ac[1], ac[2], ac[3]);
      else lds_wait<ahead>(ac[0], ac[1]);
#pragma unroll
      for (int j = 0; j < MT; ++j) acc[j] = mfma_bf(wf[step >> 2][step & 3], ac[j], acc[j]);
      constexpr int SPO = 36 / OPT, SPV = 36 / VPT;      // MFMA groups between two output / two halo pieces
      if constexpr (step % SPO == 1 && step / SPO < OPT) writeout1(step / SPO, t - 1, t > t0, (BI ^ 1) * STAGE);
      if constexpr (step % SPV == SPV - 1 && step / SPV < VPT) {
        constexpr int i = step / SPV;
        stash1(i, tn1, (BI ^ 1) * BUF);          // (after the last tile: a harmless re-store of stale registers)
        fetch1(i, tn2, more2);
      }
    });
#ifdef RP_CONV_PROBE
    const unsigned long long c1 = PROBE_T();
#endif
    // ---- epilogue: lane = pixel, registers 4 g .. 4 g + 3 = channels 32 nh + 8 g + 4 hi + {0..3} -> the stage (written out during the
    // next tile's MFMAs)
    unsigned char* sb = lds + 2 * BUF + BI * STAGE;
#pragma unroll
    for (int j = 0; j < MT; ++j) {
      if (MT * mq + j < 7) {                     // (wave-uniform)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const unsigned w0 = pk_bf16(acc[j][4 * g], acc[j][4 * g + 1]), w1 = pk_bf16(acc[j][4 * g + 2], acc[j][4 * g + 3]);
          *reinterpret_cast<uint2*>(sb + ypix[j] + (((4 * nh + g) ^ yswz[j]) << 4)) = make_uint2(w0, w1);
          if (STATS) {
            const float v0 = bf_lo(w0), v1 = bf_hi(w0), v2 = bf_lo(w1), v3 = bf_hi(w1);
            ssum[4 * g] += v0; ssum[4 * g + 1] += v1; ssum[4 * g + 2] += v2; ssum[4 * g + 3] += v3;
            ssq[4 * g] = fmaf(v0, v0, ssq[4 * g]); ssq[4 * g + 1] = fmaf(v1, v1, ssq[4 * g + 1]);
            ssq[4 * g + 2] = fmaf(v2, v2, ssq[4 * g + 2]); ssq[4 * g + 3] = fmaf(v3, v3, ssq[4 * g + 3]);
          }
        }
      }
    }
#ifdef RP_CONV_PROBE
    const unsigned long long c2 = PROBE_T();
    __syncthreads();
    const unsigned long long c3 = PROBE_T();
    pr_loop += c1 - c0; pr_epi += c2 - c1; pr_bar += c3 - c2; pr_n += 1;
#else
    __syncthreads();
#endif
  };

  if (t0 < t1) {
#pragma unroll
    for (int i = 0; i < VPT; ++i) fetch1(i, t0, true);
#pragma unroll
    for (int i = 0; i < VPT; ++i) stash1(i, t0, 0);
#pragma unroll
    for (int i = 0; i < VPT; ++i) fetch1(i, min(t0 + 1, t1 - 1), t0 + 1 < t1);
  }
  __syncthreads();
  for (int t = t0; t < t1; t += 2) {
    tile(std::integral_constant<int, 0>{}, t);
    if (t + 1 < t1) tile(std::integral_constant<int, 1>{}, t + 1);
  }
  if (t0 < t1) {
#pragma unroll
    for (int k = 0; k < OPT; ++k) writeout1(k, t1 - 1, true, ((t1 - 1 - t0) & 1) * STAGE);
    __syncthreads();                             // (the statistics scratch below aliases the halo buffers, not the stage: order only)
  }

#ifdef RP_CONV_PROBE
  if (p.stats && lane == 0) {       // [block][wave][4] cycle sums (s_memtime) in the statistics buffer
    double* o = p.stats + ((long long)b * 8 + wave) * 4;
    o[0] = (double)pr_loop; o[1] = (double)pr_epi; o[2] = (double)pr_bar; o[3] = (double)pr_n;
  }
  return;
#endif
  if (STATS) {
    // lanes of one half-wave hold 32 pixels' partial sums of the same 16 channels: reduce over l31, then over the two M halves (waves)
    float* red = reinterpret_cast<float*>(lds);                 // [wave][2][32 channels of nh] after the loop's last barrier
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float a = ssum[r], q = ssq[r];
#pragma unroll
      for (int o = 16; o > 0; o >>= 1) { a += __shfl_xor(a, o, 64); q += __shfl_xor(q, o, 64); }
      if (l31 == 0) {
        const int ch = acc_row(r, hi);
        red[(wave * 2 + 0) * 32 + ch] = a;
        red[(wave * 2 + 1) * 32 + ch] = q;
      }
    }
    __syncthreads();
    if (tid < 128) {
      const int which = tid >> 6, ch = tid & 63, n = ch >> 5, c = ch & 31;           // waves (mq, n): wave = 2 mq + n
      double v = 0.0;
#pragma unroll
      for (int q = 0; q < NT / 128; ++q) v += (double)red[((q * 2 + n) * 2 + which) * 32 + c];
      p.stats[((long long)b * 2 + which) * C + ch] = v;
    }
  }
}

}  // namespace

extern "C" int rp_conv3x3_c64_blocks(int N) {
  const int tiles = N * TPI;
  return tiles < 256 ? tiles : 256;
}

/* y = conv3x3(act(x), w), stride 1, pad 1, 64 -> 64 channels, 56 x 56 maps, NHWC bf16; act = identity or max(0, x * scale + shift) */
extern "C" int rp_conv3x3_c64_bf16(const void* x, const void* w, void* y, const float* scale, const float* shift, double* stats, int N,
                                   int H, int W, void* stream) {
  if (!x || !w || !y || N <= 0) return RP_EBADSHAPE;
  if (H != IH || W != IW) return RP_EUNSUPPORTED;
  // the halo loads address the activation through a buffer resource whose out-of-range sentinel (offset 0x80000000) must lie beyond
  // num_records = bytes of x: more than 2 GiB of input (5349 images) would overflow the int product or let the sentinel land inside
  if ((size_t)N * IH * IW * C * 2 > 0x7fffffffull) return RP_EUNSUPPORTED;
  if ((scale == nullptr) != (shift == nullptr)) return RP_EBADSHAPE;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y) & 15) return RP_EALIGN;
  ConvP p{(const bf16_t*)x, (const bf16_t*)w, (bf16_t*)y, scale, shift, stats, N * TPI};
  const dim3 grid(rp_conv3x3_c64_blocks(N)), block(NT);
  hipStream_t st = (hipStream_t)stream;
  if (scale) {
    if (stats) hipLaunchKernelGGL((conv3x3_c64_kernel<true, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv3x3_c64_kernel<true, false>), grid, block, 0, st, p);
  } else {
    if (stats) hipLaunchKernelGGL((conv3x3_c64_kernel<false, true>), grid, block, 0, st, p);
    else hipLaunchKernelGGL((conv3x3_c64_kernel<false, false>), grid, block, 0, st, p);
  }
  RP_CHECK_LAUNCH();
  return RP_OK;
}
