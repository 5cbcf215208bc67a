// mlp_fused.hip -- the transformer MLP of a Block as ONE kernel with the hidden activation kept on chip:
//     y = x + fc2(GELU(fc1(LayerNorm(x)) + b1)) + b2          reference vision_transformer.py:353, vit_layers/mlp.py:20-26
// for the ViT of rel_pose (C = 192, hidden = 768).  Inference path (no activations are saved): x is read once, y written
// once; the normalised rows, the [rows, 768] hidden activation and its pre-activation never exist in HBM (the unfused
// chain moves 3 x 226 MB per Block at 64 pairs for them).
//
// Work decomposition.  A wave owns 16 token rows for the whole kernel; the hidden dimension is streamed in chunks of 32
// units.  Everything is computed TRANSPOSED with v_mfma_f32_16x16x4_f32 (exact fp32; the one fp32 MFMA that sustains its
// datasheet rate at every occupancy on gfx950, profiles/r2_mfma_ceiling.txt):
//     A operand: lane l holds A[i = l&15][k = l>>4]     B operand: lane l holds B[k = l>>4][j = l&15]
//     D        : reg r of lane l is D[i = 4*(l>>4) + r][j = l&15]
//   GEMM1   H^T[unit, row] = sum_k W1[unit, k] * Xn[row, k]      A = W1 rows from LDS, B = the wave's normalised rows, RESIDENT
//                                                                 in 48 VGPRs (lane (j, q) holds Xn[row j][16t + 4q + 0..3])
//   bias + GELU (branch-free erfc form, <= 1 ulp of the exact-erf value: common.h gelu_fast) on the 8 accumulator registers
//   GEMM2   Y^T[col, row] += sum_unit W2[col, unit] * H[unit, row]   A = W2 rows from LDS, B = THE GEMM1 ACCUMULATORS: reg r of
//                                                                 lane (j, q) is H[unit 4q + r][row j] = the B operand of k-step r,
//                                                                 so the hidden activation never leaves the register file.
// The contraction order inside a k-group is free, so both A operands are read as one ds_read_b128 per 4 k-steps.
// Weights: W1 chunk [32 x 192] and W2 chunk [192 x 32] (24 KB each) are staged by LDS-DMA (global_load_lds_dwordx4, lane-linear
// -> unpadded tiles, XOR-swizzled 16-byte chunks so that the 16 rows one read group touches fall into 16 distinct bank groups);
// single-buffered, two barriers per chunk: W2(c) streams in under GEMM1(c), W1(c+1) under GEMM2(c).  (The bf16 form -- one workgroup
// per CU, an eighth of the matrix time per chunk -- stages W1 | W2 (| the chunk's h_pre rows, MODE 1) in a three-stage ring filled two
// chunks ahead, with one barrier per chunk and hand-counted vmcnt waits that leave the chunk's stores in flight: see NS below.)
//
// Balance (stream-K).  The work list is (row tile of 16*NW rows) x (24 chunks), cut into gridDim.x equal contiguous ranges, one per
// resident workgroup: 576 row tiles on 512 workgroup slots would otherwise run as 1 + 1/8 rounds.  A workgroup that owns all 24
// chunks of a tile finishes it (bias, residual, store); otherwise it writes its partial Y tile to the workspace and
// mlp_fixup_kernel adds the (at most few) partials of that tile in chunk order -- fixed order, no atomics, deterministic.
#include <stdlib.h>
#include <algorithm>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
RP_DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
RP_DEV f32x4v mfma16bf(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int C = 192, HID = 768, CH = 32, NCHUNK = HID / CH;
constexpr int W1T = CH * C, W2T = C * CH;                      // floats per staged tile (6144 each = 24 KB)

struct MlpP {
  const float *x, *gamma, *beta, *w1, *b1, *w2, *b2;
  float* y;
  const float* hpre; // MODE 1: pre-activation of fc1 saved by the forward [M, HID]
  float* dhp;        // MODE 1: gradient of that pre-activation (out) [M, HID]
  float* colpart;    // MODE 1: per (row tile, chunk) column sums of dhp over the tile's rows [tiles, HID] (fc1 bias gradient partials)
  float* part;       // [gridDim.x * P][rows per tile][C] partial tiles
  int M;
  float eps;
  int tiles, base, rem, P;
  int io_bf16;       // BF only: bit 1 = dhp (MODE 1) / h, hpre (MODE 0) written as bf16, bit 2 = hpre holds bf16, bit 3 = xn_out written as bf16
  // MODE 0, TRAIN: what the backward needs, written on the way (all [M, .] fp32): the normalised rows and their statistics, the fc1
  // pre-activation (dhp doubles as its pointer) and the hidden activation
  float *xn_out, *mean_out, *rstd_out, *h_out;
  // MODE 1, optional (ln_x != nullptr): the backward of the LayerNorm in front of fc1 rides on the epilogue --
  //     y = rstd (g - mean_c(g) - xhat mean_c(g xhat)) + dy,   g = dxn o gamma,  xhat = (ln_x - mean) rstd       (dxn never reaches HBM)
  // and ln_part [tiles * LN_SUB][3 C] receives the column sums of (dxn o xhat | dxn | dy) over each row block: dgamma, dbeta and the
  // bias gradient of the Linear that produced dy (rp_layernorm_bwd's contract, reference Block.forward :353)
  const float *ln_x, *ln_gamma, *ln_mean, *ln_rstd;
  float* ln_part;
};
constexpr int LN_SUB = 4;      // partial-sum rows per row tile (a finished tile fills row 0 and zeroes the rest; the fix-up fills all)

RP_DEV void item_range(const MlpP& p, int b, int& start, int& count) {
  start = b * p.base + min(b, p.rem);
  count = p.base + (b < p.rem ? 1 : 0);
}

// MODE 0: inference forward (above).  MODE 1: the backward-data chain of the same MLP,
//     dh = dy W2 ; dhp = dh o GELU'(h_pre) (stored: fc1's weight gradient needs it) ; dxn = dhp W1        (dxn -> p.y)
// which is the SAME two chained products with the transposed weights in the two roles (p.w1 = W2^T [768,192], p.w2 = W1^T
// [192,768]), GELU replaced by the multiplication with GELU'(h_pre) read from HBM in accumulator layout, no LayerNorm on the way
// in and no bias / residual on the way out.  Also emits the column sums of dhp per (row tile, chunk) -- the fc1 bias gradient.
//
// BF (the bf16 configuration; 12-wave workgroups): both products on v_mfma_f32_16x16x32_bf16, fp32 accumulate.  The weights arrive as bf16
// copies (12 KB per staged tile): GEMM1 exactly as linear_rows.hip's bf16 form (a lane's k-set of block u is 32 u + 8 q .. + 7; the dy
// rows are rounded to bf16 once per row tile); GEMM2 contracts the chunk's 32 units in ONE MFMA per 16-column block: its B operand is
// pack8(h0, h1) -- k-slot 8 q + e <-> unit 4 q + e (e < 4) / 16 + 4 q + e - 4 -- so the caller stores the second weight with the units
// of every 32-chunk in THAT order and a lane's A operand is one ds_read_b128.  GELU', the column sums and the stream-K partials stay fp32.
// TRAIN (MODE 0 only): the training forward -- same kernel, but xn / mean / rstd (by the workgroup that runs a tile's first chunk), the
// pre-activation and the hidden activation are stored for the backward (the chain then costs the MFMA time of its two products instead
// of a LayerNorm+fc1 launch and an fc2 launch that re-reads h).
// LNB (MODE 1 only): the LayerNorm backward on the epilogue (MlpP::ln_x) -- its own instantiation, so that the plain backward keeps its
// register allocation.
template <int NW, int WPS, int MODE, bool BF = false, bool TRAIN = false, bool LNB = false>
__global__ __launch_bounds__(NW * 64, WPS) void mlp_fused_kernel(MlpP p) {
  static_assert(!LNB || MODE == 1, "the LayerNorm fold belongs to the backward");
  constexpr int NT = NW * 64, ROWS = NW * 16;
  constexpr int TILE_FL = BF ? W1T / 2 : W1T;                   // floats per staged weight tile (bf16 weights: half)
  constexpr int DMA = (TILE_FL / 4) / NT;                       // 16-byte chunks per thread per tile (3 for NW = 8)
  static_assert((TILE_FL / 4) % NT == 0, "tile must be a whole number of DMA rounds");
  static_assert(!BF || NW == 12, "the bf16 form stages its 12 KB weight tiles in one DMA round of 768 threads");
  static_assert(!TRAIN || MODE == 0, "TRAIN is the training form of the forward");
  // fp32: one W1 and one W2 tile, single-buffered (two barriers per chunk; three workgroups per CU cover each other's DMA latency).
  // BF: ONE workgroup per CU and 8x less matrix time per chunk -- a ring of NS stages, each the W1 | W2 tiles of one chunk (24 KB), filled
  // two chunks ahead; one barrier per chunk (a single-buffered tile exposed the L2 -> LDS latency twice per chunk: 3.3 us per chunk for
  // 0.5 us of MFMA work)
  constexpr int NS = BF ? 3 : 1;
  constexpr int STG = (BF && MODE == 1 ? 3 : 2) * TILE_FL;      // floats per stage (BF, MODE 1: + the h_pre tile of the chunk, [192 rows][32 units] bf16)
  __shared__ __attribute__((aligned(16))) float wt[NS * STG];
  float* const w1t = wt;
  float* const w2t = wt + TILE_FL;
  // MODE 0: gamma | beta of the LayerNorm (b1 is read from L2: staging all of it here would cost the third resident workgroup per
  // CU, measured with tools/lab/rows_probe); MODE 1: per-wave column sums of a chunk
  // (BF: + all of b1 for MODE 0 -- LDS is not what limits a 12-wave workgroup; MODE 1: two parities of the per-wave sums)
  __shared__ __attribute__((aligned(16))) float b1s[MODE == 0 ? (BF ? 3 * C + HID : 3 * C) : (BF ? 2 : 1) * NW * CH];
  __shared__ __attribute__((aligned(16))) float lnred[LNB ? NW * 3 * C : 1];      // per-wave column sums of the LayerNorm backward
  if (MODE == 0) {
    for (int i = threadIdx.x; i < C; i += NW * 64) {
      b1s[i] = p.gamma[i];
      b1s[C + i] = p.beta[i];
      b1s[2 * C + i] = p.b2[i];
    }
    if (BF)
      for (int i = threadIdx.x; i < HID; i += NW * 64) b1s[3 * C + i] = p.b1[i];
    __syncthreads();
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, q = lane >> 4;

  // DMA source offsets (bytes) of this thread's LDS positions; LDS position pp (16-byte units) = round * NT + tid
  unsigned off1[DMA], off2[DMA];
#pragma unroll
  for (int r = 0; r < DMA; ++r) {
    const int pp = r * NT + tid;
    if (BF) {
      const int row1 = pp / 24, sl1 = pp % 24, ch1 = (sl1 & ~7) | ((sl1 & 7) ^ ((row1 >> 1) & 7));
      off1[r] = (unsigned)(row1 * C * 2 + ch1 * 16);
      const int row2 = pp >> 2, ch2 = (pp & 3) ^ ((0 - (row2 >> 2)) & 3);      // (key: dx_lnbwd_bf16.hip -- ds_read_b128's real lane groups)
      // (io_bf16 bit 4: the second weight arrives CHUNK-MAJOR, [24][192][32] -- a staged tile is 12 KB contiguous, 128-byte L2 requests
      // instead of 192 x 64-byte row pieces at a 1536-byte stride: the L1's pending-request limit is what bounds this kernel)
      off2[r] = (p.io_bf16 & 16) ? (unsigned)(row2 * CH * 2 + ch2 * 16) : (unsigned)(row2 * HID * 2 + ch2 * 16);
    } else {
      const int row1 = pp / 48, ch1 = (pp % 48) ^ (row1 & 15);
      off1[r] = (unsigned)((row1 * C + ch1 * 4) * 4);
      const int row2 = pp >> 3, ch2 = (pp & 7) ^ ((row2 >> 1) & 7);
      off2[r] = (unsigned)((row2 * HID + ch2 * 4) * 4);
    }
  }
  const unsigned l1 = lds_byte_addr(w1t) + wave * 1024, l2 = lds_byte_addr(w2t) + wave * 1024;
  auto issue_w1 = [&](int c) {
    const float* src = uniform_ptr(p.w1 + (long long)c * CH * (BF ? C / 2 : C));
#pragma unroll
    for (int r = 0; r < DMA; ++r) glds16(src, off1[r], l1 + r * NT * 16);
  };
  auto issue_w2 = [&](int c) {
    const float* src = uniform_ptr(p.w2 + c * (BF ? CH / 2 : CH));
#pragma unroll
    for (int r = 0; r < DMA; ++r) glds16(src, off2[r], l2 + r * NT * 16);
  };

  const bool hp_dma = BF && MODE == 1 && (p.io_bf16 & 4);       // bf16 h_pre rows travel with the weights (one more DMA instruction per wave)
  auto issue_stage = [&](int item) {               // BF: the tiles of work item `item` (row tile item / 24, chunk item % 24) into ring stage item % NS
    const int cc = item % NCHUNK;
    const unsigned so = (unsigned)(item % NS) * (STG * 4);
    glds16(uniform_ptr(p.w1 + (long long)cc * CH * (C / 2)), off1[0], l1 + so);
    glds16(uniform_ptr((p.io_bf16 & 16) ? p.w2 + cc * (C * CH / 2) : p.w2 + cc * (CH / 2)), off2[0], l2 + so);
    if (MODE == 1 && hp_dma) {                      // LDS position tid = (row tid >> 2, 16-byte slot tid & 3), rows past M clamped
      const long long r0 = (long long)(item / NCHUNK) * ROWS;
      const int r = (int)min((long long)(tid >> 2), (long long)p.M - 1 - r0);
      glds16(uniform_ptr(reinterpret_cast<const float*>(reinterpret_cast<const unsigned short*>(p.hpre) + r0 * HID + cc * CH)), (unsigned)(r * HID * 2 + ((tid & 3) ^ ((tid >> 4) & 3)) * 16),
             l2 + so + TILE_FL * 4);
    }
  };
  auto wait_vm = [&](int n) {                       // s_waitcnt vmcnt(n) (+ lgkmcnt(0)) and the workgroup barrier; n is wave-uniform
    switch (n) {
      case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      case 10: asm volatile("s_waitcnt vmcnt(10) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
      default: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); break;
    }
  };

  int it, cnt;
  item_range(p, blockIdx.x, it, cnt);
  const int end = it + cnt;
  if constexpr (BF) {
    for (int i = 0; i < NS - 1; ++i)
      if (cnt > i) issue_stage(it + i);
  } else {
    if (cnt > 0) issue_w1(it % NCHUNK);
  }
  // BF: besides its DMA instructions a wave issues ns stores of the hidden tensors per chunk.  Vector-memory operations retire in issue
  // order (gfx9: loads and stores share vmcnt), so "stage `it` has landed" is vmcnt(K), K = everything issued after DMA(it) = the stores
  // of chunks it - 2 and it - 1 and the DMA instructions of the top of chunk it - 1 (wave 0's column-sum stores of MODE 1 only make its
  // wait stricter).  Exact only when every wave stores in every chunk: M a whole number of tiles (576 tokens = 3 tiles per image) --
  // otherwise, and with fp32 h_pre rows (plain loads in the chunk), every wait is vmcnt(0).
  const int ns_bf = TRAIN ? ((p.io_bf16 & 2) ? 2 : 4) : MODE == 1 ? ((p.io_bf16 & 2) ? 1 : 2) : 0;
  const int kvm = (BF && p.M % ROWS == 0 && (MODE != 1 || hp_dma)) ? (NS - 1) * ns_bf + (NS - 2) * (MODE == 1 ? 3 : 2) : 0;
  int seg = 0;
  while (it < end) {
    const int tile = it / NCHUNK, c0 = it % NCHUNK, c1 = min(NCHUNK, c0 + end - it);
    const int row = tile * ROWS + wave * 16 + j;
    const bool live = row < p.M;
    const float* xrow = p.x + (long long)min(row, p.M - 1) * C;
    auto colof = [&](int t) { return BF ? 32 * (t >> 1) + 8 * q + 4 * (t & 1) : 16 * t + 4 * q; };   // first column of register group t
    // ---- the wave's 16 rows (MODE 0: layer-normalised) straight into the B-operand registers (lane (j, q): columns 16t + 4q + 0..3;
    // BF: columns 32 (t >> 1) + 8 q + 4 (t & 1) + 0..3, i.e. 8 consecutive per 32-wide k block)
    float xn[48];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      const float4 v = ld4(xrow + colof(t));
      xn[4 * t] = v.x; xn[4 * t + 1] = v.y; xn[4 * t + 2] = v.z; xn[4 * t + 3] = v.w;
    }
    if (MODE == 0) {
      float s = 0.f;
#pragma unroll
      for (int t = 0; t < 12; ++t) s += (xn[4 * t] + xn[4 * t + 1]) + (xn[4 * t + 2] + xn[4 * t + 3]);
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mu = s * (1.0f / C);
      float var = 0.f;
#pragma unroll
      for (int i = 0; i < 48; ++i) {
        const float d = xn[i] - mu;
        var = fmaf(d, d, var);          // (explicit: every instantiation must contract the same way -- an ulp here is a bf16 step later)
      }
      var += __shfl_xor(var, 16, 64);
      var += __shfl_xor(var, 32, 64);
      const float rs = 1.0f / sqrtf(var * (1.0f / C) + p.eps);
      const bool owner = TRAIN && c0 == 0 && live;                  // the range holding the tile's first chunk writes xn / stats
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        const float4 g = ld4(b1s + colof(t)), bb = ld4(b1s + (MODE == 0 ? C : 0) + colof(t));
        xn[4 * t] = fmaf((xn[4 * t] - mu) * rs, g.x, bb.x);
        xn[4 * t + 1] = fmaf((xn[4 * t + 1] - mu) * rs, g.y, bb.y);
        xn[4 * t + 2] = fmaf((xn[4 * t + 2] - mu) * rs, g.z, bb.z);
        xn[4 * t + 3] = fmaf((xn[4 * t + 3] - mu) * rs, g.w, bb.w);
        if (TRAIN && owner && !(BF && (p.io_bf16 & 8)))
          st4(p.xn_out + (long long)row * C + colof(t), make_float4(xn[4 * t], xn[4 * t + 1], xn[4 * t + 2], xn[4 * t + 3]));
      }
      if (TRAIN && owner && q == 0) {
        p.mean_out[row] = mu;
        p.rstd_out[row] = rs;
      }
    }
    f32x4v acc[12];
    // BF forward, whole tile: the accumulators START at x + b2 (the row re-read here, in accumulator layout, hits L1 / L2 right behind the
    // LayerNorm's load of it; the epilogue's re-read 24 chunks later came from HBM again, 113 MB per launch, in the phase where every
    // workgroup of the chip waits on the same loads) -- the epilogue is then stores only
    constexpr bool ACCX = BF && MODE == 0 && TRAIN;      // (the inference form is at its register limit: it keeps the epilogue re-read)
    const bool acc_from_x = ACCX && c0 == 0 && c1 == NCHUNK;
#pragma unroll
    for (int ob = 0; ob < 12; ++ob) {
      acc[ob] = f32x4v{0.f, 0.f, 0.f, 0.f};
      if (ACCX && acc_from_x) {
        const float4 xv = ld4(xrow + 16 * ob + 4 * q), bv = ld4(b1s + 2 * C + 16 * ob + 4 * q);
        acc[ob] = f32x4v{xv.x + bv.x, xv.y + bv.y, xv.z + bv.z, xv.w + bv.w};
      }
    }
    bf16x8 xb[BF ? 6 : 1];
    if constexpr (BF) {
#pragma unroll
      for (int u = 0; u < 6; ++u) xb[u] = pack8(xn + 8 * u);
      if (MODE == 0 && TRAIN && (p.io_bf16 & 8) && c0 == 0 && live) {      // xn as the bf16 rows the MFMA consumes: 6 x 16 bytes
        unsigned short* xo = reinterpret_cast<unsigned short*>(p.xn_out) + (long long)row * C + 8 * q;
#pragma unroll
        for (int u = 0; u < 6; ++u) *reinterpret_cast<bf16x8*>(xo + 32 * u) = xb[u];
      }
    }
    const bool dhp_bf = BF && (p.io_bf16 & 2), hpre_bf = BF && (p.io_bf16 & 4);
    const int bfo = (q & 1) ? 16 + 4 * (q - 1) - 4 * q : 0;      // element offset of a lane's 8 consecutive units in a bf16 row (st/ld_bf16x8)

    for (int c = c0; c < c1; ++c, ++it) {
      const long long ho = (long long)min(row, p.M - 1) * HID + c * CH + 4 * q;
      float4 g0p = make_float4(0.f, 0.f, 0.f, 0.f), g1p = g0p;
      const float* w1s = w1t;                       // this chunk's staged tiles
      const float* w2s = w2t;
      if constexpr (BF) {
        // stage `it` landed everywhere (exact count from the third chunk of a tile segment on and while the DMA of it + 1 was issued;
        // everything older is complete after the first chunk's vmcnt(0)); everybody is past chunk it - 1: its stage takes it + 2
        wait_vm((c >= c0 + NS - 1 && it + NS - 2 < end) ? kvm : 0);
        if (MODE == 1) {
          if (!hp_dma) {
            if (hpre_bf) {
              ld_bf16x8(reinterpret_cast<const unsigned short*>(p.hpre) + ho + bfo, g0p, g1p, q);
            } else {
              g0p = ld4(p.hpre + ho);
              g1p = ld4(p.hpre + ho + 16);
            }
          }
          if (c > c0 && tid < CH) {                 // the previous chunk's column sums (all waves passed the barrier after writing them)
            float sum = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) sum += b1s[((c - 1) & 1) * NW * CH + w * CH + tid];
            p.colpart[(long long)tile * HID + ((it - 1) % NCHUNK) * CH + tid] = sum;
          }
        }
        if (MODE == 0) {
          g0p = ld4(b1s + 3 * C + c * CH + 4 * q);
          g1p = ld4(b1s + 3 * C + c * CH + 16 + 4 * q);
        }
        if (it + NS - 1 < end) issue_stage(it + NS - 1);
        w1s = wt + (it % NS) * STG;
        w2s = w1s + TILE_FL;
      } else {
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // W1(c) landed everywhere; W2 tile free
      issue_w2(c);
      if (MODE == 0) {
        g0p = ld4(p.b1 + c * CH + 4 * q);
        g1p = ld4(p.b1 + c * CH + 16 + 4 * q);
      }
      if (MODE == 1) {
        g0p = ld4(p.hpre + ho);
        g1p = ld4(p.hpre + ho + 16);
        asm volatile("" ::: "memory");            // issue here, under GEMM1 -- not where the values are first used
      }
      }
      // ---- GEMM1: two 16-unit blocks, K = 192
      f32x4v h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
      if constexpr (BF) {
        const float* w0 = w1s + j * (C / 2);
        const float* w1r = w0 + 16 * (C / 2);
        const int key = (j >> 1) & 7;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int sl = 4 * u + q, ch = ((sl & ~7) | ((sl & 7) ^ key)) * 4;
          const float4 l = ld4(w0 + ch), m = ld4(w1r + ch);
          h0 = mfma16bf(__builtin_bit_cast(bf16x8, l), xb[u], h0);
          h1 = mfma16bf(__builtin_bit_cast(bf16x8, m), xb[u], h1);
        }
      } else {
      const float* a0p = w1t + j * C;
      const float* a1p = w1t + (16 + j) * C;
      float4 a0 = ld4(a0p + ((q ^ j) * 4)), a1 = ld4(a1p + ((q ^ j) * 4));
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);            // group 0's reads; the loop keeps one group of reads in flight
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        float4 n0 = a0, n1 = a1;
        if (t + 1 < 12) {                                           // next k-group's operands are in flight under these MFMAs
          const int ch = ((4 * (t + 1) + q) ^ j) * 4;
          n0 = ld4(a0p + ch);
          n1 = ld4(a1p + ch);
        }
        h0 = mfma16(a0.x, xn[4 * t], h0);
        h1 = mfma16(a1.x, xn[4 * t], h1);
        h0 = mfma16(a0.y, xn[4 * t + 1], h0);
        h1 = mfma16(a1.y, xn[4 * t + 1], h1);
        h0 = mfma16(a0.z, xn[4 * t + 2], h0);
        h1 = mfma16(a1.z, xn[4 * t + 2], h1);
        h0 = mfma16(a0.w, xn[4 * t + 3], h0);
        h1 = mfma16(a1.w, xn[4 * t + 3], h1);
        a0 = n0;
        a1 = n1;
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);          // the two ds_read_b128 of the NEXT group first ...
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);          // ... then this group's eight MFMAs
      }
      }
      float4 pre0 = make_float4(0.f, 0.f, 0.f, 0.f), pre1 = pre0;
      if (MODE == 0) {
        const float4 ba = g0p, bb = g1p;
        if (TRAIN) {
          pre0 = make_float4(h0[0] + ba.x, h0[1] + ba.y, h0[2] + ba.z, h0[3] + ba.w);
          pre1 = make_float4(h1[0] + bb.x, h1[1] + bb.y, h1[2] + bb.z, h1[3] + bb.w);
        }
        if constexpr (BF) {       // the bf16 configuration's GELU (common.h: gelu_bf; below the storage rounding of h, 40 % fewer VALU)
          h0[0] = gelu_bf(h0[0] + ba.x); h0[1] = gelu_bf(h0[1] + ba.y);
          h0[2] = gelu_bf(h0[2] + ba.z); h0[3] = gelu_bf(h0[3] + ba.w);
          h1[0] = gelu_bf(h1[0] + bb.x); h1[1] = gelu_bf(h1[1] + bb.y);
          h1[2] = gelu_bf(h1[2] + bb.z); h1[3] = gelu_bf(h1[3] + bb.w);
        } else {
        h0[0] = gelu_fast(h0[0] + ba.x); h0[1] = gelu_fast(h0[1] + ba.y);
        h0[2] = gelu_fast(h0[2] + ba.z); h0[3] = gelu_fast(h0[3] + ba.w);
        h1[0] = gelu_fast(h1[0] + bb.x); h1[1] = gelu_fast(h1[1] + bb.y);
        h1[2] = gelu_fast(h1[2] + bb.z); h1[3] = gelu_fast(h1[3] + bb.w);
        }
      } else {
        if (BF && hp_dma) {     // this lane's eight h_pre values from the staged tile: row 16 wave + j, units 4 q .. + 3 and 16 + 4 q .. + 3
          // (16-byte slots of a row XOR-ed with (row >> 2) & 3 by the DMA: rows j, j + 4, j + 8, j + 12 of a half-wave land in different banks)
          const unsigned short* hp = reinterpret_cast<const unsigned short*>(w2s + TILE_FL) + (wave * 16 + j) * CH + 4 * (q & 1);
          const int hk = (j >> 2) & 3;
          g0p = widen4(*reinterpret_cast<const uint2*>(hp + 8 * ((q >> 1) ^ hk)));
          g1p = widen4(*reinterpret_cast<const uint2*>(hp + 8 * ((2 + (q >> 1)) ^ hk)));
        }
        // (opaque redefinition after GEMM1's LDS reads: otherwise hipcc starts GELU' -- and waits for the loads -- at the chunk's top)
        asm volatile("" : "+v"(g0p.x), "+v"(g0p.y), "+v"(g0p.z), "+v"(g0p.w), "+v"(g1p.x), "+v"(g1p.y), "+v"(g1p.z), "+v"(g1p.w)
                     :: "memory");
        if constexpr (BF) {
          h0[0] *= gelu_bf_grad(g0p.x); h0[1] *= gelu_bf_grad(g0p.y);
          h0[2] *= gelu_bf_grad(g0p.z); h0[3] *= gelu_bf_grad(g0p.w);
          h1[0] *= gelu_bf_grad(g1p.x); h1[1] *= gelu_bf_grad(g1p.y);
          h1[2] *= gelu_bf_grad(g1p.z); h1[3] *= gelu_bf_grad(g1p.w);
        } else {
        h0[0] *= gelu_grad_fast(g0p.x); h0[1] *= gelu_grad_fast(g0p.y);
        h0[2] *= gelu_grad_fast(g0p.z); h0[3] *= gelu_grad_fast(g0p.w);
        h1[0] *= gelu_grad_fast(g1p.x); h1[1] *= gelu_grad_fast(g1p.y);
        h1[2] *= gelu_grad_fast(g1p.z); h1[3] *= gelu_grad_fast(g1p.w);
        }
        // column sums of the chunk over the wave's 16 rows (fixed shuffle tree), one row of b1s per wave
        float cs[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          cs[r] = live ? h0[r] : 0.f;
          cs[4 + r] = live ? h1[r] : 0.f;
        }
#pragma unroll
        for (int r = 0; r < 8; ++r) cs[r] = row16_sum(cs[r]);
        if (j == 0) {
          float* cp = b1s + (BF ? (c & 1) * NW * CH : 0) + wave * CH;
          st4(cp + 4 * q, make_float4(cs[0], cs[1], cs[2], cs[3]));
          st4(cp + 16 + 4 * q, make_float4(cs[4], cs[5], cs[6], cs[7]));
        }
      }
      if constexpr (!BF) {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");      // W2(c) landed everywhere; W1 tile free
        if (it + 1 < end) issue_w1(((it + 1) % NCHUNK));
      }
      if (TRAIN && live) {      // (after the DMA issue, like MODE 1's stores)
        // h and h_pre are read by the BACKWARD, long after they have left every cache (590 MB per launch in fp32): non-temporal stores,
        // so that their lines do not displace the weight tiles this kernel keeps re-reading from L2 (round 6, like the stored-P tiles)
        if (dhp_bf) {           // bf16 configuration: the hidden tensors live in bf16
          st_bf16x8<true>(reinterpret_cast<unsigned short*>(p.dhp) + ho + bfo, pre0, pre1, q);
          st_bf16x8<true>(reinterpret_cast<unsigned short*>(p.h_out) + ho + bfo, make_float4(h0[0], h0[1], h0[2], h0[3]),
                          make_float4(h1[0], h1[1], h1[2], h1[3]), q);
        } else {
          st4_nt(p.dhp + ho, pre0);
          st4_nt(p.dhp + ho + 16, pre1);
          st4_nt(p.h_out + ho, make_float4(h0[0], h0[1], h0[2], h0[3]));
          st4_nt(p.h_out + ho + 16, make_float4(h1[0], h1[1], h1[2], h1[3]));
        }
      }
      if (MODE == 1) {          // after the DMA issue: these stores have the whole of GEMM2 to retire before the next vmcnt(0)
        if (live) {
          if (dhp_bf) {
            st_bf16x8(reinterpret_cast<unsigned short*>(p.dhp) + ho + bfo, make_float4(h0[0], h0[1], h0[2], h0[3]),
                      make_float4(h1[0], h1[1], h1[2], h1[3]), q);
          } else {
            st4(p.dhp + ho, make_float4(h0[0], h0[1], h0[2], h0[3]));
            st4(p.dhp + ho + 16, make_float4(h1[0], h1[1], h1[2], h1[3]));
          }
        }
        if (!BF && tid < CH) {
          float sum = 0.f;
#pragma unroll
          for (int w = 0; w < NW; ++w) sum += b1s[w * CH + tid];
          p.colpart[(long long)tile * HID + c * CH + tid] = sum;
        }
      }
      // ---- GEMM2: 12 column blocks of 16, K = the 32 units of this chunk
      if constexpr (BF) {
        const bf16x8 hb = pack8(h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]);
#pragma unroll
        for (int ob = 0; ob < 12; ++ob) {
          const int r2 = 16 * ob + j;
          const float4 a = ld4(w2s + r2 * (CH / 2) + ((q ^ ((0 - (r2 >> 2)) & 3)) * 4));
          acc[ob] = mfma16bf(__builtin_bit_cast(bf16x8, a), hb, acc[ob]);
        }
      } else {
      auto w2frag = [&](int ob, float4& f0, float4& f1) {
        const int r2 = 16 * ob + j, sw = (r2 >> 1) & 7;
        const float* ap = w2t + r2 * CH;
        f0 = ld4(ap + ((q ^ sw) * 4));
        f1 = ld4(ap + (((4 + q) ^ sw) * 4));
      };
      float4 g0, g1;
      w2frag(0, g0, g1);
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int ob = 0; ob < 12; ++ob) {
        float4 n0 = g0, n1 = g1;
        if (ob + 1 < 12) w2frag(ob + 1, n0, n1);
        acc[ob] = mfma16(g0.x, h0[0], acc[ob]);
        acc[ob] = mfma16(g0.y, h0[1], acc[ob]);
        acc[ob] = mfma16(g0.z, h0[2], acc[ob]);
        acc[ob] = mfma16(g0.w, h0[3], acc[ob]);
        acc[ob] = mfma16(g1.x, h1[0], acc[ob]);
        acc[ob] = mfma16(g1.y, h1[1], acc[ob]);
        acc[ob] = mfma16(g1.z, h1[2], acc[ob]);
        acc[ob] = mfma16(g1.w, h1[3], acc[ob]);
        g0 = n0;
        g1 = n1;
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
      }
    }
    if (BF && MODE == 1) {                          // the segment's last chunk's column sums
      __syncthreads();
      if (tid < CH) {
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += b1s[((c1 - 1) & 1) * NW * CH + w * CH + tid];
        p.colpart[(long long)tile * HID + ((it - 1) % NCHUNK) * CH + tid] = sum;
      }
    }
    // ---- epilogue: lane (j, q) holds Y[row j][16 ob + 4q + 0..3]
    if (LNB && c0 == 0 && c1 == NCHUNK) {
      // LayerNorm backward on the accumulators: lane (j, q) holds dxn[row j][16 ob + 4 q + 0..3]; a row's 192 columns are 48 values in
      // each of its four lanes (q).  Two passes over the row of x (the second hits L1 / L2) keep the live registers at the
      // accumulators' 48: the workgroup runs at three waves per SIMD.
      const long long rc = min(row, p.M - 1);
      const float* xr = p.ln_x + rc * C + 4 * q;
      const float* ar = p.x + rc * C + 4 * q;                 // the residual-branch gradient IS this kernel's dy
      const float* gr = p.ln_gamma + 4 * q;
      const float mu = p.ln_mean[rc], rs = p.ln_rstd[rc];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int ob = 0; ob < 12; ++ob) {
        const float4 xv = ld4(xr + 16 * ob), gv = ld4(gr + 16 * ob);
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float g = acc[ob][r] * ga[r];
          s1 += g;
          s2 = fmaf(g, (xa[r] - mu) * rs, s2);
        }
      }
      s1 += __shfl_xor(s1, 16, 64);
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const float m1 = s1 * (1.0f / C), m2 = s2 * (1.0f / C);
      float* lr = lnred + wave * 3 * C + 4 * q;
#pragma unroll
      for (int ob = 0; ob < 12; ++ob) {
        const float4 xv = ld4(xr + 16 * ob), gv = ld4(gr + 16 * ob), av = ld4(ar + 16 * ob);
        const float xa[4] = {xv.x, xv.y, xv.z, xv.w}, ga[4] = {gv.x, gv.y, gv.z, gv.w}, aa[4] = {av.x, av.y, av.z, av.w};
        float o[4], t0[4], t1[4], t2[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float xh = (xa[r] - mu) * rs, d = acc[ob][r];
          o[r] = rs * (d * ga[r] - m1 - xh * m2) + aa[r];
          t0[r] = row16_sum(live ? d * xh : 0.f);
          t1[r] = row16_sum(live ? d : 0.f);
          t2[r] = row16_sum(live ? aa[r] : 0.f);
        }
        if (live) st4(p.y + (long long)row * C + 16 * ob + 4 * q, make_float4(o[0], o[1], o[2], o[3]));
        if (j == 0) {
          st4(lr + 16 * ob, make_float4(t0[0], t0[1], t0[2], t0[3]));
          st4(lr + C + 16 * ob, make_float4(t1[0], t1[1], t1[2], t1[3]));
          st4(lr + 2 * C + 16 * ob, make_float4(t2[0], t2[1], t2[2], t2[3]));
        }
      }
      __syncthreads();
      for (int i = tid; i < 3 * C; i += NT) {                 // waves in order: fixed summation order
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += lnred[w * 3 * C + i];
        float* dst = p.ln_part + (long long)tile * LN_SUB * 3 * C + i;
        dst[0] = sum;
#pragma unroll
        for (int u = 1; u < LN_SUB; ++u) dst[u * 3 * C] = 0.f;
      }
      __syncthreads();                                        // lnred is reused by the next tile
    } else if (c0 == 0 && c1 == NCHUNK) {
      if (live) {
        float* yr = p.y + (long long)row * C + 4 * q;
        // ALL residual loads first, then the stores: y may alias x for the compiler, so "load, add, store" per block came out as twelve
        // serialized round trips (s_waitcnt vmcnt(0) before every store, behind every store already in flight -- vmcnt retires in order)
        float4 rr[12];
#pragma unroll
        for (int ob = 0; ob < 12; ++ob) rr[ob] = (MODE == 0 && !ACCX) ? ld4(xrow + 16 * ob + 4 * q) : make_float4(0.f, 0.f, 0.f, 0.f);
        if (MODE == 0 && !ACCX) asm volatile("" ::: "memory");
#pragma unroll
        for (int ob = 0; ob < 12; ++ob) {
          float4 b2 = make_float4(0.f, 0.f, 0.f, 0.f);
          if (MODE == 0 && !ACCX) b2 = ld4(b1s + 2 * C + 16 * ob + 4 * q);      // (ACCX: already in the accumulators)
          const float4 r = rr[ob];
          const float4 yv = make_float4(acc[ob][0] + b2.x + r.x, acc[ob][1] + b2.y + r.y, acc[ob][2] + b2.z + r.z, acc[ob][3] + b2.w + r.w);
          st4(yr + 16 * ob, yv);
        }
      }
    } else {
      float* pr = p.part + (((long long)blockIdx.x * p.P + seg) * ROWS + wave * 16 + j) * C + 4 * q;
#pragma unroll
      for (int ob = 0; ob < 12; ++ob) st4(pr + 16 * ob, make_float4(acc[ob][0], acc[ob][1], acc[ob][2], acc[ob][3]));
    }
    ++seg;
  }
}

// y[tile] = x + b2 + sum of the tile's partials in chunk order, for the tiles that no single workgroup finished
template <int ROWS, int MODE>
__global__ __launch_bounds__(256) void mlp_fixup_kernel(MlpP p, int G) {
  const int tile = blockIdx.y;
  const int first = tile * NCHUNK, last = first + NCHUNK - 1;
  auto owner = [&](int item) {
    const int big = p.rem * (p.base + 1);
    return item < big ? item / (p.base + 1) : p.rem + (item - big) / p.base;
  };
  const int w0 = owner(first), w1 = owner(last);
  if (w0 == w1) return;                                          // finished by its owner
  const int e = (blockIdx.x * 256 + threadIdx.x) * 4;            // element of the [ROWS, C] tile
  if (e >= ROWS * C) return;
  const long long row = (long long)tile * ROWS + e / C;
  if (row >= p.M) return;
  const int col = e % C;
  float4 xv = make_float4(0.f, 0.f, 0.f, 0.f), bv = xv;
  if (MODE == 0) {
    xv = ld4(p.x + row * C + col);
    bv = ld4(p.b2 + col);
  }
  float4 a = make_float4(0.f, 0.f, 0.f, 0.f);
  for (int w = w0; w <= w1; ++w) {
    int st, cn;
    item_range(p, w, st, cn);
    const int seg = tile - st / NCHUNK;
    const float4 v = ld4(p.part + ((long long)w * p.P + seg) * ROWS * C + e);
    a.x += v.x; a.y += v.y; a.z += v.z; a.w += v.w;
  }
  st4(p.y + row * C + col, make_float4(a.x + bv.x + xv.x, a.y + bv.y + xv.y, a.z + bv.z + xv.z, a.w + bv.w + xv.w));
}

// The same for the backward with the LayerNorm folded in (MlpP::ln_x): block (sub, tile) sums the partials of ROWS / LN_SUB rows of
// a tile that no single workgroup finished and applies the LayerNorm backward to them -- a wave per row, lane = columns lane + 64 j,
// arithmetic of ln_bwd_kernel (rowwise.hip) -- and writes the row block's column sums to ln_part[tile * LN_SUB + sub].
template <int ROWS>
__global__ __launch_bounds__(256) void mlp_fixup_ln_kernel(MlpP p, int G) {
  constexpr int J = C / 64, RB = ROWS / LN_SUB;
  static_assert(ROWS % (4 * LN_SUB) == 0, "row blocks are split over four waves");
  __shared__ float red[3][4][C];
  const int tile = blockIdx.y, sub = blockIdx.x;
  const int first = tile * NCHUNK, last = first + NCHUNK - 1;
  auto owner = [&](int item) {
    const int big = p.rem * (p.base + 1);
    return item < big ? item / (p.base + 1) : p.rem + (item - big) / p.base;
  };
  const int w0 = owner(first), w1 = owner(last);
  if (w0 == w1) return;                                          // finished (and its partial sums written) by its owner
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float g[J], dg[J], db[J], da[J];
#pragma unroll
  for (int j = 0; j < J; ++j) {
    g[j] = p.ln_gamma[lane + 64 * j];
    dg[j] = db[j] = da[j] = 0.f;
  }
  for (int rr = 0; rr < RB / 4; ++rr) {
    const int rt = sub * RB + wave * (RB / 4) + rr;              // row of the tile
    const long long row = (long long)tile * ROWS + rt;
    if (row >= p.M) break;
    float d[J] = {};
    for (int w = w0; w <= w1; ++w) {
      int st, cn;
      item_range(p, w, st, cn);
      const int seg = tile - st / NCHUNK;
      const float* src = p.part + (((long long)w * p.P + seg) * ROWS + rt) * C + lane;
#pragma unroll
      for (int j = 0; j < J; ++j) d[j] += src[64 * j];
    }
    const float mu = p.ln_mean[row], rs = p.ln_rstd[row];
    float xh[J], dxh[J], s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < J; ++j) {
      xh[j] = (p.ln_x[row * C + lane + 64 * j] - mu) * rs;
      dxh[j] = d[j] * g[j];
      s1 += dxh[j];
      s2 += dxh[j] * xh[j];
      dg[j] += d[j] * xh[j];
      db[j] += d[j];
    }
    const float m1 = wave_sum(s1) * (1.0f / C), m2 = wave_sum(s2) * (1.0f / C);
#pragma unroll
    for (int j = 0; j < J; ++j) {
      const float a = p.x[row * C + lane + 64 * j];
      da[j] += a;
      p.y[row * C + lane + 64 * j] = rs * (dxh[j] - m1 - xh[j] * m2) + a;
    }
  }
#pragma unroll
  for (int j = 0; j < J; ++j) {
    red[0][wave][lane + 64 * j] = dg[j];
    red[1][wave][lane + 64 * j] = db[j];
    red[2][wave][lane + 64 * j] = da[j];
  }
  __syncthreads();
  float* dst = p.ln_part + ((long long)tile * LN_SUB + sub) * 3 * C;
  for (int i = threadIdx.x; i < 3 * C; i += 256) {
    const int k = i / C, c = i % C;
    dst[i] = red[k][0][c] + red[k][1][c] + red[k][2][c] + red[k][3][c];
  }
}

// NW waves per workgroup (16 rows each), WPS waves per SIMD the register allocation is held to
template <int NW, int WPS, int MODE, bool BF = false, bool TRAIN = false>
struct Variant {
  static constexpr int ROWS = NW * 16;
  static int grid(int tiles) {
    static int slots = 0;
    if (!slots) {
      int dev = 0, cus = 256, per_cu = 1;
      (void)hipGetDevice(&dev);
      (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
      (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, mlp_fused_kernel<NW, WPS, MODE, BF, TRAIN>, NW * 64, 0);
      slots = cus * (per_cu > 0 ? per_cu : 1);
    }
    const long long items = (long long)tiles * NCHUNK;
    return (int)(items < slots ? items : slots);
  }
  static int partition(MlpP& p) {
    p.tiles = (p.M + ROWS - 1) / ROWS;
    const int G = grid(p.tiles), items = p.tiles * NCHUNK;
    p.base = items / G;
    p.rem = items % G;
    p.P = (p.base + 1 + NCHUNK - 1) / NCHUNK + 1;                 // row tiles one range can touch
    return G;
  }
  static size_t workspace(int M) {
    MlpP p{};
    p.M = M;
    const int G = partition(p);
    return (size_t)G * p.P * ROWS * C * sizeof(float);
  }
  static int launch(MlpP p, hipStream_t st) {
    const int G = partition(p);
    if (MODE == 1 && p.ln_x) hipLaunchKernelGGL((mlp_fused_kernel<NW, WPS, MODE, BF, TRAIN, MODE == 1>), dim3(G), dim3(NW * 64), 0, st, p);
    else hipLaunchKernelGGL((mlp_fused_kernel<NW, WPS, MODE, BF, TRAIN>), dim3(G), dim3(NW * 64), 0, st, p);
    RP_CHECK_LAUNCH();
    if (p.rem != 0 || p.base % NCHUNK != 0) {                     // some tile is shared between workgroups
      if (MODE == 1 && p.ln_x) hipLaunchKernelGGL((mlp_fixup_ln_kernel<ROWS>), dim3(LN_SUB, p.tiles), dim3(256), 0, st, p, G);
      else hipLaunchKernelGGL((mlp_fixup_kernel<ROWS, MODE>), dim3((ROWS * C / 4 + 255) / 256, p.tiles), dim3(256), 0, st, p, G);
      RP_CHECK_LAUNCH();
    }
    return RP_OK;
  }
};

int mlp_variant() {
  static int v = -1;
  if (v < 0) {
    const char* e = getenv("RP_MLP_VARIANT");
    v = e ? atoi(e) : 0;
  }
  return v;
}

}  // namespace

extern "C" size_t rp_mlp_fused_workspace_bytes(int M) {
  if (M <= 0) return 0;
  const size_t bfw = std::max(Variant<12, 3, 0, true, true>::workspace(M), Variant<12, 3, 0, true, false>::workspace(M));   // bf16 forms
  switch (mlp_variant()) {
    // (enough for the inference and the training form: their occupancy, hence their stream-K partition, may differ)
    case 1: return std::max(bfw, std::max(Variant<8, 2, 0>::workspace(M), Variant<8, 2, 0, false, true>::workspace(M)));
    case 2: return std::max(bfw, std::max(Variant<12, 3, 0>::workspace(M), Variant<12, 3, 0, false, true>::workspace(M)));
    default: return std::max(bfw, std::max(Variant<4, 3, 0>::workspace(M), Variant<4, 3, 0, false, true>::workspace(M)));
  }
}

extern "C" int rp_mlp_fused_fwd(const float* x, const float* gamma, const float* beta, const float* w1, const float* b1,
                                const float* w2, const float* b2, float* y, void* workspace, int M, int dim, int hidden, float eps,
                                float* xn_out, float* mean_out, float* rstd_out, float* h_out, float* hpre_out, int precision,
                                int io_bf16, void* stream) {
  if (M <= 0 || dim != C || hidden != HID || !x || !gamma || !beta || !w1 || !b1 || !w2 || !b2 || !y || !workspace)
    return RP_EBADSHAPE;
  const bool train = xn_out || mean_out || rstd_out || h_out || hpre_out;
  if (train && !(xn_out && mean_out && rstd_out && h_out && hpre_out)) return RP_EBADSHAPE;      // the training outputs come as a set
  if ((precision != 0 && precision != 1) || (io_bf16 && (precision != 1 || (!train && (io_bf16 & ~16)) || (io_bf16 & ~26)))) return RP_EUNSUPPORTED;
  MlpP p{x, gamma, beta, w1, b1, w2, b2, y, nullptr, hpre_out, nullptr, (float*)workspace, M, eps, 0, 0, 0, 0, io_bf16,
         xn_out, mean_out, rstd_out, h_out};
  hipStream_t st = (hipStream_t)stream;
  if (precision == 1) return train ? Variant<12, 3, 0, true, true>::launch(p, st) : Variant<12, 3, 0, true, false>::launch(p, st);
  if (train) {
    switch (mlp_variant()) {
      case 1: return Variant<8, 2, 0, false, true>::launch(p, st);
      case 2: return Variant<12, 3, 0, false, true>::launch(p, st);
      default: return Variant<4, 3, 0, false, true>::launch(p, st);
    }
  }
  switch (mlp_variant()) {
    case 1: return Variant<8, 2, 0>::launch(p, st);
    case 2: return Variant<12, 3, 0>::launch(p, st);
    default: return Variant<4, 3, 0>::launch(p, st);
  }
}

// Backward-data of the MLP (training): dhp [M,768] = (dy W2) o GELU'(hpre), dxn [M,192] = dhp W1, colpart [tiles,768] = column
// sums of dhp per row tile (sum them for the fc1 bias gradient).  w2t = W2^T [768,192], w1t = W1^T [192,768] (contiguous).
// Workgroup shape: 12 waves x 3 per SIMD (192-row tiles, one workgroup per CU) at every precision -- the tile-rows / part-rows /
// workspace queries below describe exactly the kernel that runs (the 8 x 2 shape that used to hide behind RP_MLP_BWD_VARIANT lost its
// A/B in round 3 and made the queries disagree with the bf16 launch: retired).  The 4 x 3 shape of the forward spills here (the h_pre
// operands are 8 more live registers).
extern "C" size_t rp_mlp_fused_bwd_workspace_bytes(int M) { return M <= 0 ? 0 : Variant<12, 3, 1>::workspace(M); }
extern "C" int rp_mlp_fused_bwd_tile_rows(void) { return Variant<12, 3, 1>::ROWS; }
extern "C" int rp_mlp_fused_bwd_ln_part_rows(int M) { return M <= 0 ? 0 : (M + rp_mlp_fused_bwd_tile_rows() - 1) / rp_mlp_fused_bwd_tile_rows() * LN_SUB; }
static int mlp_bwd_launch(MlpP p, int precision, hipStream_t st) {
  if (precision == 1) return Variant<12, 3, 1, true>::launch(p, st);      // (same 192-row tiles: same workspace / tile rows)
  return Variant<12, 3, 1>::launch(p, st);
}
extern "C" int rp_mlp_fused_bwd_ln(const float* dy, const float* hpre, const float* w2t, const float* w1t, float* dhp, float* dx,
                                   float* colpart, void* workspace, int M, int dim, int hidden, int precision, int io_bf16,
                                   const float* ln_x, const float* ln_gamma, const float* ln_mean, const float* ln_rstd, float* ln_part,
                                   void* stream) {
  if (M <= 0 || dim != C || hidden != HID || !dy || !hpre || !w2t || !w1t || !dhp || !dx || !colpart || !workspace || !ln_x ||
      !ln_gamma || !ln_mean || !ln_rstd || !ln_part || dx == dy)
    return RP_EBADSHAPE;
  if ((precision != 0 && precision != 1) || (io_bf16 && (precision != 1 || (io_bf16 & ~22)))) return RP_EUNSUPPORTED;
  MlpP p{dy, nullptr, nullptr, w2t, nullptr, w1t, nullptr, dx, hpre, dhp, colpart, (float*)workspace, M, 0.f, 0, 0, 0, 0, io_bf16,
         nullptr, nullptr, nullptr, nullptr, ln_x, ln_gamma, ln_mean, ln_rstd, ln_part};
  return mlp_bwd_launch(p, precision, (hipStream_t)stream);
}
extern "C" int rp_mlp_fused_bwd(const float* dy, const float* hpre, const float* w2t, const float* w1t, float* dhp, float* dxn,
                                float* colpart, void* workspace, int M, int dim, int hidden, int precision, int io_bf16, void* stream) {
  if (M <= 0 || dim != C || hidden != HID || !dy || !hpre || !w2t || !w1t || !dhp || !dxn || !colpart || !workspace)
    return RP_EBADSHAPE;
  if ((precision != 0 && precision != 1) || (io_bf16 && (precision != 1 || (io_bf16 & ~22)))) return RP_EUNSUPPORTED;
  MlpP p{dy, nullptr, nullptr, w2t, nullptr, w1t, nullptr, dxn, hpre, dhp, colpart, (float*)workspace, M, 0.f, 0, 0, 0, 0, io_bf16,
         nullptr, nullptr, nullptr, nullptr};
  if (precision == 1) return Variant<12, 3, 1, true>::launch(p, (hipStream_t)stream);      // (same 192-row tiles: same workspace / tile rows)
  return Variant<12, 3, 1>::launch(p, (hipStream_t)stream);
}
