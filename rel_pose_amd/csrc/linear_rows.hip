// linear_rows.hip -- y = act(LN?(x) W^T + b) (+ residual) for the K = 192 Linear layers of the ViT (qkv 192->576, attention proj
// 192->192, fc1 192->768; vision_transformer.py:323,330,352-353, vit_layers/mlp.py:22-23), "row-resident" formulation:
//
//   * a wave owns 16 token rows; because K = 192 is the WHOLE row, the (optionally LayerNorm-ed) rows live in
//     48 VGPRs for the kernel's lifetime as the B operand of v_mfma_f32_16x16x4_f32 -- x is read from HBM once, LayerNorm costs
//     no pass of its own, and LDS holds only the weights (one ds_read_b128 per 8 MFMAs instead of two operands per step);
//   * the product is formed transposed, Y^T[unit, row] = sum_k W[unit, k] X[row, k], so the accumulators of lane (j, q) are
//     Y[row j][4q + 0..3] of a 16-unit block: bias / GELU / residual apply in registers and the result leaves as 16-byte
//     stores (4 lanes = 64 contiguous bytes per row, the neighbouring block completes the 128-byte line) -- no LDS epilogue;
//   * the weights stream through LDS in chunks of 32 units ([32 x 192] = 24 KB, double buffered, LDS-DMA, XOR-swizzled), one
//     barrier per chunk (96 MFMAs per wave);
//   * work list = (row tile of 64 rows) x (N / 32 chunks), cut into gridDim.x equal contiguous ranges (one per resident
//     workgroup slot; stream-K without a fix-up, because different chunks are different output columns).
// Against the generic LDS-DMA GEMM (csrc/gemm_dma.hip) this removes the A-operand staging, the LDS transpose of the
// epilogue and the separate LayerNorm kernel for exactly the shapes whose C-store-to-flop ratio is worst (K = 192).
#include <stdlib.h>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef float f32x4v __attribute__((ext_vector_type(4)));
RP_DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
// bf16 operand precision (BF = true, the bf16 configuration): v_mfma_f32_16x16x32_bf16.  Lane (j, q) of a 32-wide k block u supplies
// the 8 CONSECUTIVE values k = 32 u + 8 q .. + 7 of its row: the weights arrive as bf16 (the host keeps a bf16 copy, refreshed when
// the fp32 master changes), so a staged chunk is [32 units][192] bf16 = 12 KB (half the DMA and LDS traffic of the fp32 form) and
// one ds_read_b128 IS one A operand -- no conversion in the loop; the x rows are rounded to bf16 (nearest even) once per row tile.
// The accumulator layout, and so the whole epilogue, is that of the fp32 form.  LDS image of a chunk: 24 slots of 16 B per unit
// row (384 B = 1.5 bank rows), slot index XOR-swizzled within its group of 8 by (row >> 1) & 7: the 16 lanes of a read then cover
// 16 distinct 16-byte slots of the 256-byte bank row.
RP_DEV f32x4v mfma16bf(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int C = 192, CH = 32, WT = CH * C;               // 6144 floats = 24 KB per staged weight chunk
constexpr int NW = 4, NT = NW * 64, ROWS = NW * 16, DMA = (WT / 4) / NT;   // 6 LDS-DMA rounds per chunk
constexpr int MAXN = 1024;

struct RowsP {
  const float *x, *w, *bias, *res, *gamma, *beta;
  float *y, *ypre, *xn, *mean, *rstd;
  const float* aux;          // [M,N]: y *= GELU'(aux) (input-gradient form: x = dy, w = W^T, aux = the layer's saved pre-activation)
  float* colpart;            // [tiles,N]: column sums of y over each 64-row tile (the bias gradient of the layer below), or null
  int M, N;
  float eps;
  int act;                   // 0 none, 1 GELU
  int nchunk, tiles, base, rem;
  int io_bf16;               // BF only: bit 0 = x holds bf16 rows (no LayerNorm), bit 1 = y / ypre stored as bf16, bit 2 = aux holds bf16
                             // (RpGemm.io_bf16's meaning), bit 3 = xn_out stored as bf16 (the rounded rows the MFMA consumed)
#ifdef RP_ROWS_PROBE
  long long* probe;          // tools/lab/rows_probe: shader-clock stamps [block][chunk][wave][5]
#endif
};

#ifdef RP_ROWS_PROBE
#define RP_STAMP(k)                                                                                             \
  if (lane == 0 && blockIdx.x < 8 && nstamp < 64)                                                               \
  p.probe[((blockIdx.x * 64 + nstamp) * NW + wave) * 5 + (k)] = __builtin_readcyclecounter()
#else
#define RP_STAMP(k)
#endif

template <bool LN, bool BF = false>
__global__ __launch_bounds__(NT, 3) void linear_rows_kernel(RowsP p) {
  constexpr int WTB = BF ? WT / 2 : WT;             // floats per staged chunk (bf16 weights: half)
  constexpr int NDMA = BF ? DMA / 2 : DMA;
  // bf16 form: FOUR 12 KB stages (the same 48 KB), the weight DMA runs up to three chunks ahead with a counted vmcnt -- its chunk costs
  // 12 MFMAs of 16 cycles instead of 96 of 32, so nothing hides a DMA issued only one chunk ahead
  constexpr int NS = BF ? 4 : 2;
  __shared__ __attribute__((aligned(16))) float wt[NS][WTB];
  // 1.5 KB shared by two uses that never coincide: gamma | beta of the fused LayerNorm (forward), or the per-wave column sums of a
  // chunk by chunk parity (input-gradient form; readers of chunk k never meet writers of k + 1).  Kept this small on purpose: with
  // 53 KB per workgroup only two, not three, workgroups are resident per CU (measured: tools/lab/rows_probe; the occupancy API still reports three).
  __shared__ __attribute__((aligned(16))) float aux_lds[2 * C];
  float (*cs_lds)[NW * CH] = reinterpret_cast<float (*)[NW * CH]>(aux_lds);
  static_assert(2 * NW * CH <= 2 * C, "column-sum scratch must fit the shared 1.5 KB");
  if (LN) {
    for (int i = threadIdx.x; i < C; i += NT) {
      aux_lds[i] = p.gamma[i];
      aux_lds[C + i] = p.beta[i];
    }
    __syncthreads();
  }
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, q = lane >> 4;
  unsigned off[NDMA];
#pragma unroll
  for (int r = 0; r < NDMA; ++r) {
    const int pp = r * NT + tid;
    if (BF) {
      const int row = pp / 24, sl = pp % 24, ch = (sl & ~7) | ((sl & 7) ^ ((row >> 1) & 7));
      off[r] = (unsigned)(row * C * 2 + ch * 16);
    } else {
      const int row = pp / 48, ch = (pp % 48) ^ (row & 15);
      off[r] = (unsigned)((row * C + ch * 4) * 4);
    }
  }
  const unsigned l0 = lds_byte_addr(&wt[0][0]) + wave * 1024;
  auto issue = [&](int c, int buf) {
    const float* src = uniform_ptr(BF ? p.w + (long long)c * CH * (C / 2) : p.w + (long long)c * CH * C);   // (bf16 weights: C/2 floats per row)
#pragma unroll
    for (int r = 0; r < NDMA; ++r) glds16(src, off[r], l0 + buf * (WTB * 4) + r * NT * 16);
  };
  // column of the first of the 4 consecutive row elements register group t (0..11) of lane (j, q) holds
  auto colof = [&](int t) { return BF ? 32 * (t >> 1) + 8 * q + 4 * (t & 1) : 16 * t + 4 * q; };
  const int b = blockIdx.x;
  int it = b * p.base + min(b, p.rem);
  const int end = it + p.base + (b < p.rem ? 1 : 0);
  int buf = 0;
#ifdef RP_ROWS_PROBE
  int nstamp = 0;
  if (lane == 0 && blockIdx.x < 8) {
    p.probe[((blockIdx.x * 64 + 63) * NW + wave) * 5 + 0] = __builtin_readcyclecounter();
    p.probe[((blockIdx.x * 64 + 63) * NW + wave) * 5 + 3] = wall_clock64();
  }
#endif
  if (it < end) issue(it % p.nchunk, 0);
  if constexpr (BF) {
#pragma unroll
    for (int s = 1; s < NS - 1; ++s)
      if (it + s < end) issue((it + s) % p.nchunk, s);
  }
  int par = 0;               // parity of the column-sum scratch
  const bool has_bias = p.bias != nullptr, pre = p.ypre != nullptr, has_res = p.res != nullptr, has_aux = p.aux != nullptr, want_cs = p.colpart != nullptr;
  while (it < end) {
    const int tile = it / p.nchunk, c0 = it % p.nchunk, c1 = min(p.nchunk, c0 + end - it);
    const int row = tile * ROWS + wave * 16 + j;
    const bool live = row < p.M;
    const long long rclamp = min(row, p.M - 1);
    const float* xr = p.x + rclamp * C;
    float xn[48];
    bf16x8 xb[BF ? 6 : 1];
    bool have_xb = false;
    if constexpr (BF && !LN) {
      if (p.io_bf16 & 1) {       // bf16 rows: lane (j, q) takes its six 8-element MFMA operands straight from memory, no conversion
        const unsigned short* xr16 = reinterpret_cast<const unsigned short*>(p.x) + rclamp * C + 8 * q;
#pragma unroll
        for (int u = 0; u < 6; ++u) xb[u] = *reinterpret_cast<const bf16x8*>(xr16 + 32 * u);
        have_xb = true;
      }
    }
    if (!have_xb) {
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      const float4 v = ld4(xr + colof(t));
      xn[4 * t] = v.x; xn[4 * t + 1] = v.y; xn[4 * t + 2] = v.z; xn[4 * t + 3] = v.w;
    }
    }
    const bool xn_bf = BF && (p.io_bf16 & 8);
    if (LN) {
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
        var += d * d;
      }
      var += __shfl_xor(var, 16, 64);
      var += __shfl_xor(var, 32, 64);
      const float rs = 1.0f / sqrtf(var * (1.0f / C) + p.eps);
      const bool owner = c0 == 0 && live;                            // the range holding the tile's first chunk writes xn / stats
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        const float4 g = ld4(aux_lds + colof(t)), bb = ld4(aux_lds + C + colof(t));
        xn[4 * t] = (xn[4 * t] - mu) * rs * g.x + bb.x;
        xn[4 * t + 1] = (xn[4 * t + 1] - mu) * rs * g.y + bb.y;
        xn[4 * t + 2] = (xn[4 * t + 2] - mu) * rs * g.z + bb.z;
        xn[4 * t + 3] = (xn[4 * t + 3] - mu) * rs * g.w + bb.w;
        if (p.xn && owner && !xn_bf) st4(p.xn + (long long)row * C + colof(t), make_float4(xn[4 * t], xn[4 * t + 1], xn[4 * t + 2], xn[4 * t + 3]));
      }
      if (owner && q == 0) {
        if (p.mean) p.mean[row] = mu;
        if (p.rstd) p.rstd[row] = rs;
      }
    }
    if constexpr (BF) {
      if (!have_xb) {
#pragma unroll
        for (int u = 0; u < 6; ++u) xb[u] = pack8(xn + 8 * u);
      }
      if (LN && xn_bf && p.xn && c0 == 0 && live) {      // the normalised rows as the MFMA consumed them: 6 x 16 bytes instead of 12
        unsigned short* xo = reinterpret_cast<unsigned short*>(p.xn) + (long long)row * C + 8 * q;
#pragma unroll
        for (int u = 0; u < 6; ++u) *reinterpret_cast<bf16x8*>(xo + 32 * u) = xb[u];
      }
    }
    const bool y_bf = BF && (p.io_bf16 & 2), aux_bf = BF && (p.io_bf16 & 4);
    // The stores of chunk c are issued after the barrier of chunk c + 1, BEFORE the next weight DMA: loads and stores retire in
    // order on one counter, so the barrier's vmcnt(0) then waits for a DMA issued a whole chunk ago and for stores older still.
    float4 pv0, pv1, pp0, pp1;
    long long po = 0;
    bool pending = false;
    int cs_tile = 0, cs_chunk = 0, cs_par = 0;
    auto flush = [&]() {
      if (pending && want_cs && tid < CH) {          // (callers sit behind a barrier that follows the cs_lds writes)
        float sum = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) sum += cs_lds[cs_par][w * CH + tid];
        p.colpart[(long long)cs_tile * p.N + cs_chunk * CH + tid] = sum;
      }
      if (pending && live) {
        if (y_bf) {
          // bf16 rows: a lane's 4 + 4 units would be two 8-byte stores; lanes q and q ^ 1 swap halves instead so that every lane owns 8
          // CONSECUTIVE units (even q: units 4q .. 4q+7 of the first 16, odd q: 16 + 4(q-1) .. + 7) = one 16-byte store, 64 contiguous
          // bytes per row and instruction like the fp32 form (the memory system is bound by requests here, not bytes)
          const long long pb16 = po - 4 * q + ((q & 1) ? 16 + 4 * (q - 1) : 4 * q);
          if (pre) st_bf16x8(reinterpret_cast<unsigned short*>(p.ypre) + pb16, pp0, pp1, q);
          st_bf16x8(reinterpret_cast<unsigned short*>(p.y) + pb16, pv0, pv1, q);
        } else {
          if (pre) {
            st4(p.ypre + po, pp0);
            st4(p.ypre + po + 16, pp1);
          }
          st4(p.y + po, pv0);
          st4(p.y + po + 16, pv1);
        }
      }
      pending = false;
    };
    for (int c = c0; c < c1; ++c, ++it) {
      RP_STAMP(0);
      if constexpr (BF) {
        // chunk `it` must have landed; the DMAs of the (up to NS - 2) chunks after it are the youngest VM operations of this wave
        // (order per iteration: epilogue-operand loads, stores of the previous chunk, DMA) and may stay in flight
        const int younger = min(NS - 2, end - 1 - it);
        if (younger >= 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(2 * NDMA) : "memory");
        else if (younger == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(NDMA) : "memory");
        else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      }
      RP_STAMP(1);
      asm volatile("s_barrier" ::: "memory");                                       // W(c) landed; the stage refilled below is free
      RP_STAMP(2);
      if constexpr (!BF) {
        flush();
        if (it + 1 < end) issue((it + 1) % p.nchunk, buf ^ 1);
      }
      const long long o = rclamp * p.N + c * CH + 4 * q;
      float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, g0 = r0, g1 = r0, ba = r0, bb = r0;
      if (has_bias) {          // (from L2; staging the bias in LDS costs the third resident workgroup per CU)
        ba = ld4(p.bias + c * CH + 4 * q);
        bb = ld4(p.bias + c * CH + 16 + 4 * q);
      }
      if (has_res) {
        r0 = ld4(p.res + o);
        r1 = ld4(p.res + o + 16);
      }
      if (has_aux) {
        if (aux_bf) {
          ld_bf16x8(reinterpret_cast<const unsigned short*>(p.aux) + o - 4 * q + ((q & 1) ? 16 + 4 * (q - 1) : 4 * q), g0, g1, q);
        } else {
          g0 = ld4(p.aux + o);
          g1 = ld4(p.aux + o + 16);
        }
        asm volatile("" ::: "memory");
      }
      if constexpr (BF) {
        flush();
        if (it + NS - 1 < end) issue((it + NS - 1) % p.nchunk, buf == 0 ? NS - 1 : buf - 1);
      }
      f32x4v h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
      const float* a0p = &wt[buf][0] + j * C;
      const float* a1p = a0p + 16 * C;
      if constexpr (BF) {
        const float* w0 = &wt[buf][0] + j * (C / 2);            // unit row j of the chunk, in floats (2 bf16 each)
        const float* w1 = w0 + 16 * (C / 2);
        const int key = (j >> 1) & 7;
#pragma unroll
        for (int u = 0; u < 6; ++u) {
          const int sl = 4 * u + q, ch = ((sl & ~7) | ((sl & 7) ^ key)) * 4;
          const float4 l = ld4(w0 + ch), m = ld4(w1 + ch);
          h0 = mfma16bf(__builtin_bit_cast(bf16x8, l), xb[u], h0);
          h1 = mfma16bf(__builtin_bit_cast(bf16x8, m), xb[u], h1);
        }
      } else {
      float4 a0 = ld4(a0p + ((q ^ j) * 4)), a1 = ld4(a1p + ((q ^ j) * 4));
      __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
#pragma unroll
      for (int t = 0; t < 12; ++t) {
        float4 n0 = a0, n1 = a1;
        if (t + 1 < 12) {
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
        __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
        __builtin_amdgcn_sched_group_barrier(0x008, 8, 0);
      }
      }
      RP_STAMP(3);
      float4 v0 = make_float4(h0[0] + ba.x, h0[1] + ba.y, h0[2] + ba.z, h0[3] + ba.w);
      float4 v1 = make_float4(h1[0] + bb.x, h1[1] + bb.y, h1[2] + bb.z, h1[3] + bb.w);
      pp0 = v0;
      pp1 = v1;
      if (has_aux) {
        asm volatile("" : "+v"(g0.x), "+v"(g0.y), "+v"(g0.z), "+v"(g0.w), "+v"(g1.x), "+v"(g1.y), "+v"(g1.z), "+v"(g1.w) :: "memory");
        if constexpr (BF) {       // (the bf16 configuration's GELU pair, common.h: gelu_bf / gelu_bf_grad -- the same one mlp_fused.hip uses)
          v0 = make_float4(v0.x * gelu_bf_grad(g0.x), v0.y * gelu_bf_grad(g0.y), v0.z * gelu_bf_grad(g0.z), v0.w * gelu_bf_grad(g0.w));
          v1 = make_float4(v1.x * gelu_bf_grad(g1.x), v1.y * gelu_bf_grad(g1.y), v1.z * gelu_bf_grad(g1.z), v1.w * gelu_bf_grad(g1.w));
        } else {
        v0 = make_float4(v0.x * gelu_grad_fast(g0.x), v0.y * gelu_grad_fast(g0.y), v0.z * gelu_grad_fast(g0.z), v0.w * gelu_grad_fast(g0.w));
        v1 = make_float4(v1.x * gelu_grad_fast(g1.x), v1.y * gelu_grad_fast(g1.y), v1.z * gelu_grad_fast(g1.z), v1.w * gelu_grad_fast(g1.w));
        }
      }
      if (p.act == 1) {
        if constexpr (BF) {
          v0 = make_float4(gelu_bf(v0.x), gelu_bf(v0.y), gelu_bf(v0.z), gelu_bf(v0.w));
          v1 = make_float4(gelu_bf(v1.x), gelu_bf(v1.y), gelu_bf(v1.z), gelu_bf(v1.w));
        } else {
        v0 = make_float4(gelu_fast(v0.x), gelu_fast(v0.y), gelu_fast(v0.z), gelu_fast(v0.w));
        v1 = make_float4(gelu_fast(v1.x), gelu_fast(v1.y), gelu_fast(v1.z), gelu_fast(v1.w));
        }
      }
      pv0 = make_float4(v0.x + r0.x, v0.y + r0.y, v0.z + r0.z, v0.w + r0.w);
      pv1 = make_float4(v1.x + r1.x, v1.y + r1.y, v1.z + r1.z, v1.w + r1.w);
      if (want_cs) {     // column sums of this chunk over the wave's rows (fixed DPP tree), combined across waves after the barrier
        const float m = live ? 1.f : 0.f;
        float4 c0v = make_float4(row16_sum(m * pv0.x), row16_sum(m * pv0.y), row16_sum(m * pv0.z), row16_sum(m * pv0.w));
        float4 c1v = make_float4(row16_sum(m * pv1.x), row16_sum(m * pv1.y), row16_sum(m * pv1.z), row16_sum(m * pv1.w));
        if (j == 0) {
          st4(&cs_lds[par][0] + wave * CH + 4 * q, c0v);
          st4(&cs_lds[par][0] + wave * CH + 16 + 4 * q, c1v);
        }
        cs_par = par;
        cs_tile = tile;
        cs_chunk = c;
      }
      po = o;
      pending = true;
      buf = buf + 1 == NS ? 0 : buf + 1;
      par ^= 1;
      RP_STAMP(4);
#ifdef RP_ROWS_PROBE
      ++nstamp;
#endif
    }
    if (want_cs) __syncthreads();
    flush();
    if (want_cs) __syncthreads();                  // cs_lds is rewritten by the next segment's first chunk
  }
#ifdef RP_ROWS_PROBE
  if (lane == 0 && blockIdx.x < 8) {
    p.probe[((blockIdx.x * 64 + 63) * NW + wave) * 5 + 1] = __builtin_readcyclecounter();
    p.probe[((blockIdx.x * 64 + 63) * NW + wave) * 5 + 2] = wall_clock64();
  }
#endif
}

template <bool LN, bool BF>
int rows_slots() {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, linear_rows_kernel<LN, BF>, NT, 0);
    slots = cus * (per_cu > 0 ? per_cu : 1);
  }
  return slots;
}

}  // namespace

extern "C" int rp_linear_rows192_tile_rows(void) { return ROWS; }

extern "C" int rp_linear_rows192(const float* x, const float* w, const float* bias, const float* residual, const float* ln_gamma,
                                 const float* ln_beta, float eps, float* y, float* y_pre, float* xn_out, float* mean_out,
                                 float* rstd_out, const float* dact_aux, float* colsum_part, int M, int N, int K, int act,
                                 int precision, int io_bf16, void* stream) {
  if (M <= 0 || K != C || N <= 0 || N % CH != 0 || N > MAXN || !x || !w || !y || act < 0 || act > 1) return RP_EBADSHAPE;
  if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return RP_EBADSHAPE;
  const bool ln = ln_gamma != nullptr;
  if (!ln && (xn_out || mean_out || rstd_out)) return RP_EBADSHAPE;
  if (precision != 0 && precision != 1) return RP_EUNSUPPORTED;
  if (io_bf16 && (precision != 1 || (io_bf16 & ~15) || ((io_bf16 & 4) && !dact_aux) || ((io_bf16 & 1) && ln) || ((io_bf16 & 8) && !xn_out)))
    return RP_EUNSUPPORTED;
  const bool bf = precision == 1;
  RowsP p{x, w, bias, residual, ln_gamma, ln_beta, y, y_pre, xn_out, mean_out, rstd_out, dact_aux, colsum_part, M, N, eps, act, N / CH, (M + ROWS - 1) / ROWS, 0, 0, io_bf16};
  const long long items = (long long)p.tiles * p.nchunk;
  const int slots = bf ? (ln ? rows_slots<true, true>() : rows_slots<false, true>()) : (ln ? rows_slots<true, false>() : rows_slots<false, false>());
  const int G = (int)(items < slots ? items : slots);
  p.base = (int)(items / G);
  p.rem = (int)(items % G);
  hipStream_t st = (hipStream_t)stream;
  if (bf) {
    if (ln) hipLaunchKernelGGL((linear_rows_kernel<true, true>), dim3(G), dim3(NT), 0, st, p);
    else hipLaunchKernelGGL((linear_rows_kernel<false, true>), dim3(G), dim3(NT), 0, st, p);
  } else if (ln) hipLaunchKernelGGL((linear_rows_kernel<true, false>), dim3(G), dim3(NT), 0, st, p);
  else hipLaunchKernelGGL((linear_rows_kernel<false, false>), dim3(G), dim3(NT), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
