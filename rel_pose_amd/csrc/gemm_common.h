// gemm_common.h -- pieces shared by the two rp_gemm kernels (gemm.hip: register-staged main loop, all operand precisions;
// gemm_dma.hip: LDS-DMA-staged fp32 main loop): the launch parameters, the epilogue modes and the tile epilogue.
#pragma once
#include "common.h"

namespace rpgemm {

enum { EPI_RAW = 0, EPI_BIAS, EPI_BIAS_RES, EPI_RES, EPI_BIAS_GELU, EPI_BIAS_GELU_PRE, EPI_BIAS_RELU, EPI_DGELU,
       EPI_DRELU, EPI_GENERIC, EPI_LNBWD };

struct GemmP {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  long long sa, sb, sc;
  int split_k;
  int k_per_split;
  const float* bias;
  float* pre_out;
  int act, dact;
  const float* aux;
  const float* residual;
  int epi_mode;
  int trans_c;
  int limbs;
  float* colsum_part;
  // EPI_LNBWD (RpGemm.ln_x != NULL): the product is the gradient of a LayerNorm OUTPUT; the epilogue applies the LayerNorm
  // backward and stores the gradient of its INPUT (+ residual) -- see lnbwd_epilogue
  const float* ln_x; const float* ln_mean; const float* ln_rstd; const float* ln_gamma;
  float* ln_part;
  // bf16 STORAGE of activation-sized operands (precision 1 = the bf16 configuration only; RpGemm.io_bf16): bit 0 = A holds bf16
  // (2 bytes per element, lda in elements), bit 1 = C and pre_out are written as bf16 (round to nearest even), bit 2 = aux holds bf16.
  // The operands are rounded to bf16 for the MFMA in this precision anyway, so reading them as bf16 changes no arithmetic when the
  // producer stored bf16 -- it halves the bytes of the HBM-bound Linear launches.
  int io_bf16;
};

RP_DEV float4 widen_bf16x4(uint2 w) {
  return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                     __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
}
// 4 consecutive elements at element offset `off` of a tensor that is fp32 or (bf = true) bf16 in memory
RP_DEV float4 ld4_io(const float* base, long long off, bool bf) {
  if (bf) return widen_bf16x4(*reinterpret_cast<const uint2*>(reinterpret_cast<const unsigned short*>(base) + off));
  return ld4(base + off);
}
RP_DEV void st4_io(float* base, long long off, float4 v, bool bf) {
  if (bf) *reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(base) + off) = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w));
  else st4(base + off, v);
}

// The activation pair of an epilogue follows the operand precision: precision 1 (the bf16 configuration) uses the same sigmoid-form
// GELU and its exact derivative as the BF instantiations of mlp_fused.hip / linear_rows.hip (common.h: gelu_bf / gelu_bf_grad), so the
// fallback paths (RP_ROWS_LINEAR=0, RP_ROWS_DX=0, RP_MLP_FUSED_*=0) compute the SAME function forward and backward as the default
// ones; every other precision keeps the reference's erf GELU.
RP_DEV float gelu_sel(float v, bool bf) { return bf ? gelu_bf(v) : gelu_exact(v); }
RP_DEV float gelu_grad_sel(float v, bool bf) { return bf ? gelu_bf_grad(v) : gelu_grad(v); }

RP_DEV float epilogue(float v, int m, int n, const GemmP& p) {
  if (p.bias) v += p.bias[n];
  const long long off = (long long)m * p.ldc + n;
  if (p.pre_out) p.pre_out[off] = v;
  if (p.act == 1) v = gelu_sel(v, p.limbs == 1);
  else if (p.act == 2) v = fmaxf(v, 0.f);
  if (p.dact == 1) v *= gelu_grad_sel(p.aux[off], p.limbs == 1);
  else if (p.dact == 2) v = p.aux[off] > 0.f ? v : 0.f;
  if (p.residual) v += p.residual[off];
  return v;
}


// LDS-staged epilogue of one mode.  The accumulators (lane = column, register = row) are transposed through the wave's own LDS
// region so that every global access -- C, bias, residual, GELU'/ReLU' aux, pre-activation copy -- is a 16-byte row segment
// per lane (global_load/store_dwordx4) instead of 16*TM*TN dword accesses per lane (the dword store tail alone was 11 % of the
// kernel).  Straight-line per mode: the bias segment and ALL [M,N] operand segments (residual / aux) of the lane's rows are
// requested up front -- before the accumulators go to LDS -- and the stores then issue back to back.  (The first version
// tested the mode per element group and waited vmcnt(0) after each operand load, i.e. also for the previous store: one
// store -> load -> store round trip per row group, 8-12 per tile.)  Out-of-range rows / columns load from clamped addresses
// and are masked at the store only, so edge tiles run the same code.
template <int TM, int TN, int MODE, bool BFIO = false>
RP_DEV void staged_epilogue(const GemmP& p, f32x16 (&acc)[TM][TN], float* lds, int m0, int n0, int mt, float* C, int ldc,
                            const float* bias, float* pre_out, const float* aux, const float* res, bool partial) {
  const bool c_bf = BFIO && !partial && (p.io_bf16 & 2), aux_bf = BFIO && (p.io_bf16 & 4);
  constexpr int CST = 32 * TN + 4;
  constexpr int C4 = 8 * TN;             // float4 per staged row
  constexpr int NIT = (32 * TM * C4) / 64;
  constexpr bool HAS_BIAS = MODE == EPI_BIAS || MODE == EPI_BIAS_RES || MODE == EPI_BIAS_GELU || MODE == EPI_BIAS_GELU_PRE ||
                            MODE == EPI_BIAS_RELU;
  constexpr bool HAS_RES = MODE == EPI_BIAS_RES || MODE == EPI_RES;
  constexpr bool HAS_AUX = MODE == EPI_DGELU || MODE == EPI_DRELU;
  constexpr bool LANE_COL = (64 % C4) == 0;          // a lane keeps its column group through the loop
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  const float* mn = HAS_RES ? res : aux;
  float4 b4[LANE_COL ? 1 : NIT], o4[(HAS_RES || HAS_AUX) ? NIT : 1];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = lane + 64 * it;
    const int row = idx / C4, c4 = idx % C4;
    const int mc = min(m0 + wm0 + row, p.M - 1), nc = min(n0 + wn0 + 4 * c4, p.N - 4);
    if (HAS_BIAS && (it == 0 || !LANE_COL)) b4[LANE_COL ? 0 : it] = ld4(bias + nc);
    if (HAS_RES || HAS_AUX) o4[it] = ld4_io(mn, (long long)mc * ldc + nc, HAS_AUX && aux_bf);
  }
  float* cs = lds + wave * (32 * TM * CST);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[(32 * i + acc_row(r, hi)) * CST + 32 * j + l31] = acc[i][j][r];
  __syncthreads();
  float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);   // this lane's column sums of the final values (colsum_part != nullptr)
  const bool want_cs = LANE_COL && p.colsum_part != nullptr && !partial;
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = lane + 64 * it;
    const int row = idx / C4, c4 = idx % C4;
    const int m = m0 + wm0 + row, n = n0 + wn0 + 4 * c4;
    const bool ok = m < p.M && n < p.N;
    float4 v = ld4(cs + row * CST + 4 * c4);
    const long long off = (long long)m * ldc + n;
    if (HAS_BIAS) {
      const float4 b = b4[LANE_COL ? 0 : it];
      v.x += b.x; v.y += b.y; v.z += b.z; v.w += b.w;
    }
    if (MODE == EPI_BIAS_GELU_PRE && ok) st4_io(pre_out, off, v, c_bf);
    if (MODE == EPI_BIAS_GELU || MODE == EPI_BIAS_GELU_PRE) {
      const bool gb = p.limbs == 1;
      v.x = gelu_sel(v.x, gb); v.y = gelu_sel(v.y, gb); v.z = gelu_sel(v.z, gb); v.w = gelu_sel(v.w, gb);
    } else if (MODE == EPI_BIAS_RELU) {
      v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
    } else if (MODE == EPI_DGELU) {
      const float4 a4 = o4[it];
      const bool gb = p.limbs == 1;
      v.x *= gelu_grad_sel(a4.x, gb); v.y *= gelu_grad_sel(a4.y, gb); v.z *= gelu_grad_sel(a4.z, gb); v.w *= gelu_grad_sel(a4.w, gb);
    } else if (MODE == EPI_DRELU) {
      const float4 a4 = o4[it];
      v.x = a4.x > 0.f ? v.x : 0.f; v.y = a4.y > 0.f ? v.y : 0.f; v.z = a4.z > 0.f ? v.z : 0.f; v.w = a4.w > 0.f ? v.w : 0.f;
    }
    if (HAS_RES) {
      const float4 r4 = o4[it];
      v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
    }
    if (ok) {
      st4_io(C, off, v, c_bf);
      if (want_cs) { csum.x += v.x; csum.y += v.y; csum.z += v.z; csum.w += v.w; }
    }
  }
  if (want_cs) {
    // a lane keeps the same column group c4 = lane % C4 through the loop: fold the 64 / C4 lanes of each group, then lanes
    // 0..C4-1 store one row of per-(tile, wave-row) column sums; a small rp_colsum over them finishes the bias gradient
#pragma unroll
    for (int o = 32; o >= C4; o >>= 1) {
      csum.x += __shfl_xor(csum.x, o, 64); csum.y += __shfl_xor(csum.y, o, 64);
      csum.z += __shfl_xor(csum.z, o, 64); csum.w += __shfl_xor(csum.w, o, 64);
    }
    const int n = n0 + wn0 + 4 * lane;
    if (lane < C4 && n < p.N) st4(p.colsum_part + (long long)(2 * mt + (wave >> 1)) * p.N + n, csum);
  }
}

// LayerNorm backward as a GEMM epilogue (tile 64 x 192 = whole rows of the 192-wide model dimension, N == 192).
// The dX GEMM of the Linear that follows a LayerNorm (qkv, fc1) produces dY_n = d loss / d LN(x); reference autograd then
// runs the LayerNorm backward as its own pass: read dY_n, x and the residual-branch gradient, write dx (226 MB per call at
// 64 pairs).  Fused here, dY_n never leaves the chip: the accumulators are staged in LDS, then every wave walks 16 rows with
// lane = column (3 columns per lane), exactly like ln_bwd_kernel in rowwise.hip -- row sums by wave reduction, dx = rstd *
// (dY gamma - mean(dY gamma) - xhat * mean(dY gamma xhat)) (+ add), per-lane column sums of dY xhat (dgamma), dY (dbeta) and
// add (the bias gradient of the Linear that produced the residual branch), combined over the four waves in a fixed order
// and written as one partial row [np * 192] per tile; a column sum over the tiles finishes them (deterministic).
RP_DEV void lnbwd_epilogue(const GemmP& p, f32x16 (&acc)[1][3], float* lds, int m0, int mt, int zb) {
  constexpr int CST = 100, C = 192;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const float* add = p.residual;
  const int rbase = m0 + 16 * wave;
  const int nrows = max(0, min(16, p.M - rbase));            // valid rows of this wave (ragged last tile)
  const int mrow = min(rbase + (lane & 15), p.M - 1);
  const float mu_l = p.ln_mean[mrow], rs_l = p.ln_rstd[mrow];   // row statistics: first 16 lanes, broadcast by shuffle
  float g[3], dg[3] = {0.f, 0.f, 0.f}, db[3] = {0.f, 0.f, 0.f}, da[3] = {0.f, 0.f, 0.f};
  int coff[3];                          // column lane + 64 j inside the staged tile: (column half) * 3200 + column % 96
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int c = lane + 64 * j;
    g[j] = p.ln_gamma[c];
    coff[j] = c < 96 ? c : 32 * CST + (c - 96);
  }
  // The wave's 16 rows go in two batches of 8: all global operands of a batch (LayerNorm input, residual-branch gradient:
  // 48 dwords per lane) are requested together, then the batch is reduced and stored -- one exposed memory latency per batch
  // instead of one per row, at a register cost that keeps two waves per SIMD (all 16 rows at once: 260+ registers).
  // Rows past M re-read the last valid row (no stride) and are masked.
  float xv[8][3], av[8][3];
  auto fetch = [&](int b) {
    const int r0 = 8 * b;
    const float* xrow = p.ln_x + (long long)min(rbase + r0, p.M - 1) * C + lane;
    const float* arow = add ? add + (long long)min(rbase + r0, p.M - 1) * C + lane : nullptr;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        xv[q][j] = xrow[64 * j];
        av[q][j] = add ? arow[64 * j] : 0.f;
      }
      if (r0 + q + 1 < nrows) { xrow += C; if (add) arow += C; }
    }
  };
  auto reduce = [&](int b) {
    float* crow = p.C + (long long)min(rbase + 8 * b, p.M - 1) * C + lane;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      const int rr = 8 * b + q, r = 16 * wave + rr;
      const bool ok = rr < nrows;
      const float mu = __shfl(mu_l, rr, 64), rs = __shfl(rs_l, rr, 64);
      const float* drow = lds + (r >> 5) * (2 * 32 * CST) + (r & 31) * CST;
      float xh[3], dxh[3];
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float d = ok ? drow[coff[j]] : 0.f;
        xh[j] = (xv[q][j] - mu) * rs;
        dxh[j] = d * g[j];
        s1 += dxh[j];
        s2 += dxh[j] * xh[j];
        dg[j] += d * xh[j];
        db[j] += d;
      }
      const float c1 = wave_sum(s1) * (1.0f / C), c2 = wave_sum(s2) * (1.0f / C);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const float a = ok ? av[q][j] : 0.f;
        da[j] += a;
        if (ok) crow[64 * j] = rs * (dxh[j] - c1 - xh[j] * c2) + a;
      }
      crow += C;
    }
  };
  fetch(0);                             // in flight while the accumulators are staged
  float* cs = lds + wave * (32 * CST);
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) cs[acc_row(r, hi) * CST + 32 * j + l31] = acc[0][j][r];
  __syncthreads();
  reduce(0);
  __builtin_amdgcn_sched_barrier(0);    // keep the second batch's loads behind the first batch's arithmetic (register budget)
  fetch(1);
  reduce(1);
  __syncthreads();                      // every wave is done with the staged tile: reuse it for the column partials
  float* red = lds;                     // [3][4][192]
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    red[(0 * 4 + wave) * C + lane + 64 * j] = dg[j];
    red[(1 * 4 + wave) * C + lane + 64 * j] = db[j];
    red[(2 * 4 + wave) * C + lane + 64 * j] = da[j];
  }
  __syncthreads();
  const int np = add ? 3 : 2;
  float* part = p.ln_part + (long long)mt * np * C;
  for (int c = tid; c < np * C; c += 256) {
    const int q = c / C, cc = c % C;
    part[c] = (red[(q * 4 + 0) * C + cc] + red[(q * 4 + 1) * C + cc]) + (red[(q * 4 + 2) * C + cc] + red[(q * 4 + 3) * C + cc]);
  }
}

// Tile epilogue.  acc: the wave's (32 TM) x (32 TN) accumulators (lane = column, register = row); lds: the workgroup's LDS
// (>= 4 * 32 * TM * (32 TN + 4) floats when STAGED), free to overwrite; (m0, n0) tile origin, (mt) row-panel index,
// zb / zid batch / split indices as in the kernels.
template <int TM, int TN, bool STAGED, bool BFIO = false>
RP_DEV void tile_epilogue(const GemmP& p, f32x16 (&acc)[TM][TN], float* lds, int m0, int n0, int mt, int zb, int zid) {
  constexpr int BM = 64 * TM, BN = 64 * TN;
  constexpr int CST = 32 * TN + 4;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  // ---- epilogue: one wave-uniform switch, then straight-line code per element --------------------
  float* C = p.C + zb * p.sc;
  const bool partial = p.split_k > 1;
  if (partial) C = p.C + (long long)zid * p.M * p.N;   // workspace slab [z][M][N]
  const int ldc = partial ? p.N : p.ldc;
  const float* bias = p.bias;
  float* pre_out = p.pre_out ? p.pre_out + zb * p.sc : nullptr;
  const float* aux = p.aux ? p.aux + zb * p.sc : nullptr;
  const float* res = p.residual ? p.residual + zb * p.sc : nullptr;
  const int mode = partial ? EPI_RAW : p.epi_mode;
  const bool interior = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  if (mode == EPI_LNBWD) {               // rp_gemm only selects it with the 64 x 192 tile (N == 192, one column tile)
    if constexpr (TM == 1 && TN == 3 && STAGED) lnbwd_epilogue(p, acc, lds, m0, mt, zb);
    return;
  }

  if (STAGED && (p.N & 3) == 0 && mode != EPI_GENERIC) {
    // LDS-staged epilogue, one straight-line instance per mode (wave-uniform switch outside every loop): see staged_epilogue
    switch (mode) {
      case EPI_RAW: staged_epilogue<TM, TN, EPI_RAW, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      case EPI_BIAS: staged_epilogue<TM, TN, EPI_BIAS, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      case EPI_BIAS_RES: staged_epilogue<TM, TN, EPI_BIAS_RES, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      case EPI_RES: staged_epilogue<TM, TN, EPI_RES, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      case EPI_BIAS_GELU: staged_epilogue<TM, TN, EPI_BIAS_GELU, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      case EPI_BIAS_GELU_PRE: staged_epilogue<TM, TN, EPI_BIAS_GELU_PRE, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      case EPI_BIAS_RELU: staged_epilogue<TM, TN, EPI_BIAS_RELU, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      case EPI_DGELU: staged_epilogue<TM, TN, EPI_DGELU, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
      default: staged_epilogue<TM, TN, EPI_DRELU, BFIO>(p, acc, lds, m0, n0, mt, C, ldc, bias, pre_out, aux, res, partial); break;
    }
  } else {
#define RP_EPI_LOOP(BODY)                                                              \
  _Pragma("unroll") for (int i = 0; i < TM; ++i)                                       \
  _Pragma("unroll") for (int j = 0; j < TN; ++j) {                                     \
    const int n = n0 + wn0 + 32 * j + l31;                                             \
    _Pragma("unroll") for (int r = 0; r < 16; ++r) {                                   \
      const int m = m0 + wm0 + 32 * i + acc_row(r, hi);                                \
      if ((interior || (m < p.M && n < p.N))) {                          \
        const long long off = (long long)m * ldc + n;                                  \
        float v = acc[i][j][r];                                                        \
        BODY;                                                                          \
        C[off] = v;                                                                    \
      }                                                                                \
    }                                                                                  \
  }
  if (mode == EPI_RAW) {
    RP_EPI_LOOP((void)0)
  } else {
    GemmP q = p;
    q.pre_out = pre_out; q.aux = aux; q.residual = res;
    RP_EPI_LOOP(v = epilogue(v, m, n, q))
  }
  }
#undef RP_EPI_LOOP
}

// gemm_dma.hip: LDS-DMA-staged main loop for precision 0 when every split holds whole 32-wide k-tiles
bool dma_eligible(const GemmP& p);
void launch_dma(const GemmP& p, int nz, int a_layout, int b_layout, int tm, int tn, hipStream_t st);

}  // namespace rpgemm
