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

template <bool LN>
__global__ __launch_bounds__(NT, 3) void linear_rows_kernel(RowsP p) {
  __shared__ __attribute__((aligned(16))) float wt[2][WT];
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
  unsigned off[DMA];
#pragma unroll
  for (int r = 0; r < DMA; ++r) {
    const int pp = r * NT + tid, row = pp / 48, ch = (pp % 48) ^ (row & 15);
    off[r] = (unsigned)((row * C + ch * 4) * 4);
  }
  const unsigned l0 = lds_byte_addr(&wt[0][0]) + wave * 1024;
  auto issue = [&](int c, int buf) {
    const float* src = uniform_ptr(p.w + (long long)c * CH * C);
#pragma unroll
    for (int r = 0; r < DMA; ++r) glds16(src, off[r], l0 + buf * (WT * 4) + r * NT * 16);
  };
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
  const bool has_bias = p.bias != nullptr, pre = p.ypre != nullptr, has_res = p.res != nullptr, has_aux = p.aux != nullptr, want_cs = p.colpart != nullptr;
  while (it < end) {
    const int tile = it / p.nchunk, c0 = it % p.nchunk, c1 = min(p.nchunk, c0 + end - it);
    const int row = tile * ROWS + wave * 16 + j;
    const bool live = row < p.M;
    const long long rclamp = min(row, p.M - 1);
    const float* xr = p.x + rclamp * C + 4 * q;
    float xn[48];
#pragma unroll
    for (int t = 0; t < 12; ++t) {
      const float4 v = ld4(xr + 16 * t);
      xn[4 * t] = v.x; xn[4 * t + 1] = v.y; xn[4 * t + 2] = v.z; xn[4 * t + 3] = v.w;
    }
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
        const float4 g = ld4(aux_lds + 16 * t + 4 * q), bb = ld4(aux_lds + C + 16 * t + 4 * q);
        xn[4 * t] = (xn[4 * t] - mu) * rs * g.x + bb.x;
        xn[4 * t + 1] = (xn[4 * t + 1] - mu) * rs * g.y + bb.y;
        xn[4 * t + 2] = (xn[4 * t + 2] - mu) * rs * g.z + bb.z;
        xn[4 * t + 3] = (xn[4 * t + 3] - mu) * rs * g.w + bb.w;
        if (p.xn && owner) st4(p.xn + (long long)row * C + 16 * t + 4 * q, make_float4(xn[4 * t], xn[4 * t + 1], xn[4 * t + 2], xn[4 * t + 3]));
      }
      if (owner && q == 0) {
        if (p.mean) p.mean[row] = mu;
        if (p.rstd) p.rstd[row] = rs;
      }
    }
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
        if (pre) {
          st4(p.ypre + po, pp0);
          st4(p.ypre + po + 16, pp1);
        }
        st4(p.y + po, pv0);
        st4(p.y + po + 16, pv1);
      }
      pending = false;
    };
    for (int c = c0; c < c1; ++c, ++it) {
      RP_STAMP(0);
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      RP_STAMP(1);
      asm volatile("s_barrier" ::: "memory");                                       // W(c) landed; other buffer free
      RP_STAMP(2);
      flush();
      if (it + 1 < end) issue((it + 1) % p.nchunk, buf ^ 1);
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
        g0 = ld4(p.aux + o);
        g1 = ld4(p.aux + o + 16);
        asm volatile("" ::: "memory");
      }
      f32x4v h0 = {0.f, 0.f, 0.f, 0.f}, h1 = {0.f, 0.f, 0.f, 0.f};
      const float* a0p = &wt[buf][0] + j * C;
      const float* a1p = a0p + 16 * C;
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
      RP_STAMP(3);
      float4 v0 = make_float4(h0[0] + ba.x, h0[1] + ba.y, h0[2] + ba.z, h0[3] + ba.w);
      float4 v1 = make_float4(h1[0] + bb.x, h1[1] + bb.y, h1[2] + bb.z, h1[3] + bb.w);
      pp0 = v0;
      pp1 = v1;
      if (has_aux) {
        asm volatile("" : "+v"(g0.x), "+v"(g0.y), "+v"(g0.z), "+v"(g0.w), "+v"(g1.x), "+v"(g1.y), "+v"(g1.z), "+v"(g1.w) :: "memory");
        v0 = make_float4(v0.x * gelu_grad_fast(g0.x), v0.y * gelu_grad_fast(g0.y), v0.z * gelu_grad_fast(g0.z), v0.w * gelu_grad_fast(g0.w));
        v1 = make_float4(v1.x * gelu_grad_fast(g1.x), v1.y * gelu_grad_fast(g1.y), v1.z * gelu_grad_fast(g1.z), v1.w * gelu_grad_fast(g1.w));
      }
      if (p.act == 1) {
        v0 = make_float4(gelu_fast(v0.x), gelu_fast(v0.y), gelu_fast(v0.z), gelu_fast(v0.w));
        v1 = make_float4(gelu_fast(v1.x), gelu_fast(v1.y), gelu_fast(v1.z), gelu_fast(v1.w));
      }
      pv0 = make_float4(v0.x + r0.x, v0.y + r0.y, v0.z + r0.z, v0.w + r0.w);
      pv1 = make_float4(v1.x + r1.x, v1.y + r1.y, v1.z + r1.z, v1.w + r1.w);
      if (want_cs) {     // column sums of this chunk over the wave's rows (fixed DPP tree), combined across waves after the barrier
        const float m = live ? 1.f : 0.f;
        float4 c0v = make_float4(row16_sum(m * pv0.x), row16_sum(m * pv0.y), row16_sum(m * pv0.z), row16_sum(m * pv0.w));
        float4 c1v = make_float4(row16_sum(m * pv1.x), row16_sum(m * pv1.y), row16_sum(m * pv1.z), row16_sum(m * pv1.w));
        if (j == 0) {
          st4(&cs_lds[buf][0] + wave * CH + 4 * q, c0v);
          st4(&cs_lds[buf][0] + wave * CH + 16 + 4 * q, c1v);
        }
        cs_par = buf;
        cs_tile = tile;
        cs_chunk = c;
      }
      po = o;
      pending = true;
      buf ^= 1;
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

template <bool LN>
int rows_slots() {
  static int slots = 0;
  if (!slots) {
    int dev = 0, cus = 256, per_cu = 1;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, linear_rows_kernel<LN>, NT, 0);
    slots = cus * (per_cu > 0 ? per_cu : 1);
  }
  return slots;
}

}  // namespace

extern "C" int rp_linear_rows192_tile_rows(void) { return ROWS; }

extern "C" int rp_linear_rows192(const float* x, const float* w, const float* bias, const float* residual, const float* ln_gamma,
                                 const float* ln_beta, float eps, float* y, float* y_pre, float* xn_out, float* mean_out,
                                 float* rstd_out, const float* dact_aux, float* colsum_part, int M, int N, int K, int act,
                                 void* stream) {
  if (M <= 0 || K != C || N <= 0 || N % CH != 0 || N > MAXN || !x || !w || !y || act < 0 || act > 1) return RP_EBADSHAPE;
  if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return RP_EBADSHAPE;
  const bool ln = ln_gamma != nullptr;
  if (!ln && (xn_out || mean_out || rstd_out)) return RP_EBADSHAPE;
  RowsP p{x, w, bias, residual, ln_gamma, ln_beta, y, y_pre, xn_out, mean_out, rstd_out, dact_aux, colsum_part, M, N, eps, act, N / CH, (M + ROWS - 1) / ROWS, 0, 0};
  const long long items = (long long)p.tiles * p.nchunk;
  const int slots = ln ? rows_slots<true>() : rows_slots<false>();
  const int G = (int)(items < slots ? items : slots);
  p.base = (int)(items / G);
  p.rem = (int)(items % G);
  hipStream_t st = (hipStream_t)stream;
  if (ln) hipLaunchKernelGGL(linear_rows_kernel<true>, dim3(G), dim3(NT), 0, st, p);
  else hipLaunchKernelGGL(linear_rows_kernel<false>, dim3(G), dim3(NT), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
