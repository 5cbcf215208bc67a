// dx_lnbwd_bf16.hip -- input gradient of a Linear that sits behind a LayerNorm, on the bf16 data path (BASELINE.json configs[4]):
//
//     dxn = dY W                      dY [M, K] bf16 (K = 576: the qkv Linear, vision_transformer.py:323), W [K, 192]
//     dx  = LayerNorm'(dxn; x, gamma, mean, rstd) + add        (Block.forward :352: x + attn(norm1(x)); `add` = the residual branch)
//     + per-tile partial column sums of  dxn o xhat (dgamma), dxn (dbeta) and `add` (the bias gradient of the Linear that made it)
//
// "Output-resident" sibling of linear_rows.hip: a wave owns 16 token rows and keeps their whole 192-wide OUTPUT row in 48 accumulator
// registers (lane (j, q): units 16 b + 4 q .. + 3 of row j for the 12 unit blocks b) while the contraction index streams by:
//   * dY rows are loaded straight into MFMA B-operand shape (lane (j, q) takes the 8 consecutive k = 32 c + 8 q .. of its row: one
//     16-byte load per 32-wide k chunk, 18 of them), no LDS, no conversion;
//   * W^T [192][K] bf16 (a host-side copy) streams through LDS in [192 units][32 k] chunks (12 KB, LDS-DMA, 64-byte unit rows with the
//     16-byte slot XOR-swizzled by (-(unit >> 2)) & 3 -- ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ...
//     (MI355X_MICROARCH.md): with this key the 16 lanes of every group hit 16 distinct slots), one ds_read_b128 = one A operand of v_mfma_f32_16x16x32_bf16, 12 MFMAs per chunk;
//   * the LayerNorm backward runs on the accumulators: two cross-lane adds give the row sums, DPP row sums + an 8-wave LDS reduction
//     the tile's column partials; dx leaves as 16-byte fp32 stores.  The [M,192] gradient of the LayerNorm OUTPUT never exists in
//     memory, and the dY operand is read once, as bf16.
// Replaces rp_gemm(precision 1, io_bf16 bit 0, ln_*) for this shape: 194 us -> see profiles/ (128 pairs).
#include "bf16_path.h"
#include "../../include/relpose_hip.h"

namespace {
using namespace bf16path;

typedef float f32x4v __attribute__((ext_vector_type(4)));
RP_DEV f32x4v mfma16bf(bf16x8 a, bf16x8 b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0); }

constexpr int C = 192;            // output width = LayerNorm width
constexpr int NWV = 4;            // waves per workgroup: 64 rows per tile; three workgroups per CU run in different phases
constexpr int ROWS = NWV * 16;
constexpr int CHUNK_EL = C * 32;  // bf16 elements of one staged W^T chunk (12 KB)

struct DxP {
  const bf16_t* dy; const bf16_t* wt;           // dY [M][K] bf16; W^T [192][K] bf16
  const float *x, *gamma, *mean, *rstd, *add;
  float *dx, *part;                             // part [tiles][np][192]
  int M, K, np;
};

template <int KC>                                // KC = K / 32 chunks (18 for K = 576)
__global__ __launch_bounds__(NWV * 64, 3) void dx_lnbwd_bf16_kernel(DxP p) {
  __shared__ __attribute__((aligned(16))) bf16_t Ws[3][CHUNK_EL];
  __shared__ float red[3][NWV][C];
  __shared__ __attribute__((aligned(16))) float gam[C];      // gamma (read back after the main loop's barriers)
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, q = lane >> 4;
  const int tile = blockIdx.x;
  const int row = tile * ROWS + wave * 16 + j;
  const bool live = row < p.M;
  const long long rc = min(row, p.M - 1);

  // DMA plan: chunk = 12 pieces of 1 KB (16 unit rows x 64 B); wave w moves pieces 3 w .. 3 w + 2
  unsigned woff[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int u = (3 * wave + i) * 16 + (lane >> 2), slot = lane & 3;
    woff[i] = (unsigned)(u * p.K * 2 + ((slot ^ ((0 - (u >> 2)) & 3)) << 4));
  }
  const unsigned ws0 = lds_addr_of(&Ws[0][0]) + wave * 3072;
  auto issue = [&](int c, int buf) {
    const void* src = uniform_vptr(p.wt + c * 32);
#pragma unroll
    for (int i = 0; i < 3; ++i) glds16b(src, woff[i], ws0 + buf * (CHUNK_EL * 2) + i * 1024);
  };
  issue(0, 0);
  issue(1, 1);
  if (tid < C) gam[tid] = p.gamma[tid];

  // the wave's 16 dY rows as B operands: lane (j, q) holds k = 32 c + 8 q .. + 7 of row j
  bf16x8 yb[KC];
  {
    const bf16_t* yr = p.dy + rc * p.K + 8 * q;
#pragma unroll
    for (int c = 0; c < KC; ++c) yb[c] = *reinterpret_cast<const bf16x8*>(yr + 32 * c);
  }
  f32x4v acc[12];
#pragma unroll
  for (int b = 0; b < 12; ++b) acc[b] = f32x4v{0.f, 0.f, 0.f, 0.f};
  // A operand of unit block b: unit row 16 b + j, its 16-byte slot q (k = 8 q ..) swizzled by (-(unit >> 2)) & 3 = (-(j >> 2)) & 3 (16 b >> 2 is a multiple of 4)
  const int aoff = j * 32 + ((q ^ ((0 - (j >> 2)) & 3)) << 3);

#pragma unroll
  for (int c = 0; c < KC; ++c) {
    // chunk c landed (chunk c + 1 may stay in flight: 3 DMA instructions per wave and chunk) for everybody; everybody is past chunk c - 1
    // (lgkmcnt(0): this wave's LDS reads of chunk c - 1 have RETURNED before it signals the barrier -- the MFMAs that consume them may be
    // scheduled behind the barrier, and the slot they read is refilled by the DMA issued right after it)
    if (c + 1 < KC) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (c + 2 < KC) issue(c + 2, (c + 2) % 3);
    const bf16_t* wtile = Ws[c % 3] + aoff;
#pragma unroll
    for (int b = 0; b < 12; ++b) acc[b] = mfma16bf(ld_bf16x8_lds(wtile + b * 16 * 32), yb[c], acc[b]);
  }

  // ---- LayerNorm backward on the accumulators: lane (j, q) holds dxn[row j][16 b + 4 q + e], e = 0..3 -------------------------------
  const float mu = p.mean[rc], rs = p.rstd[rc];
  float xh[48], s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int b = 0; b < 12; ++b) {
    // (requesting these rows under the last MFMA chunks was measured: 131 -> 143 us -- the extra live registers cost more than the exposed latency)
    const float4 xv = ld4(p.x + rc * C + 16 * b + 4 * q), gv = ld4(gam + 16 * b + 4 * q);
    const float xs[4] = {xv.x, xv.y, xv.z, xv.w}, gs[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float h = (xs[e] - mu) * rs, d = acc[b][e];
      xh[4 * b + e] = h;
      const float dg = d * gs[e];
      s1 += dg;
      s2 = fmaf(dg, h, s2);
    }
  }
  s1 += __shfl_xor(s1, 16, 64); s1 += __shfl_xor(s1, 32, 64);
  s2 += __shfl_xor(s2, 16, 64); s2 += __shfl_xor(s2, 32, 64);
  const float c1 = s1 * (1.0f / C), c2 = s2 * (1.0f / C);
  const float m = live ? 1.f : 0.f;
  // the residual-branch rows are requested AHEAD of the stores (dx may alias add for the compiler: written as "load, compute, store"
  // per block the twelve blocks were twelve serialized memory round trips -- s_waitcnt vmcnt(0) before every store, and vmcnt retires
  // in order, so each wait also covered the previous store); a window of PF blocks stays in flight
  constexpr int PF = 4;
  float4 avq[12];
#pragma unroll
  for (int b = 0; b < 12; ++b) avq[b] = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.add) {
#pragma unroll
    for (int b = 0; b < PF; ++b) avq[b] = ld4(p.add + rc * C + 16 * b + 4 * q);
  }
#pragma unroll
  for (int b = 0; b < 12; ++b) {
    const float4 gv = ld4(gam + 16 * b + 4 * q);
    const float gs[4] = {gv.x, gv.y, gv.z, gv.w};
    if (p.add && b + PF < 12) avq[b + PF] = ld4(p.add + rc * C + 16 * (b + PF) + 4 * q);
    const float4 av = avq[b];
    const float as[4] = {av.x, av.y, av.z, av.w};
    float o[4], cg[4], cb[4], ca[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = acc[b][e], h = xh[4 * b + e];
      o[e] = rs * (d * gs[e] - c1 - h * c2) + as[e];
      cg[e] = row16_sum(m * d * h);          // column partials over the wave's 16 rows (fixed DPP tree)
      cb[e] = row16_sum(m * d);
      ca[e] = row16_sum(m * as[e]);
    }
    if (live) st4(p.dx + (long long)row * C + 16 * b + 4 * q, make_float4(o[0], o[1], o[2], o[3]));
    if (j == 0) {
      st4(&red[0][wave][16 * b + 4 * q], make_float4(cg[0], cg[1], cg[2], cg[3]));
      st4(&red[1][wave][16 * b + 4 * q], make_float4(cb[0], cb[1], cb[2], cb[3]));
      st4(&red[2][wave][16 * b + 4 * q], make_float4(ca[0], ca[1], ca[2], ca[3]));
    }
  }
  __syncthreads();
  for (int e = tid; e < p.np * C; e += NWV * 64) {
    const int k = e / C, col = e % C;
    float s = 0.f;
#pragma unroll
    for (int w = 0; w < NWV; ++w) s += red[k][w][col];
    p.part[((long long)tile * p.np + k) * C + col] = s;
  }
}

}  // namespace

extern "C" int rp_dx_lnbwd_bf16_tile_rows(void) { return ROWS; }

extern "C" int rp_dx_lnbwd_bf16(const void* dy, const void* wt, const float* x, const float* gamma, const float* mean, const float* rstd,
                                const float* add, float* dx, float* part, int M, int K, void* stream) {
  if (!dy || !wt || !x || !gamma || !mean || !rstd || !dx || !part || M <= 0) return RP_EBADSHAPE;
  if (K != 576) return RP_EUNSUPPORTED;
  if (((uintptr_t)dy | (uintptr_t)wt | (uintptr_t)x | (uintptr_t)dx | (uintptr_t)add) & 15) return RP_EALIGN;
  DxP p{(const bf16_t*)dy, (const bf16_t*)wt, x, gamma, mean, rstd, add, dx, part, M, K, add ? 3 : 2};
  hipLaunchKernelGGL((dx_lnbwd_bf16_kernel<18>), dim3((M + ROWS - 1) / ROWS), dim3(NWV * 64), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
