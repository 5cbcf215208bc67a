// emm.hip -- Essential Matrix Module kernels (rp_emm_apply, rp_emm_grad).
//
// Reference: CrossAttention.forward, ess branch (src/modules/vision_transformer.py:198-223), per image z of a
// pair (partner z^1) and head h:
//     S = scale * q_{z^1} k_z^T                       [576 x 576]
//     A = softmax(S, -1) * softmax(S, -2) = exp(2 S - rlse_i - clse_j)      (dual softmax, :205-206)
//     X = [v_z | pos | 0]                              [576 x 96], 70 live columns (:215-216)
//     F = X^T A X                                      [70 x 70] (:222-223)
// The 576x576 matrices never exist in memory: rlse / clse come from ONE statistics pass over S (rp_emm_stats: the attention
// kernel's statistics form with per-block column partials; two statistics-only passes in the bf16-MFMA mode); rp_emm_apply recomputes S tile by tile (one wave = 32 "owner" rows, tiles of 32 "loop"
// rows staged in LDS), forms A in registers, accumulates T = A X with the score accumulators used directly as
// the MFMA A operand, then contracts F_partial = X_blk^T T_blk per workgroup through LDS (the "LDS-tiled
// outer product" of the task statement).  Six workgroup partials per (z, h) are summed in fixed order by
// rp_emm_finalize (rowwise.hip) -> deterministic.
// rp_emm_grad is the gradient pass (see DESIGN.md "EMM backward" for the algebra):
//     dA = W X^T,  dS = 2 A dA - R rho_i - C gamma_j,  d owner = scale * dS * other
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

constexpr int NTOK = 576;
constexpr int NTILE = NTOK / 32;
constexpr int KST = 68;
constexpr int XW = 96;      // padded width of X / T / W rows
constexpr int NW = 3;       // waves per workgroup: 96 owner rows
constexpr int NT = NW * 64;

struct EmmP {
  const float* qkv; int ld;
  const float* x; const float* w;
  const float* rlse; const float* clse; const float* rho; const float* gamma;
  float* t_out; float* f_part; float* dqkv;
  int H; float scale; int swap; int ZH;
  int single;            // use_single_softmax (vision_transformer.py:201-203): A = softmax(S, -1) only
  const float* x_left;   // cross_features (:218-220): left operand of F = X_L^T A X comes from the partner image
  const float* s_in;     // stored-S forms: the score tiles rp_emm_stats wrote (log2 units): [Z][H][18 query blocks][18 key tiles][1024], element
                         // (query i, key j) at float ((j >> 2) * 32 + i) * 4 + (j & 3)
  float* ds;             // emm_grad, owner = query pass: optional [Z][H][18 j-blocks][18 i-blocks][16][64] = scale * dS_ij in 32x32 tiles
                         // (the MFMA accumulator image of the tile), so the key-side gradient dk = scale dS^T q is one rp_ds_matmul
                         // instead of a second pass that recomputes S and dA
};

template <int NTH>
RP_DEV void kv_gload(const float* base, int ld, int tid, float4 (&r)[(512 + NTH - 1) / NTH]) {
#pragma unroll
  for (int j = 0; j < (512 + NTH - 1) / NTH; ++j) {
    const int f = min(tid + NTH * j, 511);     // surplus threads duplicate the last element (no exec-masked guard)
    r[j] = ld4(base + (long long)(f >> 4) * ld + (f & 15) * 4);
  }
}
template <int NTH>
RP_DEV void kv_sstore(float* s, int tid, const float4 (&r)[(512 + NTH - 1) / NTH]) {
#pragma unroll
  for (int j = 0; j < (512 + NTH - 1) / NTH; ++j) {
    const int f = min(tid + NTH * j, 511);
    st4(s + (f >> 4) * KST + (f & 15) * 4, r[j]);
  }
}

template <bool BF>
RP_DEV f32x16 score_tile(const float* Ks, int l31, int hi, const float (&breg)[32], const bf16x8 (&bpk)[4]) {
  f32x16 s = zero16();
  const float* kr = Ks + l31 * KST + 32 * hi;
  if (BF) {      // bf16 operand mode (common.h): 8 consecutive k-steps of a lane = one v_mfma_f32_32x32x16_bf16
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float4 x = ld4(kr + 8 * c), y = ld4(kr + 8 * c + 4);
      s = mfma_bf(pack8(x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w), bpk[c], s);
    }
    return s;
  }
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 kf = ld4(kr + 4 * c);
    s = mfma32(kf.x, breg[4 * c + 0], s);
    s = mfma32(kf.y, breg[4 * c + 1], s);
    s = mfma32(kf.z, breg[4 * c + 2], s);
    s = mfma32(kf.w, breg[4 * c + 3], s);
  }
  return s;
}

RP_DEV void load_owner(const float* row_ptr, int hi, float mul, float (&reg)[32]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 x = ld4(row_ptr + 32 * hi + 4 * c);
    reg[4 * c + 0] = x.x * mul; reg[4 * c + 1] = x.y * mul; reg[4 * c + 2] = x.z * mul; reg[4 * c + 3] = x.w * mul;
  }
}

// ------------------------------------------------------------------------------------------------
template <bool BF>
__global__ __launch_bounds__(NT, 3) void emm_apply_kernel(EmmP p) {
  // LDS carve: loop phase  Ks[2][32*68] | Xs[2][32*96] | Cl[2][32]   (10560 floats)
  //            F phase     Ts[96][96]                                   ( 9216 floats, aliases the above)
  __shared__ __attribute__((aligned(16))) float lds[2 * 32 * KST + 2 * 32 * XW + 64];
  float* Ks = lds;
  float* Xs = lds + 2 * 32 * KST;
  float* Cl = Xs + 2 * 32 * XW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh_, wgi;
  if (!xcd_problem(NTILE / NW, p.ZH, zh_, wgi)) return;
  const int h = zh_ % p.H, z = zh_ / p.H;
  const int wg0 = wgi * (NW * 32);
  const int o0 = wg0 + wave * 32;
  const int own_img = p.swap ? z : (z ^ 1), own_col = (p.swap ? 192 : 0) + h * 64;
  const int loop_img = p.swap ? (z ^ 1) : z, loop_col = (p.swap ? 0 : 192) + h * 64;
  const long long zh = (long long)z * p.H + h;
  const float* own_lse = (p.swap ? p.clse : p.rlse) + zh * NTOK;
  const float* loop_lse = (p.swap ? p.rlse : p.clse) + zh * NTOK;
  const float* lb = p.qkv + (long long)loop_img * NTOK * p.ld + loop_col;
  const float* xb = p.x + zh * NTOK * XW;

  float oreg[32];
  load_owner(p.qkv + ((long long)own_img * NTOK + o0 + l31) * p.ld + own_col, hi, p.scale * RP_LOG2E, oreg);
  bf16x8 opk[4];
  if (BF) {
#pragma unroll
    for (int c = 0; c < 4; ++c) opk[c] = pack8(&oreg[8 * c]);
  }
  const float ls_o = own_lse[o0 + l31] * RP_LOG2E;

  f32x16 tacc[3] = {zero16(), zero16(), zero16()};
  float4 kpre[3], xpre[4];
  float cpre = 0.f;
  auto x_gload = [&](int t) {   // 32 rows x 24 float4 = 768 float4 -> 4 per thread
#pragma unroll
    for (int j = 0; j < 4; ++j) xpre[j] = ld4(xb + (long long)t * 32 * XW + (tid + NT * j) * 4);
  };
  auto x_sstore = [&](float* s) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st4(s + (tid + NT * j) * 4, xpre[j]);
  };
  kv_gload<NT>(lb, p.ld, tid, kpre);
  x_gload(0);
  // branch-free (every thread loads; lane & 31 picks the value): an exec-masked load here made hipcc wait vmcnt(0) -- for the tile
  // prefetch just issued -- at the top of every iteration (see attn_bwd_dkdv_kernel)
  const float* csrc = loop_lse + (tid & 31);
  cpre = csrc[0] * RP_LOG2E;
  kv_sstore<NT>(Ks, tid, kpre);
  x_sstore(Xs);
  if (tid < 32) Cl[tid] = cpre;
  __syncthreads();

  for (int t = 0; t < NTILE; ++t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) {
      kv_gload<NT>(lb + (long long)(t + 1) * 32 * p.ld, p.ld, tid, kpre);
      x_gload(t + 1);
      cpre = csrc[(t + 1) * 32] * RP_LOG2E;
    }
    f32x16 s = score_tile<BF>(Ks + cur * 32 * KST, l31, hi, oreg, opk);   // S^T[loop][owner]
    const float* cl = Cl + cur * 32;
    const float* xs = Xs + cur * 32 * XW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ll = cl[acc_row(r, hi)];
      // dual softmax: exp(2S - rlse_i - clse_j); single: exp(S - rlse_i) with i = owner (swap == 0) or loop row (swap != 0)
      s[r] = p.single ? fast_exp2(s[r] - (p.swap ? ll : ls_o)) : fast_exp2(2.0f * s[r] - ls_o - ll);
    }
    // T[owner][c] += sum_loop A[owner][loop] X[loop][c] : A operand = s (lane = owner), B operand = X rows
    if (BF) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float b0[8], b1[8], b2[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* xr = xs + acc_row(8 * g + j, hi) * XW + l31;
          b0[j] = xr[0]; b1[j] = xr[32]; b2[j] = xr[64];
        }
        const bf16x8 ap = pack8(s[8 * g], s[8 * g + 1], s[8 * g + 2], s[8 * g + 3], s[8 * g + 4], s[8 * g + 5], s[8 * g + 6], s[8 * g + 7]);
        tacc[0] = mfma_bf(ap, pack8(b0), tacc[0]);
        tacc[1] = mfma_bf(ap, pack8(b1), tacc[1]);
        tacc[2] = mfma_bf(ap, pack8(b2), tacc[2]);
      }
    } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* xr = xs + acc_row(r, hi) * XW + l31;
      tacc[0] = mfma32(s[r], xr[0], tacc[0]);
      tacc[1] = mfma32(s[r], xr[32], tacc[1]);
      tacc[2] = mfma32(s[r], xr[64], tacc[2]);
    }
    }
    if (t + 1 < NTILE) {
      kv_sstore<NT>(Ks + (cur ^ 1) * 32 * KST, tid, kpre);
      x_sstore(Xs + (cur ^ 1) * 32 * XW);
      if (tid < 32) Cl[(cur ^ 1) * 32 + tid] = cpre;
    }
    __syncthreads();
  }

  if (p.t_out) {
    float* tb = p.t_out + (zh * NTOK + o0) * XW;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[acc_row(r, hi) * XW + 32 * nb + l31] = tacc[nb][r];
  }
  if (!p.f_part) return;

  // ---- F_partial[a][c] = sum_{i in this workgroup's 96 rows} X[i][a] T[i][c] ----------------------
  float* Ts = lds;
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ts[(wave * 32 + acc_row(r, hi)) * XW + 32 * nb + l31] = tacc[nb][r];
  __syncthreads();
  f32x16 facc[3] = {zero16(), zero16(), zero16()};
  const float* xlb = p.x_left ? p.x_left + ((long long)(z ^ 1) * p.H + h) * NTOK * XW : xb;
  const float* xa = xlb + (long long)wg0 * XW + 32 * wave + l31;   // column a = 32*wave + l31 of the LEFT operand's rows
  if (BF) {
#pragma unroll 2
    for (int g = 0; g < 6; ++g) {
      float av[8], b0[8], b1[8], b2[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int i = 48 * hi + 8 * g + j;
        av[j] = xa[(long long)i * XW];
        const float* tr = Ts + i * XW + l31;
        b0[j] = tr[0]; b1[j] = tr[32]; b2[j] = tr[64];
      }
      const bf16x8 ap = pack8(av);
      facc[0] = mfma_bf(ap, pack8(b0), facc[0]);
      facc[1] = mfma_bf(ap, pack8(b1), facc[1]);
      facc[2] = mfma_bf(ap, pack8(b2), facc[2]);
    }
  } else {
#pragma unroll 4
  for (int t = 0; t < 48; ++t) {
    const int i = 48 * hi + t;
    const float av = xa[(long long)i * XW];
    const float* tr = Ts + i * XW + l31;
    facc[0] = mfma32(av, tr[0], facc[0]);
    facc[1] = mfma32(av, tr[32], facc[1]);
    facc[2] = mfma32(av, tr[64], facc[2]);
  }
  }
  float* fb = p.f_part + ((zh * (NTOK / (NW * 32)) + wgi) * XW + 32 * wave) * XW;
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) fb[acc_row(r, hi) * XW + 32 * nb + l31] = facc[nb][r];
}

// ------------------------------------------------------------------------------------------------
// rp_emm_apply over STORED score tiles (p.s_in, written by rp_emm_stats): the same T = A X (+ F partials) without the q k^T product --
// 48 MFMAs per tile instead of 80, no K tile in LDS, no owner rows in registers.  A wave's tile comes straight from HBM one tile ahead:
//   SWAP = false (owner = queries, the forward): the stats wave that owned the same 32 queries wrote exactly this register image --
//                four contiguous 16-byte-per-lane loads;
//   SWAP = true  (owner = keys, U = A^T X_L of the backward): the transposed view of the same tiles -- register r of lane (key j, hi) is
//                query acc_row(r, hi): sixteen 4-byte loads 16 bytes apart (the gather is on the load side, where it is hidden).
template <bool SWAP>
__global__ __launch_bounds__(NT, 3) void emm_apply_s_kernel(EmmP p) {
  // LDS carve: loop phase  Xs[2][32*96] | Cl[2][32]   (6208 floats);   F phase  Ts[96][96]  (9216 floats, aliases the above)
  __shared__ __attribute__((aligned(16))) float lds[XW * XW];
  float* Xs = lds;
  float* Cl = lds + 2 * 32 * XW;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh_, wgi;
  if (!xcd_problem(NTILE / NW, p.ZH, zh_, wgi)) return;
  const int h = zh_ % p.H, z = zh_ / p.H;
  const int wg0 = wgi * (NW * 32);
  const int o0 = wg0 + wave * 32;
  const long long zh = (long long)z * p.H + h;
  const float* own_lse = (SWAP ? p.clse : p.rlse) + zh * NTOK;
  const float* loop_lse = (SWAP ? p.rlse : p.clse) + zh * NTOK;
  const float* xb = p.x + zh * NTOK * XW;
  const float ls_o = own_lse[o0 + l31] * RP_LOG2E;
  // tile (query block qb, key tile kt) at ((zh * 18 + qb) * 18 + kt) * 1024
  const float* sp = SWAP ? p.s_in + (zh * NTILE * NTILE + (o0 >> 5)) * 1024 + (l31 >> 2) * 128 + (l31 & 3) + 16 * hi      // loop tile t = query block: + t * 18 * 1024
                         : p.s_in + (zh * NTILE + (o0 >> 5)) * NTILE * 1024 + 4 * lane;                                   // loop tile t = key tile:    + t * 1024
  auto sload = [&](f32x16& v, int t) {
    if (SWAP) {
      const float* tp = sp + (long long)t * (NTILE * 1024);
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = tp[acc_row(r, 0) * 4];
    } else {
      const float* tp = sp + t * 1024;
#pragma unroll
      for (int g = 0; g < 4; ++g) {
        const float4 x = ld4(tp + 256 * g);
        v[4 * g] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
      }
    }
  };

  f32x16 tacc[3] = {zero16(), zero16(), zero16()};
  float4 xpre[4];
  float cpre = 0.f;
  auto x_gload = [&](int t) {   // 32 rows x 24 float4 = 768 float4 -> 4 per thread
#pragma unroll
    for (int j = 0; j < 4; ++j) xpre[j] = ld4(xb + (long long)t * 32 * XW + (tid + NT * j) * 4);
  };
  auto x_sstore = [&](float* d) {
#pragma unroll
    for (int j = 0; j < 4; ++j) st4(d + (tid + NT * j) * 4, xpre[j]);
  };
  f32x16 sa, sb;
  x_gload(0);
  const float* csrc = loop_lse + (tid & 31);      // branch-free (see emm_apply_kernel)
  cpre = csrc[0] * RP_LOG2E;
  sload(sa, 0);
  x_sstore(Xs);
  if (tid < 32) Cl[tid] = cpre;
  __syncthreads();

  auto step = [&](f32x16& s, f32x16& sn, int t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) {
      x_gload(t + 1);
      cpre = csrc[(t + 1) * 32] * RP_LOG2E;
      sload(sn, t + 1);
    }
    const float* cl = Cl + cur * 32;
    const float* xs = Xs + cur * 32 * XW;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float ll = cl[acc_row(r, hi)];
      s[r] = p.single ? fast_exp2(s[r] - (SWAP ? ll : ls_o)) : fast_exp2(2.0f * s[r] - ls_o - ll);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* xr = xs + acc_row(r, hi) * XW + l31;
      tacc[0] = mfma32(s[r], xr[0], tacc[0]);
      tacc[1] = mfma32(s[r], xr[32], tacc[1]);
      tacc[2] = mfma32(s[r], xr[64], tacc[2]);
    }
    if (t + 1 < NTILE) {
      x_sstore(Xs + (cur ^ 1) * 32 * XW);
      if (tid < 32) Cl[(cur ^ 1) * 32 + tid] = cpre;
    }
    __syncthreads();
  };
  for (int t = 0; t < NTILE; t += 2) {      // two register sets rotate without copies
    step(sa, sb, t);
    step(sb, sa, t + 1);
  }

  if (p.t_out) {
    float* tb = p.t_out + (zh * NTOK + o0) * XW;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int r = 0; r < 16; ++r) tb[acc_row(r, hi) * XW + 32 * nb + l31] = tacc[nb][r];
  }
  if (!p.f_part) return;

  // ---- F_partial[a][c] = sum_{i in this workgroup's 96 rows} X[i][a] T[i][c] (as emm_apply_kernel) ----------------------
  float* Ts = lds;
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) Ts[(wave * 32 + acc_row(r, hi)) * XW + 32 * nb + l31] = tacc[nb][r];
  __syncthreads();
  f32x16 facc[3] = {zero16(), zero16(), zero16()};
  const float* xlb = p.x_left ? p.x_left + ((long long)(z ^ 1) * p.H + h) * NTOK * XW : xb;
  const float* xa = xlb + (long long)wg0 * XW + 32 * wave + l31;   // column a = 32*wave + l31 of the LEFT operand's rows
#pragma unroll 4
  for (int t = 0; t < 48; ++t) {
    const int i = 48 * hi + t;
    const float av = xa[(long long)i * XW];
    const float* tr = Ts + i * XW + l31;
    facc[0] = mfma32(av, tr[0], facc[0]);
    facc[1] = mfma32(av, tr[32], facc[1]);
    facc[2] = mfma32(av, tr[64], facc[2]);
  }
  float* fb = p.f_part + ((zh * (NTOK / (NW * 32)) + wgi) * XW + 32 * wave) * XW;
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int r = 0; r < 16; ++r) fb[acc_row(r, hi) * XW + 32 * nb + l31] = facc[nb][r];
}

// ------------------------------------------------------------------------------------------------
constexpr int XG = 72;       // live columns of X / W used by the dA contraction (70 rounded up to even, x4)
constexpr int XGS = 76;      // LDS row stride for the X tile read along c with ds_read_b128

// 2-wave workgroups: the kernel needs ~250 VGPRs (2 waves/SIMD = 8 wave slots per CU); 4 x 2 waves fill them, 2 x 3 do not
constexpr int GW = 2, GT = GW * 64;
// LS (exact fp32, owner = query pass): the score tile is LOADED from p.s_in (what rp_emm_stats stored: the same register image, four
// contiguous 16-byte-per-lane loads one tile ahead) instead of recomputed -- 68 MFMAs per tile instead of 100, no owner q rows in registers.
template <bool BF, bool LS = false>
__global__ __launch_bounds__(GT, 2) void emm_grad_kernel(EmmP p) {
  __shared__ __attribute__((aligned(16))) float Ks[2][32 * KST];
  __shared__ __attribute__((aligned(16))) float Xs[2][32 * XGS];
  __shared__ float Ll[2][64];   // loop-side lse [0..31] and rho/gamma [32..63]
  __shared__ __attribute__((aligned(16))) float Tst[(BF || LS) ? 1 : GW][(BF || LS) ? 4 : 256];      // per-wave staging of the stored dS tile (1 KB: four workgroups per CU stay)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh_, wgi;
  if (!xcd_problem(NTILE / GW, p.ZH, zh_, wgi)) return;
  const int h = zh_ % p.H, z = zh_ / p.H;
  const int o0 = (wgi * GW + wave) * 32;
  const int own_img = p.swap ? z : (z ^ 1), own_col = (p.swap ? 192 : 0) + h * 64;
  const int loop_img = p.swap ? (z ^ 1) : z, loop_col = (p.swap ? 0 : 192) + h * 64;
  const long long zh = (long long)z * p.H + h;
  const float* own_lse = (p.swap ? p.clse : p.rlse) + zh * NTOK;
  const float* loop_lse = (p.swap ? p.rlse : p.clse) + zh * NTOK;
  const float* own_g = (p.swap ? p.gamma : p.rho) + zh * NTOK;
  const float* loop_g = (p.swap ? p.rho : p.gamma) + zh * NTOK;
  const float* lb = p.qkv + (long long)loop_img * NTOK * p.ld + loop_col;
  const float* xb = p.x + zh * NTOK * XW;

  float oreg[32], wreg[36];
  if (!LS) load_owner(p.qkv + ((long long)own_img * NTOK + o0 + l31) * p.ld + own_col, hi, p.scale * RP_LOG2E, oreg);
  const float* sp = LS ? p.s_in + (zh * NTILE + (o0 >> 5)) * NTILE * 1024 + 4 * lane : nullptr;      // tile t: + t * 1024
  auto sload = [&](f32x16& v, int t) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float4 x = ld4(sp + t * 1024 + 256 * g);
      v[4 * g] = x.x; v[4 * g + 1] = x.y; v[4 * g + 2] = x.z; v[4 * g + 3] = x.w;
    }
  };
  // LS: `scale` rides on the W rows and on rho / gamma (dA' = scale dA, ...): the dS formed below IS scale * dS -- the stored tile and the
  // dq operand -- with no multiply of its own in the loop, and the tile leaves as the four 16-byte runs each lane holds (no LDS staging)
  const float pre = LS ? p.scale : 1.0f;
  {
    const float* wr = p.w + (zh * NTOK + o0 + l31) * XW + 36 * hi;
#pragma unroll
    for (int c = 0; c < 9; ++c) {
      const float4 v = ld4(wr + 4 * c);
      wreg[4 * c] = v.x * pre; wreg[4 * c + 1] = v.y * pre; wreg[4 * c + 2] = v.z * pre; wreg[4 * c + 3] = v.w * pre;
    }
  }
  const float ls_o = own_lse[o0 + l31] * RP_LOG2E, g_o = own_g[o0 + l31] * pre;
  bf16x8 opk[4], wpk[5];         // bf16 mode: the 36 W columns of this half-wave as 4 x 8 + (4 live + 4 zero)
  if (BF) {
#pragma unroll
    for (int c = 0; c < 4; ++c) { opk[c] = pack8(&oreg[8 * c]); wpk[c] = pack8(&wreg[8 * c]); }
    wpk[4] = pack8(wreg[32], wreg[33], wreg[34], wreg[35], 0.f, 0.f, 0.f, 0.f);
  }

  f32x16 d0 = zero16(), d1 = zero16();
  float4 kpre[(512 + GT - 1) / GT], xpre[(576 + GT - 1) / GT];
  float lpre = 0.f;
  auto x_gload = [&](int t) {   // 32 rows x 18 float4 (72 cols) = 576 float4 -> 3 per thread
#pragma unroll
    for (int j = 0; j < (576 + GT - 1) / GT; ++j) {
      const int f = min(tid + GT * j, 575);
      xpre[j] = ld4(xb + ((long long)t * 32 + f / 18) * XW + (f % 18) * 4);
    }
  };
  auto x_sstore = [&](float* s) {
#pragma unroll
    for (int j = 0; j < (576 + GT - 1) / GT; ++j) {
      const int f = min(tid + GT * j, 575);
      st4(s + (f / 18) * XGS + (f % 18) * 4, xpre[j]);
    }
  };
  kv_gload<GT>(lb, p.ld, tid, kpre);
  x_gload(0);
  const float* lsrc = (tid & 32) ? loop_g + (tid & 31) : loop_lse + (tid & 31);      // branch-free, see emm_apply_kernel
  const float lmul = (tid & 32) ? pre : RP_LOG2E;
  lpre = lsrc[0] * lmul;
  f32x16 sa = zero16(), sb = zero16();
  if (LS) sload(sa, 0);
  kv_sstore<GT>(Ks[0], tid, kpre);
  x_sstore(Xs[0]);
  if (tid < 64) Ll[0][tid] = lpre;
  __syncthreads();

  auto step = [&](f32x16& s, f32x16& sn, int t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) {
      kv_gload<GT>(lb + (long long)(t + 1) * 32 * p.ld, p.ld, tid, kpre);
      x_gload(t + 1);
      lpre = lsrc[(t + 1) * 32] * lmul;
      if (LS) sload(sn, t + 1);
    }
    if (!LS) s = score_tile<BF>(Ks[cur], l31, hi, oreg, opk);      // S^T[loop][owner]
    f32x16 da = zero16();                                // dA^T[loop][owner] = sum_c X[loop][c] W[owner][c]
    {
      const float* xr = Xs[cur] + l31 * XGS + 36 * hi;
      if (BF) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const float4 x = ld4(xr + 8 * c), y = ld4(xr + 8 * c + 4);
          da = mfma_bf(pack8(x.x, x.y, x.z, x.w, y.x, y.y, y.z, y.w), wpk[c], da);
        }
        const float4 x = ld4(xr + 32);        // columns 32..35 of this half; the other four k-slots are explicit zeros
        da = mfma_bf(pack8(x.x, x.y, x.z, x.w, 0.f, 0.f, 0.f, 0.f), wpk[4], da);
      } else {
#pragma unroll
      for (int c = 0; c < 9; ++c) {
        const float4 xf = ld4(xr + 4 * c);
        da = mfma32(xf.x, wreg[4 * c + 0], da);
        da = mfma32(xf.y, wreg[4 * c + 1], da);
        da = mfma32(xf.z, wreg[4 * c + 2], da);
        da = mfma32(xf.w, wreg[4 * c + 3], da);
      }
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int li = acc_row(r, hi);
      const float eo = fast_exp2(s[r] - ls_o);           // owner-side softmax factor
      const float el = fast_exp2(s[r] - Ll[cur][li]);    // loop-side softmax factor
      if (p.single) {      // dS = A dA - A rho_row,  A = row softmax; the row side is the owner (swap == 0) or the loop tile
        const float a_ = p.swap ? el : eo;
        s[r] = a_ * (da[r] - (p.swap ? Ll[cur][32 + li] : g_o));
      } else {
        s[r] = 2.0f * eo * el * da[r] - eo * g_o - el * Ll[cur][32 + li];
      }
    }
    if (p.ds) {      // (owner = query pass only) TILED: tile (loop block t, owner block) = this wave's register image [16 r][64 lanes],
      // 4 KB contiguous, rows = loop index j, columns = owners i: fully coalesced 256-byte stores; rp_ds_matmul reads it (attention.hip)
      const long long tile = (zh * (NTOK / 32) + t) * (NTOK / 32) + (o0 >> 5);
      if (BF) store_acc_image_bf16(reinterpret_cast<unsigned short*>(p.ds) + tile * 1024, s, p.scale, lane);
      else if (LS) {      // element (key j = row, query i = lane) at chunk (j >> 2) * 32 + i, dword j & 3 (attention.hip: store_tile_runs)
#pragma unroll
        for (int g = 0; g < 4; ++g) st4(p.ds + tile * 1024 + 4 * (64 * g + lane), make_float4(s[4 * g], s[4 * g + 1], s[4 * g + 2], s[4 * g + 3]));
      } else store_acc_image_lds<4>(p.ds + tile * 1024, Tst[(BF || LS) ? 0 : wave], s, p.scale, lane);
    }
    // d owner^T[d][owner] += sum_loop other[loop][d] dS^T[loop][owner]
    if (BF) {
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        float a0[8], a1[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float* kr = Ks[cur] + acc_row(8 * g + j, hi) * KST + l31;
          a0[j] = kr[0];
          a1[j] = kr[32];
        }
        const bf16x8 pb = pack8(s[8 * g], s[8 * g + 1], s[8 * g + 2], s[8 * g + 3], s[8 * g + 4], s[8 * g + 5], s[8 * g + 6], s[8 * g + 7]);
        d0 = mfma_bf(pack8(a0), pb, d0);
        d1 = mfma_bf(pack8(a1), pb, d1);
      }
    } else {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float* kr = Ks[cur] + acc_row(r, hi) * KST + l31;
      d0 = mfma32(kr[0], s[r], d0);
      d1 = mfma32(kr[32], s[r], d1);
    }
    }
    if (t + 1 < NTILE) {
      kv_sstore<GT>(Ks[cur ^ 1], tid, kpre);
      x_sstore(Xs[cur ^ 1]);
      if (tid < 64) Ll[cur ^ 1][tid] = lpre;
    }
    __syncthreads();
  };
  for (int t = 0; t < NTILE; t += 2) {      // (two register sets for the loaded score tiles rotate without copies)
    step(sa, sb, t);
    step(sb, sa, t + 1);
  }
  float* orow = p.dqkv + ((long long)own_img * NTOK + o0 + l31) * p.ld + own_col;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const float om = LS ? 1.0f : p.scale;
    st4(orow + 8 * g + 4 * hi, make_float4(d0[4 * g] * om, d0[4 * g + 1] * om, d0[4 * g + 2] * om, d0[4 * g + 3] * om));
    st4(orow + 32 + 8 * g + 4 * hi, make_float4(d1[4 * g] * om, d1[4 * g + 1] * om, d1[4 * g + 2] * om, d1[4 * g + 3] * om));
  }
}

}  // namespace

extern "C" int rp_emm_apply(const float* qkv, int ldqkv, const float* x, const float* x_left, const float* rlse,
                            const float* clse, const float* s_in, float* t_out, float* f_part, int Z, int H, float scale, int swap,
                            int single, int bf16, void* stream) {
  if (Z <= 0 || (Z & 1) || H <= 0 || (ldqkv & 3)) return RP_EBADSHAPE;
  if (swap && f_part) return RP_EUNSUPPORTED;
  if (s_in && bf16) return RP_EUNSUPPORTED;
  EmmP p{};
  p.qkv = qkv; p.ld = ldqkv; p.x = x; p.rlse = rlse; p.clse = clse; p.t_out = t_out; p.f_part = f_part;
  p.H = H; p.scale = scale; p.swap = swap ? 1 : 0; p.ZH = Z * H; p.single = single ? 1 : 0; p.x_left = x_left; p.s_in = s_in;
  if (s_in) {
    if (swap) hipLaunchKernelGGL(emm_apply_s_kernel<true>, dim3(xcd_grid(NTILE / NW, Z * H)), dim3(NT), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL(emm_apply_s_kernel<false>, dim3(xcd_grid(NTILE / NW, Z * H)), dim3(NT), 0, (hipStream_t)stream, p);
    RP_CHECK_LAUNCH();
    return RP_OK;
  }
  if (bf16) hipLaunchKernelGGL(emm_apply_kernel<true>, dim3(xcd_grid(NTILE / NW, Z * H)), dim3(NT), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(emm_apply_kernel<false>, dim3(xcd_grid(NTILE / NW, Z * H)), dim3(NT), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

static int emm_grad_impl(const float* qkv, int ldqkv, const float* x, const float* w, const float* rlse, const float* clse,
                         const float* rho, const float* gamma, float* dqkv, float* ds, int Z, int H, float scale, int swap,
                         int single, int bf16, void* stream, const float* s_in = nullptr) {
  if (Z <= 0 || (Z & 1) || H <= 0 || (ldqkv & 3)) return RP_EBADSHAPE;
  if (ds && swap) return RP_EUNSUPPORTED;
  if (s_in && (bf16 || swap)) return RP_EUNSUPPORTED;
  EmmP p{};
  p.qkv = qkv; p.ld = ldqkv; p.x = x; p.w = w; p.rlse = rlse; p.clse = clse; p.rho = rho; p.gamma = gamma;
  p.dqkv = dqkv; p.H = H; p.scale = scale; p.swap = swap ? 1 : 0; p.ZH = Z * H; p.single = single ? 1 : 0; p.ds = ds; p.s_in = s_in;
  if (s_in) hipLaunchKernelGGL((emm_grad_kernel<false, true>), dim3(xcd_grid(NTILE / GW, Z * H)), dim3(GT), 0, (hipStream_t)stream, p);
  else if (bf16) hipLaunchKernelGGL(emm_grad_kernel<true>, dim3(xcd_grid(NTILE / GW, Z * H)), dim3(GT), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(emm_grad_kernel<false>, dim3(xcd_grid(NTILE / GW, Z * H)), dim3(GT), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_emm_grad(const float* qkv, int ldqkv, const float* x, const float* w, const float* rlse,
                           const float* clse, const float* rho, const float* gamma, float* dqkv, int Z, int H,
                           float scale, int swap, int single, int bf16, void* stream) {
  return emm_grad_impl(qkv, ldqkv, x, w, rlse, clse, rho, gamma, dqkv, nullptr, Z, H, scale, swap, single, bf16, stream);
}

extern "C" int rp_emm_grad_ds(const float* qkv, int ldqkv, const float* x, const float* w, const float* rlse,
                              const float* clse, const float* rho, const float* gamma, const float* s_in, float* dqkv, float* ds, int Z,
                              int H, float scale, int single, int bf16, void* stream) {
  if (!ds) return RP_EBADSHAPE;
  return emm_grad_impl(qkv, ldqkv, x, w, rlse, clse, rho, gamma, dqkv, ds, Z, H, scale, 0, single, bf16, stream, s_in);
}
