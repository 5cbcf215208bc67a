// emm_bf16.hip -- the Essential Matrix Module on the bf16 DATA PATH (BASELINE.json configs[4]).
//
// Reference: CrossAttention.forward, ess branch (src/modules/vision_transformer.py:198-223) and its autograd, per image z of a pair
// (partner z^1) and head h -- the algebra of emm.hip / DESIGN.md section 5:
//     S = scale q_{z^1} k_z^T;  A = rowsoftmax(S) o colsoftmax(S) = exp2(2 S cs - rlse2_i - clse2_j);  X = [v_z | pos | 0]  (576 x 96)
//     T = A X;  F = X^T T;      backward: W = X dF, W' = X dF^T, U = A^T X, rho = <W, T>, gamma = <W', U>,
//     dX = T dF^T + U dF,  dA = W X^T,  dS = 2 A dA - R rho_i - C gamma_j,  dq = scale dS k,  dk = scale dS^T q.
// Here q | k | v, X, T, U, W, W' live in HBM as bf16 and every contraction runs on v_mfma_f32_32x32x16_bf16 from LDS-DMA tiles and
// ds_read_b128 / ds_read_b64_tr_b16 operands (bf16_path.h), fp32 accumulation; the normalisers (log2 units, from rp_attn_fwd_bf16's
// statistics form), rho, gamma and dF stay fp32.  The 576 x 576 matrices never exist: every pass recomputes S tile by tile.
//   rp_emm_build_x_bf16   X from the bf16 qkv and the fp32 positional features
//   rp_emm_apply_bf16     T = A X (owner = query rows) or U = A^T X (swap: owner = key rows), stored transposed-accumulated like
//                         the attention output: T^T[c][owner] += X^T P, so the accumulators leave as whole bf16 rows
//   rp_emm_f_bf16         F = X^T T per (z, h): both operands by transpose reads (rows are the contraction index)
//   rp_emm_w_bf16         W, W' (+ rho from the fp32 accumulators of W)
//   rp_emm_dx_bf16        dX -> the v columns of dqkv, gamma
//   rp_emm_grad_bf16      dq (owner = query rows) / dk (swap) -> the q / k columns of dqkv; recompute form, no stored dS
#include "bf16_path.h"
#include "../../include/relpose_hip.h"

namespace {
using namespace bf16path;

constexpr int XW = 96;                 // padded width of X / T / U / W / W' rows (70 live columns)

// transposed accumulators of a 32-row x 96-column tile (lane = row l31, register r of block nb = column 32 nb + acc_row(r, hi)), times
// mul, -> bf16 rows of dst (row stride ld elements) through 6 KB of this wave's LDS.  Rows are 12 chunks of 16 B; chunk ^ ((row >> 2) & 3)
// spreads the 8-byte writes of 16 consecutive rows (192-byte rows alias every 4) over the banks.
RP_DEV void store_ownerT96_bf16(bf16_t* Os, bf16_t* dst, int ld, int lane, const f32x16 (&o)[3], float mul) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int nb = 0; nb < 3; ++nb)
#pragma unroll
    for (int gq = 0; gq < 4; ++gq) {
      const uint2 w = make_uint2(pk_bf16(o[nb][4 * gq] * mul, o[nb][4 * gq + 1] * mul), pk_bf16(o[nb][4 * gq + 2] * mul, o[nb][4 * gq + 3] * mul));
      *reinterpret_cast<uint2*>(Os + l31 * XW + (((4 * nb + gq) ^ ((l31 >> 2) & 3)) << 3) + 4 * hi) = w;
    }
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int f = lane + 64 * i, row = f / 12, ch = f % 12;
    const uint4 w = *reinterpret_cast<const uint4*>(Os + row * XW + ((ch ^ ((row >> 2) & 3)) << 3));
    *reinterpret_cast<uint4*>(dst + (long long)row * ld + 8 * ch) = w;
  }
}

// ------------------------------------------------------------------------------------------------ X = [v | pos | 0]
__global__ __launch_bounds__(256) void emm_build_x_bf16_kernel(const bf16_t* __restrict__ qkv, const float* __restrict__ pos,
                                                               bf16_t* __restrict__ x, int H, int ld, long long total) {
  const long long e = (long long)blockIdx.x * 256 + threadIdx.x;          // one 16-byte chunk (8 columns) per thread
  if (e >= total) return;
  const int ch = (int)(e % 12);
  const long long row = e / 12;                                            // (z H + h) 576 + n
  const int n = (int)(row % NTOK), h = (int)((row / NTOK) % H);
  const long long z = row / ((long long)NTOK * H);
  uint4 w = make_uint4(0u, 0u, 0u, 0u);
  if (ch < 8) w = *reinterpret_cast<const uint4*>(qkv + (z * NTOK + n) * ld + 384 + h * 64 + 8 * ch);
  else if (ch == 8) {
    const float* pr = pos + ((z >> 1) * NTOK + n) * 6;
    w = make_uint4(pk_bf16(pr[0], pr[1]), pk_bf16(pr[2], pr[3]), pk_bf16(pr[4], pr[5]), 0u);
  }
  *reinterpret_cast<uint4*>(x + row * XW + 8 * ch) = w;
}

// ------------------------------------------------------------------------------------------------ T = A X / U = A^T X
struct EmmBfP {
  const bf16_t* qkv; int ld;
  const bf16_t* x; const bf16_t* w;
  const float* rlse; const float* clse; const float* rho; const float* gamma;
  bf16_t* t_out; bf16_t* dqkv;
  int H; float scale; int swap; int ZH;
};

// stage of SL loop rows: L rows (64 d, swizzle SWL) + X rows (96 columns, natural layout or swz_x) by DMA; 10 pieces of 1 KB per 32 rows
RP_DEV int swz_x(int r) { return (r >> 2) & 3; }       // X tiles read by rows (ds_read_b128, lane = row): 192-byte rows alias every 4

template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void emm_apply_bf16_kernel(EmmBfP p) {
  constexpr int SL = 32 * NW / 2;                  // loop rows per stage (NW = 2: 32): every wave moves 5 pieces per 32 rows ... see below
  static_assert(NW == 2, "the DMA plan below is written for 2-wave workgroups (10 pieces per 32-row stage, 5 per wave)");
  __shared__ __attribute__((aligned(16))) bf16_t Ls[2][32 * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][32 * XW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  int zh_, blk;
  if (!xcd_problem(18 / NW, p.ZH, zh_, blk)) return;
  const int h = zh_ % p.H, z = zh_ / p.H;
  const int o0 = (blk * NW + wave) * 32;
  const int own_img = p.swap ? z : (z ^ 1), own_col = (p.swap ? 192 : 0) + h * 64;
  const int loop_img = p.swap ? (z ^ 1) : z, loop_col = (p.swap ? 0 : 192) + h * 64;
  const long long zh = (long long)z * p.H + h;
  const float* own_lse = (p.swap ? p.clse : p.rlse) + zh * NTOK;
  const float* loop_lse = (p.swap ? p.rlse : p.clse) + zh * NTOK;
  const bf16_t* lb = p.qkv + (long long)loop_img * NTOK * p.ld + loop_col;
  const bf16_t* xb = p.x + zh * NTOK * XW;

  // DMA plan per 32-row stage: L tile = 4 pieces (8 rows of 128 B each), X tile = 6 pieces (lane-linear over [32][192 B]).
  // wave 0: L pieces 0, 1 + X pieces 0, 1, 2;  wave 1: L pieces 2, 3 + X pieces 3, 4, 5
  unsigned lvoff[2], xvoff[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (2 * wave + i) * 8 + (lane >> 3);
    lvoff[i] = (unsigned)(r * p.ld + (((lane & 7) ^ swz_k(r)) << 3)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) xvoff[i] = (unsigned)((3 * wave + i) * 1024 + lane * 16);          // X rows are contiguous: a plain copy
  const unsigned ls0 = lds_addr_of(&Ls[0][0]) + wave * 2048, xs0 = lds_addr_of(&Xs[0][0]) + wave * 3072;
  auto issue = [&](int s, int buf) {
    const void* ll = uniform_vptr(lb + (long long)s * 32 * p.ld);
    const void* xx = uniform_vptr(xb + (long long)s * 32 * XW);
    glds16b(ll, lvoff[0], ls0 + buf * 4096);
    glds16b(ll, lvoff[1], ls0 + buf * 4096 + 1024);
#pragma unroll
    for (int i = 0; i < 3; ++i) glds16b(xx, xvoff[i], xs0 + buf * 6144 + i * 1024);
  };
  issue(0, 0);

  bf16x8 opk[4];
  {
    const bf16_t* orow = p.qkv + ((long long)own_img * NTOK + o0 + l31) * p.ld + own_col + 8 * hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) opk[c] = *reinterpret_cast<const bf16x8*>(orow + 16 * c);
  }
  const float ls_o = own_lse[o0 + l31];
  const float cs2 = 2.0f * p.scale * RP_LOG2E;
  int koff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) koff[c] = l31 * 64 + (((2 * c + hi) ^ swz_k(l31)) << 3);
  // transpose reads of the natural-layout X tile: rows 16 g + 8 half + 4 hi + (t16 >> 2), columns 32 nb + 16 gg + 4 (t16 & 3)
  const int t16 = lane & 15, gg = (lane >> 4) & 1;
  const int xoff = (4 * hi + (t16 >> 2)) * XW + 16 * gg + 4 * (t16 & 3);

  f32x16 tacc[3] = {zero16(), zero16(), zero16()};
  for (int s = 0; s < NTOK / 32; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < NTOK / 32) issue(s + 1, buf ^ 1);
    const bf16_t* Lt = Ls[buf];
    const bf16_t* Xt = Xs[buf] + xoff;
    // loop-side normaliser of register r = loop row acc_row(r, hi): four runs of 4 consecutive rows
    const float* lq = loop_lse + s * 32 + 4 * hi;
    float4 l4[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) l4[g4] = ld4(lq + 8 * g4);
    f32x16 sc = zero16();
#pragma unroll
    for (int c = 0; c < 4; ++c) sc = mfma_bf(ld_bf16x8_lds(Lt + koff[c]), opk[c], sc);        // S^T[loop][owner]
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float lv[4] = {l4[g4].x, l4[g4].y, l4[g4].z, l4[g4].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) sc[4 * g4 + e] = fast_exp2(fmaf(sc[4 * g4 + e], cs2, -(ls_o + lv[e])));     // A = exp2(2 S - rlse - clse)
    }
    const bf16x8 p0 = pack8(sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[7]);
    const bf16x8 p1 = pack8(sc[8], sc[9], sc[10], sc[11], sc[12], sc[13], sc[14], sc[15]);
    // T^T[c][owner] += sum_loop X[loop][c] A[owner][loop]: A operand = transpose reads of X, B operand = the packed accumulators
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
      tacc[nb] = mfma_bf(tr_operand(Xt + 32 * nb, Xt + 8 * XW + 32 * nb), p0, tacc[nb]);
      tacc[nb] = mfma_bf(tr_operand(Xt + 16 * XW + 32 * nb, Xt + 24 * XW + 32 * nb), p1, tacc[nb]);
    }
  }
  __builtin_amdgcn_s_barrier();
  store_ownerT96_bf16(&Xs[0][0] + wave * 3072, p.t_out + (zh * NTOK + o0) * XW, XW, lane, tacc, 1.0f);
}

// ------------------------------------------------------------------------------------------------ F = X^T T
// one workgroup of 3 waves per (z, h): wave w owns rows a = 32 w .. of F (all 96 columns); 64-row stages of X and T by DMA
__global__ __launch_bounds__(192, 2) void emm_f_bf16_kernel(const bf16_t* __restrict__ x, const bf16_t* __restrict__ t, float* __restrict__ f, int ZH) {
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][64 * XW];
  __shared__ __attribute__((aligned(16))) bf16_t Ts[2][64 * XW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const long long zh = blockIdx.x;
  const bf16_t* xb = x + zh * NTOK * XW;
  const bf16_t* tb = t + zh * NTOK * XW;
  // 64 rows x 192 B = 12 pieces per operand and stage, 4 per wave; plain copies (natural layout: 192-byte rows are skewed by themselves)
  const unsigned xs0 = lds_addr_of(&Xs[0][0]) + wave * 4096, ts0 = lds_addr_of(&Ts[0][0]) + wave * 4096;
  auto issue = [&](int s, int buf) {
    const void* xx = uniform_vptr(xb + (long long)s * 64 * XW);
    const void* tt = uniform_vptr(tb + (long long)s * 64 * XW);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      glds16b(xx, (unsigned)((4 * wave + i) * 1024 + lane * 16), xs0 + buf * 12288 + i * 1024);
      glds16b(tt, (unsigned)((4 * wave + i) * 1024 + lane * 16), ts0 + buf * 12288 + i * 1024);
    }
  };
  issue(0, 0);
  const int t16 = lane & 15, gg = (lane >> 4) & 1;
  const int off = (8 * hi + (t16 >> 2)) * XW + 16 * gg + 4 * (t16 & 3);        // k-slot (hi, j) <-> row 16 step + 8 hi + j
  f32x16 acc[3] = {zero16(), zero16(), zero16()};
  for (int s = 0; s < NTOK / 64; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < NTOK / 64) issue(s + 1, buf ^ 1);
#pragma unroll
    for (int st = 0; st < 4; ++st) {
      const bf16_t* Xt = Xs[buf] + st * 16 * XW + off;
      const bf16_t* Tt = Ts[buf] + st * 16 * XW + off;
      const bf16x8 a = tr_operand(Xt + 32 * wave, Xt + 4 * XW + 32 * wave);
#pragma unroll
      for (int cb = 0; cb < 3; ++cb) acc[cb] = mfma_bf(a, tr_operand(Tt + 32 * cb, Tt + 4 * XW + 32 * cb), acc[cb]);
    }
  }
  float* fo = f + zh * XW * XW + (long long)(32 * wave) * XW + l31;              // F[a][c]: lane = c, register = a
#pragma unroll
  for (int cb = 0; cb < 3; ++cb)
#pragma unroll
    for (int r = 0; r < 16; ++r) fo[(long long)acc_row(r, hi) * XW + 32 * cb] = acc[cb][r];
}

// ------------------------------------------------------------------------------------------------ W, W', rho / dX, gamma
// dF (fp32 [96][96] per problem) staged in LDS as bf16, as it lies AND transposed: every product below contracts over a ROW index of
// one of the two images, i.e. takes its MFMA A operand by transpose reads (lane = output column).  MODE 0: W^T = dF^T-contraction
// (W[i][c] = sum_a X[i][a] dF[a][c]), W'^T (W'[i][a] = sum_c X[i][c] dF[a][c]), rho_i = <W_i, T_i> from the fp32 accumulators.
// MODE 1: dX[i][e] = sum_c T[i][c] dF[e][c] + sum_a U[i][a] dF[a][e] (e < 64) -> dqkv's v columns; gamma_i = <W'_i, U_i>.
struct EmmSmallP {
  const bf16_t* a0; const bf16_t* a1; const bf16_t* a2;    // MODE 0: X, T, -;   MODE 1: T, U, W'
  const float* df;
  bf16_t* o0; bf16_t* o1;                                   // MODE 0: W, W';     MODE 1: dqkv (v columns of image z), -
  float* dot;                                               // MODE 0: rho;       MODE 1: gamma
  int H, ld, ZH;
};

template <int MODE>
__global__ __launch_bounds__(384, 2) void emm_small_bf16_kernel(EmmSmallP p) {
  __shared__ __attribute__((aligned(16))) bf16_t D[XW * XW];        // dF[a][c]
  __shared__ __attribute__((aligned(16))) bf16_t Dt[XW * XW];       // dF^T[c][a]
  __shared__ __attribute__((aligned(16))) bf16_t Os[6][32 * XW];    // per-wave output staging
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int zh_ = blockIdx.x / 3, part = blockIdx.x % 3;             // 3 workgroups of 6 waves x 32 rows per (z, h)
  const long long zh = zh_;
  const int h = zh_ % p.H, z = zh_ / p.H;
  const float* dfp = p.df + zh * XW * XW;
  for (int e = tid; e < (XW / 4) * (XW / 4); e += 384) {      // one 4 x 4 block per thread: 8-byte writes into both images
    const int a = 4 * (e / (XW / 4)), c = 4 * (e % (XW / 4));
    float4 v[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) v[k] = ld4(dfp + (a + k) * XW + c);
#pragma unroll
    for (int k = 0; k < 4; ++k) *reinterpret_cast<uint2*>(D + (a + k) * XW + c) = make_uint2(pk_bf16(v[k].x, v[k].y), pk_bf16(v[k].z, v[k].w));
    *reinterpret_cast<uint2*>(Dt + (c + 0) * XW + a) = make_uint2(pk_bf16(v[0].x, v[1].x), pk_bf16(v[2].x, v[3].x));
    *reinterpret_cast<uint2*>(Dt + (c + 1) * XW + a) = make_uint2(pk_bf16(v[0].y, v[1].y), pk_bf16(v[2].y, v[3].y));
    *reinterpret_cast<uint2*>(Dt + (c + 2) * XW + a) = make_uint2(pk_bf16(v[0].z, v[1].z), pk_bf16(v[2].z, v[3].z));
    *reinterpret_cast<uint2*>(Dt + (c + 3) * XW + a) = make_uint2(pk_bf16(v[0].w, v[1].w), pk_bf16(v[2].w, v[3].w));
  }
  __syncthreads();
  const int i0 = (part * 6 + wave) * 32;
  const long long rbase = (zh * NTOK + i0 + l31) * XW + 8 * hi;     // lane (row l31, half hi): columns 16 s + 8 hi .. + 7 of its row
  const int t16 = lane & 15, gg = (lane >> 4) & 1;
  const int off = (8 * hi + (t16 >> 2)) * XW + 16 * gg + 4 * (t16 & 3);
  bf16x8 ra[6], rb[6];
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    ra[s] = *reinterpret_cast<const bf16x8*>(p.a0 + rbase + 16 * s);
    if (MODE == 1) rb[s] = *reinterpret_cast<const bf16x8*>(p.a1 + rbase + 16 * s);
  }
  if (MODE == 0) {
    f32x16 w[3] = {zero16(), zero16(), zero16()}, wp[3] = {zero16(), zero16(), zero16()};
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
      for (int nb = 0; nb < 3; ++nb) {
        // W^T[c][i] += sum_a dF[a][c] X[i][a]: rows a of D;   W'^T[a'][i] += sum_c dF^T[c][a'] X[i][c]: rows c of Dt
        w[nb] = mfma_bf(tr_operand(D + s * 16 * XW + off + 32 * nb, D + s * 16 * XW + 4 * XW + off + 32 * nb), ra[s], w[nb]);
        wp[nb] = mfma_bf(tr_operand(Dt + s * 16 * XW + off + 32 * nb, Dt + s * 16 * XW + 4 * XW + off + 32 * nb), ra[s], wp[nb]);
      }
    // rho_i = sum_c W[i][c] T[i][c] from the fp32 accumulators: register r of block nb is column 32 nb + acc_row(r, hi) -> runs of 4
    float rho = 0.f;
    const bf16_t* trow = p.a1 + (zh * NTOK + i0 + l31) * XW + 4 * hi;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
      for (int gq = 0; gq < 4; ++gq) {
        const float4 tv = widen4(*reinterpret_cast<const uint2*>(trow + 32 * nb + 8 * gq));
        rho = fmaf(w[nb][4 * gq], tv.x, rho); rho = fmaf(w[nb][4 * gq + 1], tv.y, rho);
        rho = fmaf(w[nb][4 * gq + 2], tv.z, rho); rho = fmaf(w[nb][4 * gq + 3], tv.w, rho);
      }
    rho += __shfl_xor(rho, 32, 64);
    if (hi == 0) p.dot[zh * NTOK + i0 + l31] = rho;
    store_ownerT96_bf16(Os[wave], p.o0 + (zh * NTOK + i0) * XW, XW, lane, w, 1.0f);
    store_ownerT96_bf16(Os[wave], p.o1 + (zh * NTOK + i0) * XW, XW, lane, wp, 1.0f);
  } else {
    f32x16 dx[3] = {zero16(), zero16(), zero16()};          // (block 2 unused: the positional / pad columns carry no gradient)
#pragma unroll
    for (int s = 0; s < 6; ++s)
#pragma unroll
      for (int nb = 0; nb < 2; ++nb) {
        // dX^T[e][i] += sum_c dF^T[c][e] T[i][c]  +  sum_a dF[a][e] U[i][a]
        dx[nb] = mfma_bf(tr_operand(Dt + s * 16 * XW + off + 32 * nb, Dt + s * 16 * XW + 4 * XW + off + 32 * nb), ra[s], dx[nb]);
        dx[nb] = mfma_bf(tr_operand(D + s * 16 * XW + off + 32 * nb, D + s * 16 * XW + 4 * XW + off + 32 * nb), rb[s], dx[nb]);
      }
    // gamma_i = sum_a W'[i][a] U[i][a]: this lane's 48 columns of both rows
    float gam = 0.f;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      const bf16x8 wv = *reinterpret_cast<const bf16x8*>(p.a2 + rbase + 16 * s);
      typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
      const u32x4_t a = __builtin_bit_cast(u32x4_t, wv), b = __builtin_bit_cast(u32x4_t, rb[s]);
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        gam = fmaf(__uint_as_float(a[k] << 16), __uint_as_float(b[k] << 16), gam);
        gam = fmaf(__uint_as_float(a[k] & 0xffff0000u), __uint_as_float(b[k] & 0xffff0000u), gam);
      }
    }
    gam += __shfl_xor(gam, 32, 64);
    if (hi == 0) p.dot[zh * NTOK + i0 + l31] = gam;
    store_ownerT_bf16(Os[wave], p.o0 + ((long long)z * NTOK + i0) * p.ld + 384 + h * 64, p.ld, lane, dx[0], dx[1], 1.0f);
  }
}

// ------------------------------------------------------------------------------------------------ dq / dk
template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void emm_grad_bf16_kernel(EmmBfP p) {
  static_assert(NW == 2, "DMA plan written for 2-wave workgroups");
  __shared__ __attribute__((aligned(16))) bf16_t Ls[2][32 * 64];
  __shared__ __attribute__((aligned(16))) bf16_t Xs[2][32 * XW];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  int zh_, blk;
  if (!xcd_problem(18 / NW, p.ZH, zh_, blk)) return;
  const int h = zh_ % p.H, z = zh_ / p.H;
  const int o0 = (blk * NW + wave) * 32;
  const int own_img = p.swap ? z : (z ^ 1), own_col = (p.swap ? 192 : 0) + h * 64;
  const int loop_img = p.swap ? (z ^ 1) : z, loop_col = (p.swap ? 0 : 192) + h * 64;
  const long long zh = (long long)z * p.H + h;
  const float* own_lse = (p.swap ? p.clse : p.rlse) + zh * NTOK;
  const float* loop_lse = (p.swap ? p.rlse : p.clse) + zh * NTOK;
  const float* own_g = (p.swap ? p.gamma : p.rho) + zh * NTOK;
  const float* loop_g = (p.swap ? p.rho : p.gamma) + zh * NTOK;
  const bf16_t* lb = p.qkv + (long long)loop_img * NTOK * p.ld + loop_col;
  const bf16_t* xb = p.x + zh * NTOK * XW;

  // L tile: swz_d (read by rows for S and by transpose reads for the owner gradient); X tile: swz_x (read by rows for dA)
  unsigned lvoff[2], xvoff[3];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (2 * wave + i) * 8 + (lane >> 3);
    lvoff[i] = (unsigned)(r * p.ld + (((lane & 7) ^ swz_d(r)) << 3)) * 2u;
  }
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int byte = (3 * wave + i) * 1024 + lane * 16, r = byte / 192, slot = (byte % 192) >> 4;
    xvoff[i] = (unsigned)(r * 192 + ((slot ^ swz_x(r)) << 4));
  }
  const unsigned ls0 = lds_addr_of(&Ls[0][0]) + wave * 2048, xs0 = lds_addr_of(&Xs[0][0]) + wave * 3072;
  auto issue = [&](int s, int buf) {
    const void* ll = uniform_vptr(lb + (long long)s * 32 * p.ld);
    const void* xx = uniform_vptr(xb + (long long)s * 32 * XW);
    glds16b(ll, lvoff[0], ls0 + buf * 4096);
    glds16b(ll, lvoff[1], ls0 + buf * 4096 + 1024);
#pragma unroll
    for (int i = 0; i < 3; ++i) glds16b(xx, xvoff[i], xs0 + buf * 6144 + i * 1024);
  };
  issue(0, 0);

  bf16x8 opk[4], wpk[6];
  {
    const bf16_t* orow = p.qkv + ((long long)own_img * NTOK + o0 + l31) * p.ld + own_col + 8 * hi;
    const bf16_t* wrow = p.w + (zh * NTOK + o0 + l31) * XW + 8 * hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) opk[c] = *reinterpret_cast<const bf16x8*>(orow + 16 * c);
#pragma unroll
    for (int c = 0; c < 6; ++c) wpk[c] = *reinterpret_cast<const bf16x8*>(wrow + 16 * c);
  }
  const float nls_o = -own_lse[o0 + l31], g_o = own_g[o0 + l31];
  const float cs = p.scale * RP_LOG2E;
  int koff[4], xoff[6];
#pragma unroll
  for (int c = 0; c < 4; ++c) koff[c] = l31 * 64 + (((2 * c + hi) ^ swz_d(l31)) << 3);
#pragma unroll
  for (int c = 0; c < 6; ++c) xoff[c] = l31 * XW + (((2 * c + hi) ^ swz_x(l31)) << 3);
  int toff[2][2];
  tr_offsets_d(lane, toff);

  f32x16 d0 = zero16(), d1 = zero16();
  for (int s = 0; s < NTOK / 32; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < NTOK / 32) issue(s + 1, buf ^ 1);
    const bf16_t* Lt = Ls[buf];
    const bf16_t* Xt = Xs[buf];
    const float* lq = loop_lse + s * 32 + 4 * hi;
    const float* gq_ = loop_g + s * 32 + 4 * hi;
    float4 l4[4], g4v[4];
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) { l4[g4] = ld4(lq + 8 * g4); g4v[g4] = ld4(gq_ + 8 * g4); }
    f32x16 sc = zero16(), da = zero16();
#pragma unroll
    for (int c = 0; c < 4; ++c) sc = mfma_bf(ld_bf16x8_lds(Lt + koff[c]), opk[c], sc);        // S^T[loop][owner]
#pragma unroll
    for (int c = 0; c < 6; ++c) da = mfma_bf(ld_bf16x8_lds(Xt + xoff[c]), wpk[c], da);        // dA^T[loop][owner] = sum_c X[loop][c] W[owner][c]
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
      const float lv[4] = {l4[g4].x, l4[g4].y, l4[g4].z, l4[g4].w};
      const float gv[4] = {g4v[g4].x, g4v[g4].y, g4v[g4].z, g4v[g4].w};
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = 4 * g4 + e;
        const float eo = fast_exp2(fmaf(sc[r], cs, nls_o));            // owner-side softmax factor
        const float el = fast_exp2(fmaf(sc[r], cs, -lv[e]));           // loop-side softmax factor
        sc[r] = fmaf(2.0f * eo * el, da[r], -fmaf(eo, g_o, el * gv[e]));      // dS = 2 A dA - R rho - C gamma (both row/column roles)
      }
    }
    const bf16x8 s0 = pack8(sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[7]);
    const bf16x8 s1 = pack8(sc[8], sc[9], sc[10], sc[11], sc[12], sc[13], sc[14], sc[15]);
    // d owner^T[d][owner] += sum_loop other[loop][d] dS^T[loop][owner]
    d0 = mfma_bf(tr_operand(Lt + toff[0][0], Lt + toff[0][1]), s0, d0);
    d1 = mfma_bf(tr_operand(Lt + toff[1][0], Lt + toff[1][1]), s0, d1);
    d0 = mfma_bf(tr_operand(Lt + 1024 + toff[0][0], Lt + 1024 + toff[0][1]), s1, d0);
    d1 = mfma_bf(tr_operand(Lt + 1024 + toff[1][0], Lt + 1024 + toff[1][1]), s1, d1);
  }
  __builtin_amdgcn_s_barrier();
  store_ownerT_bf16(&Xs[0][0] + wave * 2048, p.dqkv + ((long long)own_img * NTOK + o0) * p.ld + own_col, p.ld, lane, d0, d1, p.scale);
}

}  // namespace

extern "C" int rp_emm_build_x_bf16(const void* qkv, const float* pos, void* x, int Z, int H, int ldqkv, void* stream) {
  if (!qkv || !pos || !x || Z <= 0 || (Z & 1) || H <= 0 || (ldqkv & 7)) return RP_EBADSHAPE;
  const long long total = (long long)Z * H * NTOK * 12;
  hipLaunchKernelGGL(emm_build_x_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)qkv, pos,
                     (bf16_t*)x, H, ldqkv, total);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_emm_apply_bf16(const void* qkv, int ldqkv, const void* x, const float* rlse2, const float* clse2, void* t_out, int Z, int H,
                                 float scale, int swap, void* stream) {
  if (!qkv || !x || !rlse2 || !clse2 || !t_out || Z <= 0 || (Z & 1) || H <= 0 || (ldqkv & 7)) return RP_EBADSHAPE;
  EmmBfP p{};
  p.qkv = (const bf16_t*)qkv; p.ld = ldqkv; p.x = (const bf16_t*)x; p.rlse = rlse2; p.clse = clse2; p.t_out = (bf16_t*)t_out;
  p.H = H; p.scale = scale; p.swap = swap ? 1 : 0; p.ZH = Z * H;
  hipLaunchKernelGGL(emm_apply_bf16_kernel<2>, dim3(xcd_grid(9, Z * H)), dim3(128), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_emm_f_bf16(const void* x, const void* t, float* f, int Z, int H, void* stream) {
  if (!x || !t || !f || Z <= 0 || (Z & 1) || H <= 0) return RP_EBADSHAPE;   // images come in pairs, as in rp_emm_apply_bf16
  hipLaunchKernelGGL(emm_f_bf16_kernel, dim3(Z * H), dim3(192), 0, (hipStream_t)stream, (const bf16_t*)x, (const bf16_t*)t, f, Z * H);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_emm_w_bf16(const void* x, const void* t, const float* df, void* w, void* wp, float* rho, int Z, int H, void* stream) {
  if (!x || !t || !df || !w || !wp || !rho || Z <= 0 || (Z & 1) || H <= 0) return RP_EBADSHAPE;
  EmmSmallP p{(const bf16_t*)x, (const bf16_t*)t, nullptr, df, (bf16_t*)w, (bf16_t*)wp, rho, H, 0, Z * H};
  hipLaunchKernelGGL(emm_small_bf16_kernel<0>, dim3(Z * H * 3), dim3(384), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_emm_dx_bf16(const void* t, const void* u, const void* wp, const float* df, void* dqkv, int ldqkv, float* gamma, int Z, int H,
                              void* stream) {
  if (!t || !u || !wp || !df || !dqkv || !gamma || Z <= 0 || (Z & 1) || H <= 0 || (ldqkv & 7)) return RP_EBADSHAPE;
  EmmSmallP p{(const bf16_t*)t, (const bf16_t*)u, (const bf16_t*)wp, df, (bf16_t*)dqkv, nullptr, gamma, H, ldqkv, Z * H};
  hipLaunchKernelGGL(emm_small_bf16_kernel<1>, dim3(Z * H * 3), dim3(384), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_emm_grad_bf16(const void* qkv, int ldqkv, const void* x, const void* w, const float* rlse2, const float* clse2, const float* rho,
                                const float* gamma, void* dqkv, int Z, int H, float scale, int swap, void* stream) {
  if (!qkv || !x || !w || !rlse2 || !clse2 || !rho || !gamma || !dqkv || Z <= 0 || (Z & 1) || H <= 0 || (ldqkv & 7)) return RP_EBADSHAPE;
  EmmBfP p{};
  p.qkv = (const bf16_t*)qkv; p.ld = ldqkv; p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.rlse = rlse2; p.clse = clse2; p.rho = rho;
  p.gamma = gamma; p.dqkv = (bf16_t*)dqkv; p.H = H; p.scale = scale; p.swap = swap ? 1 : 0; p.ZH = Z * H;
  hipLaunchKernelGGL(emm_grad_bf16_kernel<2>, dim3(xcd_grid(9, Z * H)), dim3(128), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
