// attention.hip -- fused softmax attention for N = 576 tokens, head dim 64 (rp_attn_fwd / rp_attn_bwd).
//
// Replaces  attn = softmax(q k^T * scale); x = attn @ v  of reference Attention.forward
// (src/modules/vision_transformer.py:325-329) without materialising the 576x576 score matrix, and -- in
// stats_only mode -- produces the row / column log-sum-exp normalisers of the dual softmax of
// CrossAttention.forward (:205-206).
//
// Work split: one wave owns 32 query rows ("owner"); a workgroup of NW waves shares K/V tiles of 32 keys
// staged in LDS (double buffered, one barrier per tile, next tile's global loads in flight during the
// MFMAs of the current one).  Everything is computed TRANSPOSED so that all per-query softmax state is
// lane-local (lane & 31 = query, half-wave = which 16 of the 32 keys):
//   S^T[kv][q]  = sum_d K[kv][d] Q[q][d]      A = K (LDS, 8 x ds_read_b128 per lane), B = Q (32 VGPRs, pre-scaled)
//   O^T[d][q]  += sum_kv V[kv][d] P[q][kv]    A = V (LDS, ds_read_b32 rows),          B = P = the S^T accumulators
// i.e. the score accumulators feed the second MFMA directly as its B operand (register r of half-wave hi is
// key acc_row(r,hi)), no LDS round trip, no cross-lane traffic except one xor-32 shuffle for the row max.
// fp32 v_mfma_f32_32x32x2_f32 throughout: 64 MFMAs (4096 cycles) per 32x32 tile pair, exact fp32 products.
#include "common.h"
#include "../../include/relpose_hip.h"
#include <stdlib.h>
#include <type_traits>

namespace {

template <int OFF> RP_DEV float lds_rd32(unsigned addr) {      // ds_read_b32 with an immediate byte offset (invisible to hipcc's waitcnt pass)
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N, class F> RP_DEV void static_for(F&& f) {
  if constexpr (N > 0) {
    static_for<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

constexpr int NTOK = 576;
constexpr int KST = 68;   // LDS row stride (floats) for tiles read along d with ds_read_b128
constexpr int NTILE = NTOK / 32;

struct AttnP {
  const float* q; const float* k; const float* v;
  float* o; float* lse;
  int H, ldq, ldk, ldv, ldo, q_xor, k_xor;
  float scale;
  int ZH;
  float* colpart;   // COLS only: [ZH][18 owner blocks][576 loop rows][2] = (max, sum of exp2(. - max)) of each loop row over one owner block
  float* pst;       // SAVEP only: [ZH][18 query blocks][18 key tiles][32 queries][32 keys]: exp2(s - running max) of every tile, and
  float* mrun;      // [ZH][18 key tiles][576 queries]: the running max (log2 units) each tile was normalised with
};

// cooperative global -> register prefetch of a [32][64] tile (32 rows x 16 float4).  No exec-masked guards: when the
// thread count does not divide 512 the surplus threads of the last round re-load (and later re-store) element 511-ish
// duplicates -- a guarded load becomes its own basic block and hipcc then drains vmcnt(0) before every one of them.
template <int NT>
RP_DEV void tile_gload(const float* base, int ld, int tid, float4 (&r)[(512 + NT - 1) / NT]) {
#pragma unroll
  for (int j = 0; j < (512 + NT - 1) / NT; ++j) {
    int f = tid + NT * j;
    if (512 % NT != 0) f = min(f, 511);
    r[j] = ld4(base + (long long)(f >> 4) * ld + (f & 15) * 4);
  }
}
template <int NT, int STRIDE>
RP_DEV void tile_sstore(float* s, int tid, const float4 (&r)[(512 + NT - 1) / NT]) {
#pragma unroll
  for (int j = 0; j < (512 + NT - 1) / NT; ++j) {
    int f = tid + NT * j;
    if (512 % NT != 0) f = min(f, 511);
    st4(s + (f >> 4) * STRIDE + (f & 15) * 4, r[j]);
  }
}

// S^T tile: s[r] = sum_d Ks[kv = acc_row(r,hi)][d] * breg[q = l31][d]; breg[t] holds d = 32*hi + t (fp32 mode) / bpk[c] = the
// same 32 values as 4 x 8 bf16 (bf16 mode, see common.h)
template <bool BF>
RP_DEV f32x16 score_tile(const float* Ks, int l31, int hi, const float (&breg)[32], const bf16x8 (&bpk)[4]) {
  f32x16 s = zero16();
  const float* kr = Ks + l31 * KST + 32 * hi;
  if (BF) {
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

// acc^T[d][owner] += sum_t Ts[t = acc_row(r,hi)][d] * p[r]   for the two 32-wide d blocks (row-pattern reads)
template <int STRIDE, bool BF>
RP_DEV void accum_tile(const float* Ts, int l31, int hi, const f32x16& p, f32x16& o0, f32x16& o1) {
  if (BF) {
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float a0[8], a1[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float* vr = Ts + acc_row(8 * g + j, hi) * STRIDE + l31;
        a0[j] = vr[0];
        a1[j] = vr[32];
      }
      const bf16x8 pb = pack8(p[8 * g], p[8 * g + 1], p[8 * g + 2], p[8 * g + 3], p[8 * g + 4], p[8 * g + 5], p[8 * g + 6], p[8 * g + 7]);
      o0 = mfma_bf(pack8(a0), pb, o0);
      o1 = mfma_bf(pack8(a1), pb, o1);
    }
    return;
  }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* vr = Ts + acc_row(r, hi) * STRIDE + l31;
    o0 = mfma32(vr[0], p[r], o0);
    o1 = mfma32(vr[32], p[r], o1);
  }
}

// store acc^T (rows d, cols owner) to out[owner row][d] with optional scale; 4-float runs per register group
RP_DEV void store_ownerT(float* row_ptr, int hi, const f32x16& o0, const f32x16& o1, float mul) {
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    st4(row_ptr + 8 * g + 4 * hi, make_float4(o0[4 * g] * mul, o0[4 * g + 1] * mul, o0[4 * g + 2] * mul, o0[4 * g + 3] * mul));
    st4(row_ptr + 32 + 8 * g + 4 * hi, make_float4(o1[4 * g] * mul, o1[4 * g + 1] * mul, o1[4 * g + 2] * mul, o1[4 * g + 3] * mul));
  }
}

// column sums of the same acc^T tile over its 32 owner rows (x mul): part[d] for d = 0..63 -- a partial of the bias gradient of the
// Linear that produced the operand (qkv), so that gradient needs no pass of its own over the [tokens, 576] tensor.  Register r of a
// 32-lane half holds column d = acc_row(r, hi) (+32 for o1) of 32 different rows: DPP row sums + one cross-row exchange, fixed order.
RP_DEV void colsum_ownerT(float* part, int l31, int hi, const f32x16& o0, const f32x16& o1, float mul) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    float a = row16_sum(o0[r]), b = row16_sum(o1[r]);
    a += __shfl_xor(a, 16, 64);
    b += __shfl_xor(b, 16, 64);
    if (l31 == 0) {
      part[acc_row(r, hi)] = a * mul;
      part[32 + acc_row(r, hi)] = b * mul;
    }
  }
}

// load the owner operand (32 rows x 64) into registers: lane (row l31, half hi) keeps cols 32*hi .. +31
RP_DEV void load_owner(const float* row_ptr, int hi, float mul, float (&reg)[32]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const float4 x = ld4(row_ptr + 32 * hi + 4 * c);
    reg[4 * c + 0] = x.x * mul; reg[4 * c + 1] = x.y * mul; reg[4 * c + 2] = x.z * mul; reg[4 * c + 3] = x.w * mul;
  }
}

RP_DEV void pack_owner(const float (&reg)[32], bf16x8 (&pk)[4]) {
#pragma unroll
  for (int c = 0; c < 4; ++c) pk[c] = pack8(&reg[8 * c]);
}

// A wave's 32 x 32 accumulator tile (lane (c = l31, hf), register 4 g + j = row 8 g + 4 hf + j) stored as the four 16-byte runs each lane
// holds, run (g, lane) at chunk 64 g + lane:  element (row, c) sits at chunk (row >> 2) * 32 + c, dword row & 3.  Four
// global_store_dwordx4 of 1 KB CONTIGUOUS each, straight from the accumulators: no LDS round trip, no VALU (an fp32 MFMA kernel pays
// for every VALU instruction: profiles/r5_shadow_lab.txt), and 16 write requests per instruction (a row-major tile -- 16-byte pieces
// 128 bytes apart -- cost the forward 20 us and the backward 45 us per launch in request issue alone).  The readers pay the gather
// instead, on the load side where it is hidden: attn_bwd_dkdv_p_kernel (P tiles, 4-byte loads) and ds_matmul_t_kernel (dS tiles,
// through LDS-DMA).
// Non-temporal stores: the tiles are read by a LATER kernel, long after they have left every cache (510 MB per launch at 128 images);
// without the hint their lines displace the K / V (Q / dO) tiles the same workgroups keep re-reading from L2 (forward: -13 us).
template <bool NTS = true>      // NTS = false: plain stores, for tiles the NEXT kernel reads (the EMM's score tiles)
RP_DEV void store_tile_runs(float* tile, const f32x16& v, int lane) {
  typedef float f4v __attribute__((ext_vector_type(4)));
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    const f4v x = {v[4 * g], v[4 * g + 1], v[4 * g + 2], v[4 * g + 3]};
    if (NTS) __builtin_nontemporal_store(x, reinterpret_cast<f4v*>(tile + 4 * (64 * g + lane)));
    else *reinterpret_cast<f4v*>(tile + 4 * (64 * g + lane)) = x;
  }
}

// Software-pipelined tile loop (one barrier per tile, two LDS buffers).  MFMA operands are fetched from LDS one phase
// ahead of the MFMAs that consume them, so the matrix pipe never waits on a ds_read (the first version issued each
// read right before its MFMA and exposed ~2000 cycles of LDS latency per 4096-cycle tile):
//   top      : global loads of tile t+1 -> VGPRs;  LDS reads: K[t] second half, V[t] rows 0-7
//   S chain  : 16 MFMAs on K[t] first half (read during the previous tile's last phase) + 16 on the second half
//   softmax  : VALU;  LDS reads: V[t] rows 8-15
//   PV part 1: 16 MFMAs;  tile t+1 VGPRs -> LDS[other];  barrier;  LDS reads: K[t+1] first half
//   PV part 2: 16 MFMAs (cover the K[t+1] read)
// COLS (with STATS): the same pass also reduces every score tile along the OTHER index -- per loop row (key) the maximum and the sum of
// exp2 over the wave's 32 owner rows (DPP row reductions + one cross-row exchange, no LDS) -- and writes these partials; 18 of them
// per key combine into the column log-sum-exp (colstats_finalize_kernel).  The dual softmax's row and column normalisers then cost
// ONE pass over S instead of two (rp_emm_stats).
// SAVEP (training forward of the stored-P backward, exact fp32 only): every tile's exp2(s - running max) also leaves for HBM
// (store_tile_runs) together with that running max; the backward then needs neither Q K^T nor an exponential (attn_bwd_dkdv_p_kernel).
template <int NW, bool STATS, int WPS, bool BF, bool COLS = false, bool SAVEP = false>
__global__ __launch_bounds__(NW * 64, WPS) void attn_fwd_kernel(AttnP p) {
  constexpr int NT = NW * 64;
  constexpr int NPF = (512 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float Ks[2][32 * KST];
  __shared__ __attribute__((aligned(16))) float Vs[STATS ? 1 : 2][STATS ? 4 : 32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh, qblk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, qblk)) return;
  const int h = zh % p.H, z = zh / p.H;
  const int q0 = (qblk * NW + wave) * 32;
  const float* qb = p.q + (long long)(z ^ p.q_xor) * NTOK * p.ldq + h * 64;
  const float* kb = p.k + (long long)(z ^ (p.k_xor & 1)) * NTOK * p.ldk + h * 64;
  const float* vb = STATS ? nullptr : p.v + (long long)(z ^ (p.k_xor >> 1)) * NTOK * p.ldv + h * 64;

  float qreg[32];
  load_owner(qb + (long long)(q0 + l31) * p.ldq, hi, p.scale * RP_LOG2E, qreg);   // scores in log2 units
  bf16x8 qpk[4];
  if (BF) pack_owner(qreg, qpk);

  f32x16 o0 = zero16(), o1 = zero16();
  float m = -INFINITY, l = 0.f;

  float4 kpre[NPF], vpre[NPF];
  tile_gload<NT>(kb, p.ldk, tid, kpre);
  if (!STATS) tile_gload<NT>(vb, p.ldv, tid, vpre);
  tile_sstore<NT, KST>(Ks[0], tid, kpre);
  if (!STATS) tile_sstore<NT, 64>(Vs[0], tid, vpre);
  __syncthreads();

  const int krow = l31 * KST + 32 * hi;          // this lane's K row / column half inside a tile
  float4 ka[4], kb2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) ka[c] = ld4(Ks[0] + krow + 4 * c);

  // tile 1 is already in flight while tile 0 is being consumed (global loads are issued right after each barrier, one
  // full tile of MFMAs ahead of the LDS store that needs them)
  tile_gload<NT>(kb + (long long)32 * p.ldk, p.ldk, tid, kpre);
  if (!STATS) tile_gload<NT>(vb + (long long)32 * p.ldv, p.ldv, tid, vpre);

  for (int t = 0; t < NTILE; ++t) {
    const int cur = t & 1;
    const bool more = t + 1 < NTILE;
    float va[16], vb_[16];
#pragma unroll
    for (int c = 0; c < 4; ++c) kb2[c] = ld4(Ks[cur] + krow + 16 + 4 * c);
    if (!STATS) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float* vr = Vs[cur] + acc_row(r, hi) * 64 + l31;
        va[2 * r] = vr[0];
        va[2 * r + 1] = vr[32];
      }
    }
    __builtin_amdgcn_sched_barrier(0);
    f32x16 s = zero16();
    if (BF) {
      s = mfma_bf(pack8(ka[0].x, ka[0].y, ka[0].z, ka[0].w, ka[1].x, ka[1].y, ka[1].z, ka[1].w), qpk[0], s);
      s = mfma_bf(pack8(ka[2].x, ka[2].y, ka[2].z, ka[2].w, ka[3].x, ka[3].y, ka[3].z, ka[3].w), qpk[1], s);
      s = mfma_bf(pack8(kb2[0].x, kb2[0].y, kb2[0].z, kb2[0].w, kb2[1].x, kb2[1].y, kb2[1].z, kb2[1].w), qpk[2], s);
      s = mfma_bf(pack8(kb2[2].x, kb2[2].y, kb2[2].z, kb2[2].w, kb2[3].x, kb2[3].y, kb2[3].z, kb2[3].w), qpk[3], s);
    } else {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s = mfma32(ka[c].x, qreg[4 * c + 0], s);
      s = mfma32(ka[c].y, qreg[4 * c + 1], s);
      s = mfma32(ka[c].z, qreg[4 * c + 2], s);
      s = mfma32(ka[c].w, qreg[4 * c + 3], s);
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      s = mfma32(kb2[c].x, qreg[16 + 4 * c + 0], s);
      s = mfma32(kb2[c].y, qreg[16 + 4 * c + 1], s);
      s = mfma32(kb2[c].z, qreg[16 + 4 * c + 2], s);
      s = mfma32(kb2[c].w, qreg[16 + 4 * c + 3], s);
    }
    }
    if (STATS && SAVEP)      // the EMM's score tiles (log2 units, scale folded in): rp_emm_apply / rp_emm_grad_ds read them instead of recomputing q k^T
      store_tile_runs<false>(p.pst + (((long long)zh * NTILE + (q0 >> 5)) * NTILE + t) * 1024, s, lane);
    if (COLS) {
      float2* cp = reinterpret_cast<float2*>(p.colpart) + ((long long)zh * NTILE + (q0 >> 5)) * NTOK + t * 32;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        float cm = row16_max(s[r]);
        cm = fmaxf(cm, __shfl_xor(cm, 16, 64));                    // the other DPP row of this 32-lane half (same hi = same loop row)
        float cl = row16_sum(fast_exp2(s[r] - cm));
        cl += __shfl_xor(cl, 16, 64);
        cp[acc_row(r, hi)] = make_float2(cm, cl);                  // every lane of the half holds the pair: one 8-byte line write, no exec-masked block
      }
    }
    float mx = s[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, s[r]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    const float mn = fmaxf(m, mx);
    const float alpha = fast_exp2(m - mn);
    m = mn;
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      s[r] = fast_exp2(s[r] - mn);
      ps += s[r];
    }
    l = l * alpha + ps;
    if (SAVEP && !STATS) {
      store_tile_runs(p.pst + (((long long)zh * NTILE + (q0 >> 5)) * NTILE + t) * 1024, s, lane);      // element (key k, query q): chunk (k >> 2) * 32 + q, dword k & 3
      p.mrun[((long long)zh * NTILE + t) * NTOK + q0 + l31] = mn;      // (both halves hold the same value: no exec-masked block)
    }
    if (!STATS) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const float* vr = Vs[cur] + acc_row(8 + r, hi) * 64 + l31;
        vb_[2 * r] = vr[0];
        vb_[2 * r + 1] = vr[32];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        o0[r] *= alpha;
        o1[r] *= alpha;
      }
      __builtin_amdgcn_sched_barrier(0);
      if (BF) {
        const bf16x8 pb = pack8(s[0], s[1], s[2], s[3], s[4], s[5], s[6], s[7]);
        o0 = mfma_bf(pack8(va[0], va[2], va[4], va[6], va[8], va[10], va[12], va[14]), pb, o0);
        o1 = mfma_bf(pack8(va[1], va[3], va[5], va[7], va[9], va[11], va[13], va[15]), pb, o1);
      } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        o0 = mfma32(va[2 * r], s[r], o0);
        o1 = mfma32(va[2 * r + 1], s[r], o1);
      }
      }
    }
    if (more) {
      tile_sstore<NT, KST>(Ks[cur ^ 1], tid, kpre);
      if (!STATS) tile_sstore<NT, 64>(Vs[cur ^ 1], tid, vpre);
    }
    __syncthreads();
    if (more) {
#pragma unroll
      for (int c = 0; c < 4; ++c) ka[c] = ld4(Ks[cur ^ 1] + krow + 4 * c);
      if (t + 2 < NTILE) {
        tile_gload<NT>(kb + (long long)(t + 2) * 32 * p.ldk, p.ldk, tid, kpre);
        if (!STATS) tile_gload<NT>(vb + (long long)(t + 2) * 32 * p.ldv, p.ldv, tid, vpre);
      }
    }
    if (!STATS) {
      __builtin_amdgcn_sched_barrier(0);
      if (BF) {
        const bf16x8 pb = pack8(s[8], s[9], s[10], s[11], s[12], s[13], s[14], s[15]);
        o0 = mfma_bf(pack8(vb_[0], vb_[2], vb_[4], vb_[6], vb_[8], vb_[10], vb_[12], vb_[14]), pb, o0);
        o1 = mfma_bf(pack8(vb_[1], vb_[3], vb_[5], vb_[7], vb_[9], vb_[11], vb_[13], vb_[15]), pb, o1);
      } else {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        o0 = mfma32(vb_[2 * r], s[8 + r], o0);
        o1 = mfma32(vb_[2 * r + 1], s[8 + r], o1);
      }
      }
    }
  }
  const float lt = l + __shfl_xor(l, 32, 64);
  if (!STATS) store_ownerT(p.o + ((long long)z * NTOK + q0 + l31) * p.ldo + h * 64, hi, o0, o1, 1.0f / lt);
  if (hi == 0) p.lse[((long long)z * p.H + h) * NTOK + q0 + l31] = m * RP_LN2 + logf(lt);   // natural-log lse
}

// ------------------------------------------------------------------------------------------------
// backward.  P = exp(S - lse_q);  dV = P^T dO;  dP = dO V^T;  dS = P o (dP - delta_q);
//            dQ = scale * dS K;  dK = scale * dS^T Q        (S already contains scale)
// Two deterministic passes (no atomics):
//   dkdv: wave owns 32 keys (K, V rows in VGPRs); loops over query tiles {Q, dO, lse, delta} in LDS.
//         S[q][kv] = sum_d Q[q][d] K[kv][d]   -> lane = key, regs = queries  (A = Q tile, B = K regs)
//         dP[q][kv] = sum_d dO[q][d] V[kv][d]                                 (A = dO tile, B = V regs)
//         dV^T[d][kv] += sum_q dO[q][d] P[q][kv] ; dK^T[d][kv] += sum_q Q[q][d] dS[q][kv]   (row-pattern reads)
//   dq  : wave owns 32 queries (Q, dO rows in VGPRs, lse/delta lane-local); loops over key tiles {K, V}.
//         S^T[kv][q], dP^T[kv][q] (A = K / V tile, B = Q / dO regs); dQ^T[d][q] += sum_kv K[kv][d] dS^T[kv][q]
// ------------------------------------------------------------------------------------------------
struct AttnBwdP {
  const float* q; const float* k; const float* v; const float* dout; const float* lse; const float* delta;
  float* dq; float* dk; float* dv;
  int H, ldq, ldk, ldv, lddo, lddq, lddk, lddv;
  float scale;
  int ZH;
  int kv_xor;   // 1: keys/values (and dK, dV) of problem z live at image z^1 relative to its queries (cross attention)
  float* ds;    // optional [Z][H][18 q-blocks][18 key-blocks][16][64]: scale * dS in 32x32 tiles (accumulator image) written by the
                // dK/dV pass, so dQ = dS K is one streaming rp_ds_matmul
  float *dk_colpart, *dv_colpart;   // optional [Z * 18][ldp]: column sums of dk / dv over each 32-row block (first of the H*64 columns)
  int ldp;
  const float* pst;    // stored-P pass only: what attn_fwd_kernel<SAVEP> wrote (AttnP::pst, AttnP::mrun)
  const float* mrun;
};

#ifdef RP_DKDV_PROBE
__device__ unsigned long long g_probe[16];
#define STAMP(i) { const unsigned long long now_ = __builtin_readcyclecounter(); acc_[i] += now_ - last_; last_ = now_; }
#else
#define STAMP(i)
#endif
template <int NW, int WPS, bool BF>
__global__ __launch_bounds__(NW * 64, WPS) void attn_bwd_dkdv_kernel(AttnBwdP p) {
#ifdef RP_DKDV_PROBE
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  constexpr int NT = NW * 64;
  constexpr int NPF = (512 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float Qs[2][32 * KST];
  __shared__ __attribute__((aligned(16))) float Ds[2][32 * KST];
  __shared__ __attribute__((aligned(16))) float Ls[2][64];   // [0..31] lse, [32..63] delta of the query tile
  __shared__ __attribute__((aligned(16))) float Tst[BF ? 1 : NW][BF ? 4 : 512];      // per-wave staging of the stored dS tile (store_acc_image_lds)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh, qblk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, qblk)) return;
  const int h = zh % p.H, z = zh / p.H;
  const int k0 = (qblk * NW + wave) * 32;
  const int zq = z ^ p.kv_xor;                       // z indexes the key/value image
  const float* qb = p.q + (long long)zq * NTOK * p.ldq + h * 64;
  const float* dob = p.dout + (long long)zq * NTOK * p.lddo + h * 64;
  const float* lseb = p.lse + ((long long)zq * p.H + h) * NTOK;
  const float* delb = p.delta + ((long long)zq * p.H + h) * NTOK;

  float kreg[32], vreg[32];
  load_owner(p.k + ((long long)z * NTOK + k0 + l31) * p.ldk + h * 64, hi, p.scale * RP_LOG2E, kreg);   // scale*log2e folded into K
  load_owner(p.v + ((long long)z * NTOK + k0 + l31) * p.ldv + h * 64, hi, 1.0f, vreg);
  bf16x8 kpk[4], vpk[4];
  if (BF) { pack_owner(kreg, kpk); pack_owner(vreg, vpk); }

  f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
  float4 qpre[NPF], dpre[NPF];
  float lpre = 0.f;
  tile_gload<NT>(qb, p.ldq, tid, qpre);
  tile_gload<NT>(dob, p.lddo, tid, dpre);
  // lse (scaled to log2 units) and delta of the query tile: loaded by EVERY thread, branch-free (lane & 63 picks the value) -- an
  // exec-masked `if (tid < 64)` load here became its own basic block and hipcc joined it with s_waitcnt vmcnt(0), i.e. every tile waited
  // for its own just-issued Q / dO prefetch at the top of the loop (tools/dkdv_probe.py: 4 k cycles per tile)
  const float* lsrc = (lane & 32) ? delb + (lane & 31) : lseb + (lane & 31);
  const float lmul = (lane & 32) ? 1.0f : RP_LOG2E;
  lpre = lsrc[0] * lmul;
  tile_sstore<NT, KST>(Qs[0], tid, qpre);
  tile_sstore<NT, KST>(Ds[0], tid, dpre);
  if (tid < 64) Ls[0][tid] = lpre;
  __syncthreads();

  STAMP(0)
  for (int t = 0; t < NTILE; ++t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) {
      tile_gload<NT>(qb + (long long)(t + 1) * 32 * p.ldq, p.ldq, tid, qpre);
      tile_gload<NT>(dob + (long long)(t + 1) * 32 * p.lddo, p.lddo, tid, dpre);
      lpre = lsrc[(t + 1) * 32] * lmul;
    }
    STAMP(1)
    f32x16 s = score_tile<BF>(Qs[cur], l31, hi, kreg, kpk);     // rows = queries acc_row(r,hi), lane = key
    f32x16 dp = score_tile<BF>(Ds[cur], l31, hi, vreg, vpk);
#ifdef RP_DKDV_PROBE
    asm volatile("" : "+v"(s), "+v"(dp));
#endif
    STAMP(2)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = acc_row(r, hi);
      const float pr = fast_exp2(s[r] - Ls[cur][qi]);
      s[r] = pr;
      dp[r] = pr * (dp[r] - Ls[cur][32 + qi]);
    }
#ifdef RP_DKDV_PROBE
    asm volatile("" : "+v"(s), "+v"(dp));
#endif
    STAMP(3)
    if (p.ds) {      // TILED: tile (query block t, key block) = this wave's register image [16 r][64 lanes], 4 KB contiguous, fully
      // coalesced 256-byte stores; rp_ds_matmul (below) reads it -- the dQ pass then needs neither S nor dP again
      const long long tile = (((long long)zq * p.H + h) * NTILE + t) * NTILE + (k0 >> 5);
      if (BF) store_acc_image_bf16(reinterpret_cast<unsigned short*>(p.ds) + tile * 1024, dp, p.scale, lane);   // bf16 operands downstream: bf16 tiles
      else store_acc_image_lds(p.ds + tile * 1024, Tst[BF ? 0 : wave], dp, p.scale, lane);
    }
    STAMP(4)
    accum_tile<KST, BF>(Ds[cur], l31, hi, s, dv0, dv1);     // dV^T += dO^T P
    accum_tile<KST, BF>(Qs[cur], l31, hi, dp, dk0, dk1);    // dK^T += Q^T dS
#ifdef RP_DKDV_PROBE
    asm volatile("" : "+v"(dv0), "+v"(dv1), "+v"(dk0), "+v"(dk1));
#endif
    STAMP(5)
    if (t + 1 < NTILE) {
      tile_sstore<NT, KST>(Qs[cur ^ 1], tid, qpre);
      tile_sstore<NT, KST>(Ds[cur ^ 1], tid, dpre);
      if (tid < 64) Ls[cur ^ 1][tid] = lpre;
    }
    STAMP(6)
    __syncthreads();
    STAMP(7)
  }
#ifdef RP_DKDV_PROBE
  if (lane == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&g_probe[i], acc_[i]); atomicAdd(&g_probe[8], 1ull); }
#endif
  store_ownerT(p.dv + ((long long)z * NTOK + k0 + l31) * p.lddv + h * 64, hi, dv0, dv1, 1.0f);
  store_ownerT(p.dk + ((long long)z * NTOK + k0 + l31) * p.lddk + h * 64, hi, dk0, dk1, p.scale);
  if (p.dk_colpart) {
    const long long prow = ((long long)z * NTILE + (k0 >> 5)) * p.ldp + h * 64;
    colsum_ownerT(p.dv_colpart + prow, l31, hi, dv0, dv1, 1.0f);
    colsum_ownerT(p.dk_colpart + prow, l31, hi, dk0, dk1, p.scale);
  }
}

// ------------------------------------------------------------------------------------------------
// Stored-P form of the dK/dV pass: the training forward left exp2(s - m_t) of every tile (store_tile_runs) and the running
// maxima m_t, so
//   P[q][kv] = Pt * exp2(m_t[q] - lse2[q])        one multiply per element, the factor formed once per (query, tile) by the loader lanes
// and the pass executes THREE products per tile (dP = dO V^T, dV += dO^T P, dK += Q^T dS) instead of four: no Q K^T recompute, no
// exponential, K is not read at all.  With rp_ds_matmul_t for dQ the attention backward executes exactly its four algorithmic products.
// A wave owns 32 keys (V rows in VGPRs, dV / dK in 64 accumulators) and walks the 18 query tiles {Q, dO} staged in LDS; its P tile
// (4 KB, this wave's alone) comes straight from HBM one tile ahead -- sixteen 4-byte loads, register r of lane (kv, hi) = query
// acc_row(r, hi), the accumulator layout of the dP product -- and scale * dS leaves as store_tile_runs writes it: neither tile touches
// LDS or costs a VALU instruction.  `scale` rides on the V rows and on delta:
// dP' = scale dP, delta' = scale delta  =>  P (dP' - delta') = scale dS, the stored tile and the dK operand.
// ------------------------------------------------------------------------------------------------
template <int NW, int WPS>
__global__ __launch_bounds__(NW * 64, WPS) void attn_bwd_dkdv_p_kernel(AttnBwdP p) {
#ifdef RP_DKDV_PROBE
  unsigned long long acc_[8] = {0, 0, 0, 0, 0, 0, 0, 0}, last_ = __builtin_readcyclecounter();
#endif
  constexpr int NT = NW * 64;
  constexpr int NPF = (512 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float Qs[2][32 * KST];
  __shared__ __attribute__((aligned(16))) float Ds[2][32 * KST];
  __shared__ __attribute__((aligned(16))) float Ls[2][NW][64];   // per wave (the factor depends on the key block): [0..31] exp2(m_t - lse2), [32..63] scale * delta
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh, qblk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, qblk)) return;
  const int h = zh % p.H, z = zh / p.H;
  const int kb = qblk * NW + wave, k0 = kb * 32;
  const float* qb = p.q + (long long)z * NTOK * p.ldq + h * 64;
  const float* dob = p.dout + (long long)z * NTOK * p.lddo + h * 64;
  const float* lseb = p.lse + (long long)zh * NTOK;
  const float* delb = p.delta + (long long)zh * NTOK;
  const float* mb = p.mrun + ((long long)zh * NTILE + kb) * NTOK;
  // P tile (t, kb): + t * 18 * 1024; element (key kk = l31, query q) at chunk (kk >> 2) * 32 + q, dword kk & 3 (store_tile_runs by the forward's
  // lanes = queries): register r = query acc_row(r, hi) is a 4-byte load 16 (acc_row(r, 0) + 4 hi) bytes further on
  const float* pt = p.pst + ((long long)zh * NTILE * NTILE + kb) * 1024 + (l31 >> 2) * 128 + (l31 & 3) + 16 * hi;
  float* dst = p.ds + ((long long)zh * NTILE * NTILE + kb) * 1024;                           // likewise

  float vreg[32];
  load_owner(p.v + ((long long)z * NTOK + k0 + l31) * p.ldv + h * 64, hi, p.scale, vreg);
  const bf16x8 nopk[4] = {};
  f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
  float4 qpre[NPF], dpre[NPF];
  // branch-free loader (see attn_bwd_dkdv_kernel): lanes 0-31 form the factor of query l31, lanes 32-63 carry its scaled delta
  const float* asrc = (lane & 32) ? delb + l31 : lseb + l31;
  const float* bsrc = (lane & 32) ? delb + l31 : mb + l31;
  // pure arithmetic, no select: a ternary around the exponential became a divergent branch whose blocks hipcc joined with
  // s_waitcnt vmcnt(0) -- every tile then waited for its own just-issued prefetch (the pitfall attn_bwd_dkdv_kernel's loader documents)
  // The two values are only LOADED at the top of a tile; the factor is formed when it is written to LDS at the tile's end -- computed
  // where it is loaded it put s_waitcnt vmcnt(0) behind the loads, in front of the tile's prefetch (vmcnt retires in order).
  const float km = (lane & 32) ? p.scale : 0.f, ke = (lane & 32) ? 0.f : 1.f;
  auto lfac = [&](float a, float b) { return fmaf(km, a, ke * fast_exp2(ke * fmaf(-a, RP_LOG2E, b))); };
  auto pload = [&](f32x16& x, int t) {
    const float* tp = pt + (long long)t * (NTILE * 1024);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = tp[acc_row(r, 0) * 4];
  };
  f32x16 pa, pb;
  tile_gload<NT>(qb, p.ldq, tid, qpre);
  tile_gload<NT>(dob, p.lddo, tid, dpre);
  float la = asrc[0], lb = bsrc[0];
  pload(pa, 0);
  tile_sstore<NT, KST>(Qs[0], tid, qpre);
  tile_sstore<NT, KST>(Ds[0], tid, dpre);
  Ls[0][wave][lane] = lfac(la, lb);
  __syncthreads();

  auto step = [&](f32x16& pc, f32x16& pn, int t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) {
      tile_gload<NT>(qb + (long long)(t + 1) * 32 * p.ldq, p.ldq, tid, qpre);
      tile_gload<NT>(dob + (long long)(t + 1) * 32 * p.lddo, p.lddo, tid, dpre);
      la = asrc[(t + 1) * 32];
      lb = bsrc[(t + 1) * 32];
      pload(pn, t + 1);
    }
    STAMP(1)
    f32x16 dp = score_tile<false>(Ds[cur], l31, hi, vreg, nopk);      // scale dP: rows = queries acc_row(r, hi), lane = key
#ifdef RP_DKDV_PROBE
    asm volatile("" : "+v"(dp));
#endif
    STAMP(2)
    const float* L = Ls[cur][wave];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int qi = acc_row(r, hi);
      pc[r] *= L[qi];
      dp[r] = pc[r] * (dp[r] - L[32 + qi]);
    }
#ifdef RP_DKDV_PROBE
    asm volatile("" : "+v"(dp), "+v"(pc));
#endif
    STAMP(3)
    store_tile_runs(dst + (long long)t * (NTILE * 1024), dp, lane);     // scale dS; element (query q, key k): chunk (q >> 2) * 32 + k, dword q & 3
    STAMP(4)
    accum_tile<KST, false>(Ds[cur], l31, hi, pc, dv0, dv1);     // dV^T += dO^T P
    accum_tile<KST, false>(Qs[cur], l31, hi, dp, dk0, dk1);     // dK^T += Q^T (scale dS)
#ifdef RP_DKDV_PROBE
    asm volatile("" : "+v"(dv0), "+v"(dv1), "+v"(dk0), "+v"(dk1));
#endif
    STAMP(5)
    if (t + 1 < NTILE) {
      tile_sstore<NT, KST>(Qs[cur ^ 1], tid, qpre);
      tile_sstore<NT, KST>(Ds[cur ^ 1], tid, dpre);
      Ls[cur ^ 1][wave][lane] = lfac(la, lb);
    }
    STAMP(6)
    __syncthreads();
    STAMP(7)
  };
  for (int t = 0; t < NTILE; t += 2) {      // two register sets for the P tiles rotate without copies (18 tiles)
    step(pa, pb, t);
    step(pb, pa, t + 1);
  }
#ifdef RP_DKDV_PROBE
  if (lane == 0) { for (int i = 0; i < 8; ++i) atomicAdd(&g_probe[i], acc_[i]); atomicAdd(&g_probe[8], 1ull); }
#endif
  store_ownerT(p.dv + ((long long)z * NTOK + k0 + l31) * p.lddv + h * 64, hi, dv0, dv1, 1.0f);
  store_ownerT(p.dk + ((long long)z * NTOK + k0 + l31) * p.lddk + h * 64, hi, dk0, dk1, 1.0f);
  if (p.dk_colpart) {
    const long long prow = ((long long)z * NTILE + kb) * p.ldp + h * 64;
    colsum_ownerT(p.dv_colpart + prow, l31, hi, dv0, dv1, 1.0f);
    colsum_ownerT(p.dk_colpart + prow, l31, hi, dk0, dk1, 1.0f);
  }
}

template <int NW, int WPS, bool BF>
__global__ __launch_bounds__(NW * 64, WPS) void attn_bwd_dq_kernel(AttnBwdP p) {
  constexpr int NT = NW * 64;
  constexpr int NPF = (512 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float Ks[2][32 * KST];
  __shared__ __attribute__((aligned(16))) float Vs[2][32 * KST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh, qblk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, qblk)) return;
  const int h = zh % p.H, z = zh / p.H;
  const int q0 = (qblk * NW + wave) * 32;
  const float* kb = p.k + (long long)(z ^ p.kv_xor) * NTOK * p.ldk + h * 64;   // z indexes the query image
  const float* vb = p.v + (long long)(z ^ p.kv_xor) * NTOK * p.ldv + h * 64;

  float qreg[32], dreg[32];
  load_owner(p.q + ((long long)z * NTOK + q0 + l31) * p.ldq + h * 64, hi, p.scale * RP_LOG2E, qreg);
  load_owner(p.dout + ((long long)z * NTOK + q0 + l31) * p.lddo + h * 64, hi, 1.0f, dreg);
  bf16x8 qpk[4], dpk[4];
  if (BF) { pack_owner(qreg, qpk); pack_owner(dreg, dpk); }
  const float lse = p.lse[((long long)z * p.H + h) * NTOK + q0 + l31] * RP_LOG2E;
  const float del = p.delta[((long long)z * p.H + h) * NTOK + q0 + l31];

  f32x16 dq0 = zero16(), dq1 = zero16();
  float4 kpre[NPF], vpre[NPF];
  tile_gload<NT>(kb, p.ldk, tid, kpre);
  tile_gload<NT>(vb, p.ldv, tid, vpre);
  tile_sstore<NT, KST>(Ks[0], tid, kpre);
  tile_sstore<NT, KST>(Vs[0], tid, vpre);
  __syncthreads();
  for (int t = 0; t < NTILE; ++t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) {
      tile_gload<NT>(kb + (long long)(t + 1) * 32 * p.ldk, p.ldk, tid, kpre);
      tile_gload<NT>(vb + (long long)(t + 1) * 32 * p.ldv, p.ldv, tid, vpre);
    }
    f32x16 s = score_tile<BF>(Ks[cur], l31, hi, qreg, qpk);     // S^T: rows = keys, lane = query
    f32x16 dp = score_tile<BF>(Vs[cur], l31, hi, dreg, dpk);    // dP^T
#pragma unroll
    for (int r = 0; r < 16; ++r) dp[r] = fast_exp2(s[r] - lse) * (dp[r] - del);
    accum_tile<KST, BF>(Ks[cur], l31, hi, dp, dq0, dq1);    // dQ^T += K^T dS^T
    if (t + 1 < NTILE) {
      tile_sstore<NT, KST>(Ks[cur ^ 1], tid, kpre);
      tile_sstore<NT, KST>(Vs[cur ^ 1], tid, vpre);
    }
    __syncthreads();
  }
  store_ownerT(p.dq + ((long long)z * NTOK + q0 + l31) * p.lddq + h * 64, hi, dq0, dq1, p.scale);
}

// ------------------------------------------------------------------------------------------------
// out[z][i][h*64 + d] = sum_j ds[z*H + h][i][j] * b[z ^ b_xor][j][h*64 + d]    (576 x 576 per problem, 64 columns)
// The second half of the stored-dS backward (dQ = dS K of the attention, dK = dS^T-major x Q of the EMM), as a STREAMING kernel:
// round 2 ran it as three batched rp_gemm launches per call at 1.95 TB/s of a row-major dS.  dS is now stored TILED by its producers:
// tile (i-block, j-block) is the producing wave's accumulator image T[r][lane] = dS[i = acc_row(r, lane >> 5)][j = lane & 31], 4 KB
// contiguous, the 18 j-tiles of an i-block back to back (72 KB).  A wave here owns 32 rows i and takes its dS operand STRAIGHT from
// memory in MFMA B-operand shape: lane (i = l31, hi) reads the 64 contiguous bytes T[r(i)][32 hh(i) + 16 hi .. + 15] (four 16-byte
// loads; register e is column j = 16 hi + e), so a wave instruction stays inside one 4 KB tile (DRAM-page and L1 friendly: row-major
// dS made every lane touch its own 2304-byte-strided line and streamed at 2.4 TB/s), three tiles in flight.  Only the small operand
// (the 32 x 64 rows of b, L2-resident: 147 KB per problem, problems pinned to an XCD) goes through LDS, shared by the workgroup's
// waves.  out^T[d][i] += b[j][d] * ds[i][j]: 32 MFMAs per tile, no recompute, one launch for all heads.
// ------------------------------------------------------------------------------------------------
struct DsMmP {
  const float* ds; const float* b; float* out;
  int H, ldb, ldo, b_xor, ZH;
  int reverse;      // walk the problems last-to-first: the producer wrote them first-to-last, so the newest tiles may still sit in the
                    // 256 MB memory-side cache
  float* colpart;   // optional [Z * 18][ldp]: column sums of `out` over each 32-row block (first of the H*64 columns)
  int ldp;
};

// acc^T[d][owner] += sum_r Ts[row0 + r][d] * p[r]: like accum_tile but register r pairs with the CONTIGUOUS row row0 + r (row0 = 16 hi):
// the owner operand then is 64 contiguous bytes of its memory row per lane
RP_DEV void accum_rows16(const float* Ts, int l31, int hi, const f32x16& p, f32x16& o0, f32x16& o1) {
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const float* vr = Ts + (16 * hi + r) * 64 + l31;
    o0 = mfma32(vr[0], p[r], o0);
    o1 = mfma32(vr[32], p[r], o1);
  }
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void ds_matmul_kernel(DsMmP p) {
  constexpr int NT = NW * 64;
  constexpr int KT = 32;                                  // columns of ds (= rows of b) per step
  constexpr int NSTEP = NTOK / KT;
  constexpr int NPF = (KT * 16 + NT - 1) / NT;            // float4 per thread per b tile
  __shared__ __attribute__((aligned(16))) float Bs[2][KT * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh, blk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, blk)) return;
  if (p.reverse) zh = p.ZH - 1 - zh;
  const int h = zh % p.H, z = zh / p.H;
  const int i0 = (blk * NW + wave) * 32;
  const float* bb = p.b + (long long)(z ^ p.b_xor) * NTOK * p.ldb + h * 64;
  // row i = l31 of a tile lives in register r = (i & 3) + 4 (i >> 3) of half-wave hh = (i >> 2) & 1 of the producer's image
  const float* arow = p.ds + ((long long)zh * NTILE + (i0 >> 5)) * NTILE * 1024 + ((l31 & 3) + 4 * (l31 >> 3)) * 64 + 32 * ((l31 >> 2) & 1) + 16 * hi;

  f32x16 o0 = zero16(), o1 = zero16();
  float4 bpre[NPF];
  constexpr int NA = KT / 8;                              // float4 of ds per lane per step
  float4 a0[NA], a1[NA], a2[NA];
  auto aload = [&](float4 (&a)[NA], int t) {
#pragma unroll
    for (int g = 0; g < NA; ++g) {
      const float* src = arow + 1024 * ((KT / 32) * t + (g >> 2)) + 4 * (g & 3);
      a[g] = ld4(src);
    }
  };
  auto bload = [&](int t) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      int f = tid + NT * j;
      if ((KT * 16) % NT != 0) f = min(f, KT * 16 - 1);
      bpre[j] = ld4(bb + (long long)(t * KT + (f >> 4)) * p.ldb + (f & 15) * 4);
    }
  };
  auto bstore = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      int f = tid + NT * j;
      if ((KT * 16) % NT != 0) f = min(f, KT * 16 - 1);
      st4(dst + f * 4, bpre[j]);
    }
  };
  bload(0);
  aload(a0, 0);
  aload(a1, 1);
  bstore(Bs[0]);
  __syncthreads();
  bload(1);
  auto step = [&](const float4 (&a)[NA], float4 (&anext)[NA], int t) {
    const int cur = t & 1;
    if (t + 2 < NSTEP) aload(anext, t + 2);
#pragma unroll
    for (int u = 0; u < KT / 32; ++u) {
      f32x16 pr;
#pragma unroll
      for (int g = 0; g < 4; ++g) { pr[4 * g] = a[4 * u + g].x; pr[4 * g + 1] = a[4 * u + g].y; pr[4 * g + 2] = a[4 * u + g].z; pr[4 * g + 3] = a[4 * u + g].w; }
      accum_rows16(Bs[cur] + u * 32 * 64, l31, hi, pr, o0, o1);
    }
    if (t + 1 < NSTEP) bstore(Bs[cur ^ 1]);
    __syncthreads();
    if (t + 2 < NSTEP) bload(t + 2);
  };
  for (int t = 0; t < NSTEP; t += 3) {      // NSTEP = 18 or 9: three register sets rotate without copies
    step(a0, a2, t);
    step(a1, a0, t + 1);
    step(a2, a1, t + 2);
  }
  store_ownerT(p.out + ((long long)z * NTOK + i0 + l31) * p.ldo + h * 64, hi, o0, o1, 1.0f);
  if (p.colpart) colsum_ownerT(p.colpart + ((long long)z * NTILE + (i0 >> 5)) * p.ldp + h * 64, l31, hi, o0, o1, 1.0f);
}

// rp_ds_matmul_t: the same product for tiles stored by store_tile_runs (what attn_bwd_dkdv_p_kernel writes with four contiguous 16-byte
// stores per lane straight from its accumulators).  Both operands reach LDS by LDS-DMA (global_load_lds_dwordx4: no VGPR, no ds_write,
// nothing to wait for but vmcnt) -- the wave's own dS tile, 4 KB, gathered into T[j][i] row-major in a wave-private double buffer, and the
// 32 x 64 rows of b shared by the workgroup -- and feed the matrix pipe through conflict-free ds_read_b32 with
// immediate offsets: k-step s pairs column j = 2 s + hi with the two half-waves, lane (i = l31, hi) reads T[2 s + hi][i] as the B operand
// and b[2 s + hi][l31], [l31 + 32] as the A operands (three LDS reads per two MFMAs; LDS reads are free next to an fp32 MFMA).
template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void ds_matmul_t_kernel(DsMmP p) {
  __shared__ __attribute__((aligned(16))) float Bs[2][32 * 64];
  __shared__ __attribute__((aligned(16))) float Dt[NW][2][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  int zh, blk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, blk)) return;
  if (p.reverse) zh = p.ZH - 1 - zh;
  const int h = zh % p.H, z = zh / p.H;
  const int ib = blk * NW + wave, i0 = ib * 32;
  const float* bb = p.b + (long long)(z ^ p.b_xor) * NTOK * p.ldb + h * 64;
  const float* tiles = p.ds + ((long long)zh * NTILE + ib) * NTILE * 1024;
  // DMA plan per step: this wave's dS tile = 4 pieces of 1 KB as they lie; the b tile = 8 pieces of 4 rows (lane -> row lane >> 4, 16-byte
  // chunk lane & 15), wave w moves pieces w, w + NW, ... (every wave issues the same number: the last ones repeat piece 7)
  constexpr int BP = (8 + NW - 1) / NW;
  // dS tile: stored by store_tile_runs (element (query i, key j) at chunk (i >> 2) * 32 + j, dword i & 3); LDS wants T[j][i] row-major =
  // chunk j * 8 + (i >> 2): DMA piece c (LDS chunks 64 c + lane: j = 8 c + (lane >> 3), i >> 2 = lane & 7) gathers global chunk
  // (lane & 7) * 32 + 8 c + (lane >> 3) -- eight whole 128-byte lines per instruction
  const unsigned dvoff = ((lane & 7) * 32 + (lane >> 3)) * 16, bvoff = (unsigned)((lane >> 4) * p.ldb * 4 + (lane & 15) * 16);
  const unsigned dt0 = lds_byte_addr(&Dt[wave][0][0]), bs0 = lds_byte_addr(&Bs[0][0]);
  auto issue = [&](int t, int buf) {
    const float* ts = tiles + (long long)t * 1024;
#pragma unroll
    for (int c = 0; c < 4; ++c) glds16(uniform_ptr(ts + 32 * c), dvoff, dt0 + buf * 4096 + c * 1024);
#pragma unroll
    for (int c = 0; c < BP; ++c) {
      const int piece = min(wave + NW * c, 7);
      glds16(uniform_ptr(bb + (long long)(t * 32 + 4 * piece) * p.ldb), bvoff, bs0 + buf * 8192 + piece * 1024);
    }
  };
  f32x16 o0 = zero16(), o1 = zero16();
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < NTILE; ++t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) issue(t + 1, cur ^ 1);
    const float* B = Bs[cur] + hi * 64 + l31;
    const float* D = Dt[wave][cur] + hi * 32 + l31;
#pragma unroll
    for (int s = 0; s < 16; ++s) {
      const float dsv = D[64 * s];
      o0 = mfma32(B[128 * s], dsv, o0);
      o1 = mfma32(B[128 * s + 32], dsv, o1);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
  store_ownerT(p.out + ((long long)z * NTOK + i0 + l31) * p.ldo + h * 64, hi, o0, o1, 1.0f);
  if (p.colpart) colsum_ownerT(p.colpart + ((long long)z * NTILE + ib) * p.ldp + h * 64, l31, hi, o0, o1, 1.0f);
}

// ds_matmul_t_kernel on v_mfma_f32_16x16x4_f32.  Why: with several waves per SIMD taking turns, v_mfma_f32_32x32x2_f32 issues every ~82
// cycles instead of 64 (profiles/README.md "measured ceilings": 103-123 TF at 2-8 waves per SIMD against 148 TF alone) while 16x16x4 holds
// 152-155 TF at every occupancy -- and this kernel is nothing but MFMAs fed from LDS by three waves per SIMD.
//   out^T[d][i] += b[j][d] ds[i][j]:  A[m = d][k = j] = b,  lane (d' = l & 15, kq = l >> 4) reads b[4 ks + kq][16 mb + d']   (4 reads per k-step)
//                                     B[k = j][n = i] = T,  lane (i' = l & 15, kq)          reads T[4 ks + kq][16 nb + i']   (2 reads per k-step)
//   8 MFMAs of 32 cycles per k-step of four keys, 8 k-steps per tile; C block (mb, nb): lane (i', kq) register e = out[16 nb + i'][16 mb + 4 kq + e]
//   -> 16-byte stores.
// Both LDS images are laid out by the DMA's SOURCE addressing so that the four kq groups of a read hit different banks:
//   b tile : chunk c of row j sits at chunk c ^ (4 (j & 3))        -> logical column 16 mb + d' of row 4 ks + kq at  16 (mb ^ kq) + d'
//   T tile : chunk c of row j sits at chunk c ^ (4 ((j >> 1) & 1))  -> logical column 16 nb + i' of row 4 ks + kq at  16 (nb ^ (kq >> 1)) + i'
template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void ds_matmul_t16_kernel(DsMmP p) {
  __shared__ __attribute__((aligned(16))) float Bs[2][32 * 64];
  __shared__ __attribute__((aligned(16))) float Dt[NW][2][1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
  int zh, blk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, blk)) return;
  if (p.reverse) zh = p.ZH - 1 - zh;
  const int h = zh % p.H, z = zh / p.H;
  const int ib = blk * NW + wave, i0 = ib * 32;
  const float* bb = p.b + (long long)(z ^ p.b_xor) * NTOK * p.ldb + h * 64;
  const float* tiles = p.ds + ((long long)zh * NTILE + ib) * NTILE * 1024;
  constexpr int BP = (8 + NW - 1) / NW;
  // dS piece c: LDS chunks 64 c + lane -> row j = 8 c + (lane >> 3), physical chunk lane & 7 = logical (i >> 2) ^ (4 ((j >> 1) & 1));
  // the tile stores element (query i, key j) at chunk (i >> 2) * 32 + j (store_tile_runs)
  const unsigned dvoff = ((((lane & 7) ^ (4 * ((lane >> 4) & 1))) * 32) + (lane >> 3)) * 16;
  // b piece: 4 rows (lane >> 4) x 16 chunks; physical chunk lane & 15 holds logical chunk (lane & 15) ^ (4 (row & 3)), row & 3 = lane >> 4
  const unsigned bvoff = (unsigned)((lane >> 4) * p.ldb * 4 + (((lane & 15) ^ (4 * (lane >> 4))) * 16));
  const unsigned dt0 = lds_byte_addr(&Dt[wave][0][0]), bs0 = lds_byte_addr(&Bs[0][0]);
  auto issue = [&](int t, int buf) {
    const float* ts = tiles + (long long)t * 1024;
#pragma unroll
    for (int c = 0; c < 4; ++c) glds16(uniform_ptr(ts + 32 * c), dvoff, dt0 + buf * 4096 + c * 1024);
#pragma unroll
    for (int c = 0; c < BP; ++c) {
      const int piece = min(wave + NW * c, 7);
      glds16(uniform_ptr(bb + (long long)(t * 32 + 4 * piece) * p.ldb), bvoff, bs0 + buf * 8192 + piece * 1024);
    }
  };
  f32x4 acc[4][2];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int n = 0; n < 2; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  // per-lane read offsets (floats) inside a tile image; k-step ks adds 4 rows
  int aoff[4], boff[2];
#pragma unroll
  for (int m = 0; m < 4; ++m) aoff[m] = kq * 64 + 16 * (m ^ kq) + l15;
#pragma unroll
  for (int n = 0; n < 2; ++n) boff[n] = kq * 32 + 16 * (n ^ (kq >> 1)) + l15;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int t = 0; t < NTILE; ++t) {
    const int cur = t & 1;
    if (t + 1 < NTILE) issue(t + 1, cur ^ 1);
    const float* B = Bs[cur];
    const float* D = Dt[wave][cur];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
      float av[4], bv[2];
#pragma unroll
      for (int m = 0; m < 4; ++m) av[m] = B[ks * 256 + aoff[m]];
#pragma unroll
      for (int n = 0; n < 2; ++n) bv[n] = D[ks * 128 + boff[n]];
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], bv[n], acc[m][n], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }
#pragma unroll
  for (int n = 0; n < 2; ++n) {
    float* orow = p.out + ((long long)z * NTOK + i0 + 16 * n + l15) * p.ldo + h * 64 + 4 * kq;
#pragma unroll
    for (int m = 0; m < 4; ++m) st4(orow + 16 * m, make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]));
  }
  if (p.colpart) {      // column sums of `out` over the wave's 32 rows: lane (i', kq) register e is column 16 m + 4 kq + e of rows 16 n + i'
    float* part = p.colpart + ((long long)z * NTILE + ib) * p.ldp + h * 64;
#pragma unroll
    for (int m = 0; m < 4; ++m)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float v = row16_sum(acc[m][0][e] + acc[m][1][e]);
        if (l15 == 0) part[16 * m + 4 * kq + e] = v;
      }
  }
}

// The bf16 configuration's form: the producer stored bf16 tiles (store_acc_image_bf16: the same [16 r][64 lanes] image, 2 KB), so
// a lane's 16 bytes at (its row's register row, 16 hi + 8 g) ARE the B operand of one v_mfma_f32_32x32x16_bf16 (k-slots 8 hi + e <->
// columns j = 16 hi + 8 g + e); the A operand is the same eight rows of the b tile in LDS, rounded to bf16 (nearest even) as it is
// read.  4 MFMAs of 32 cycles per tile instead of 32 of 64: the launch is the dS stream (half the bytes of the fp32 form).
template <int NW>
__global__ __launch_bounds__(NW * 64, 4) void ds_matmul_bf16_kernel(DsMmP p) {
  constexpr int NT = NW * 64;
  constexpr int NPF = (512 + NT - 1) / NT;                // float4 per thread per 32 x 64 b tile
  __shared__ __attribute__((aligned(16))) float Bs[2][32 * 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hi = lane >> 5;
  int zh, blk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, blk)) return;
  if (p.reverse) zh = p.ZH - 1 - zh;
  const int h = zh % p.H, z = zh / p.H;
  const int i0 = (blk * NW + wave) * 32;
  const float* bb = p.b + (long long)(z ^ p.b_xor) * NTOK * p.ldb + h * 64;
  const unsigned short* arow = reinterpret_cast<const unsigned short*>(p.ds) + ((long long)zh * NTILE + (i0 >> 5)) * NTILE * 1024 +
                               ((l31 & 3) + 4 * (l31 >> 3)) * 64 + 32 * ((l31 >> 2) & 1) + 16 * hi;
  f32x16 o0 = zero16(), o1 = zero16();
  float4 bpre[NPF];
  uint4 a0[2], a1[2], a2[2];
  auto aload = [&](uint4 (&a)[2], int t) {
    a[0] = *reinterpret_cast<const uint4*>(arow + 1024 * t);
    a[1] = *reinterpret_cast<const uint4*>(arow + 1024 * t + 8);
  };
  auto bload = [&](int t) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      int f = tid + NT * j;
      if (512 % NT != 0) f = min(f, 511);
      bpre[j] = ld4(bb + (long long)(t * 32 + (f >> 4)) * p.ldb + (f & 15) * 4);
    }
  };
  auto bstore = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      int f = tid + NT * j;
      if (512 % NT != 0) f = min(f, 511);
      st4(dst + f * 4, bpre[j]);
    }
  };
  bload(0);
  aload(a0, 0);
  aload(a1, 1);
  bstore(Bs[0]);
  __syncthreads();
  bload(1);
  auto step = [&](const uint4 (&a)[2], uint4 (&anext)[2], int t) {
    const float* B = Bs[t & 1];
    if (t + 2 < NTILE) aload(anext, t + 2);
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float v0[8], v1[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float* br = B + (16 * hi + 8 * g + e) * 64 + l31;
        v0[e] = br[0];
        v1[e] = br[32];
      }
      const bf16x8 dsf = __builtin_bit_cast(bf16x8, a[g]);
      o0 = mfma_bf(pack8(v0), dsf, o0);
      o1 = mfma_bf(pack8(v1), dsf, o1);
    }
    if (t + 1 < NTILE) bstore(Bs[(t & 1) ^ 1]);
    __syncthreads();
    if (t + 2 < NTILE) bload(t + 2);
  };
  for (int t = 0; t < NTILE; t += 3) {
    step(a0, a2, t);
    step(a1, a0, t + 1);
    step(a2, a1, t + 2);
  }
  store_ownerT(p.out + ((long long)z * NTOK + i0 + l31) * p.ldo + h * 64, hi, o0, o1, 1.0f);
  if (p.colpart) colsum_ownerT(p.colpart + ((long long)z * NTILE + (i0 >> 5)) * p.ldp + h * 64, l31, hi, o0, o1, 1.0f);
}

// The same product on v_mfma_f32_16x16x4_f32.  Why: with >= 2 waves per SIMD taking turns MFMA by MFMA (what a short LDS-fed loop at
// 4-5 waves per SIMD does), v_mfma_f32_32x32x2_f32 issues every ~82 cycles instead of 64 (profiles/README.md "measured ceilings":
// 103-123 TF at 2-8 waves per SIMD against 148 TF alone), while 16x16x4 holds 152-155 TF at every occupancy -- the 32x32 form above
// runs 167 us with the dS loads ablated, against 104 us of MFMA work.
//   out^T[d][i] += b[j][d] ds[i][j]:  A[m = d][k = j] = b  (LDS; lane (d = l & 15, kq = l >> 4) reads b[8 kq + s][16 n + d] at k-step s)
//                                     B[k = j][n = i] = ds (registers; lane (i = l & 15, kq) holds ds[i][8 kq .. 8 kq + 7], 2 x 16 bytes)
//   C tile (n = d block, m = i block): lane (i = l & 15, dq = l >> 4), register e = out[i][16 n + 4 dq + e]: 16-byte stores.
// LDS image of a b tile: row stride 66 floats (8-byte aligned rows, staged with ds_write_b64): rows 8 apart -- the two kq groups of a
// 32-lane half -- sit 16 banks apart (conflict-free ds_read_b32 with immediate offsets).
constexpr int BST = 66;     // 8 rows apart = 16 banks apart: the two kq groups of a 32-lane half never share a bank
template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void ds_matmul16_kernel(DsMmP p) {
  constexpr int NT = NW * 64;
  constexpr int NPF = (512 + NT - 1) / NT;
  __shared__ __attribute__((aligned(16))) float Bs[2][32 * BST];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l15 = lane & 15, kq = lane >> 4;
  int zh, blk;
  if (!xcd_problem(NTILE / NW, p.ZH, zh, blk)) return;
  if (p.reverse) zh = p.ZH - 1 - zh;
  const int h = zh % p.H, z = zh / p.H;
  const int i0 = (blk * NW + wave) * 32;
  const float* bb = p.b + (long long)(z ^ p.b_xor) * NTOK * p.ldb + h * 64;
  // row i = 16 m + l15 of the tile lives in register r = (i & 3) + 4 (i >> 3), half-wave hh = (i >> 2) & 1 of the producer's image;
  // m = 1 adds 16 to i: r + 8, same hh
  const float* arow = p.ds + ((long long)zh * NTILE + (i0 >> 5)) * NTILE * 1024 + ((l15 & 3) + 4 * (l15 >> 3)) * 64 + 32 * ((l15 >> 2) & 1) + 8 * kq;

  f32x4 acc[2][4];
#pragma unroll
  for (int m = 0; m < 2; ++m)
#pragma unroll
    for (int n = 0; n < 4; ++n) acc[m][n] = f32x4{0.f, 0.f, 0.f, 0.f};
  float4 bpre[NPF];
  float4 a0[4], a1[4], a2[4];      // [2 m][2 halves of 8 floats]
  auto aload = [&](float4 (&a)[4], int t) {
#pragma unroll
    for (int m = 0; m < 2; ++m) {
      a[2 * m] = ld4(arow + 1024 * t + 512 * m);
      a[2 * m + 1] = ld4(arow + 1024 * t + 512 * m + 4);
    }
  };
  auto bstore = [&](float* dst) {
#pragma unroll
    for (int j = 0; j < NPF; ++j) {
      int f = tid + NT * j;
      if (512 % NT != 0) f = min(f, 511);
      float2* d2 = reinterpret_cast<float2*>(dst + (f >> 4) * BST + (f & 15) * 4);
      d2[0] = make_float2(bpre[j].x, bpre[j].y);
      d2[1] = make_float2(bpre[j].z, bpre[j].w);
    }
  };
  tile_gload<NT>(bb, p.ldb, tid, bpre);
  aload(a0, 0);
  aload(a1, 1);
  bstore(Bs[0]);
  __syncthreads();
  tile_gload<NT>(bb + (long long)32 * p.ldb, p.ldb, tid, bpre);
  const int brow = 8 * kq * BST + l15;        // this lane's column of LDS row j = 8 kq (k-step s adds s rows)
  auto step = [&](const float4 (&a)[4], float4 (&anext)[4], int t) {
    const float* B = Bs[t & 1];
    if (t + 2 < NTILE) aload(anext, t + 2);
    const float av[2][8] = {{a[0].x, a[0].y, a[0].z, a[0].w, a[1].x, a[1].y, a[1].z, a[1].w},
                            {a[2].x, a[2].y, a[2].z, a[2].w, a[3].x, a[3].y, a[3].z, a[3].w}};
#pragma unroll
    for (int s_ = 0; s_ < 8; ++s_) {
      const float* br = B + brow + s_ * BST;
#pragma unroll
      for (int n = 0; n < 4; ++n) {
        const float bv = br[16 * n];
        acc[0][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av[0][s_], acc[0][n], 0, 0, 0);
        acc[1][n] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv, av[1][s_], acc[1][n], 0, 0, 0);
      }
    }
    if (t + 1 < NTILE) bstore(Bs[(t & 1) ^ 1]);
    __syncthreads();
    if (t + 2 < NTILE) tile_gload<NT>(bb + (long long)(t + 2) * 32 * p.ldb, p.ldb, tid, bpre);
  };
  for (int t = 0; t < NTILE; t += 3) {
    step(a0, a2, t);
    step(a1, a0, t + 1);
    step(a2, a1, t + 2);
  }
#pragma unroll
  for (int m = 0; m < 2; ++m) {
    float* orow = p.out + ((long long)z * NTOK + i0 + 16 * m + l15) * p.ldo + h * 64 + 4 * kq;
#pragma unroll
    for (int n = 0; n < 4; ++n) st4(orow + 16 * n, make_float4(acc[m][n][0], acc[m][n][1], acc[m][n][2], acc[m][n][3]));
  }
}

// clse[zh][j] = ln sum_i exp(S[i][j]) from the 18 per-owner-block partials (log2 units: max m_b, l_b = sum exp2(s - m_b))
__global__ __launch_bounds__(256) void colstats_finalize_kernel(const float2* part, float* clse, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long long zh = idx / NTOK;
  const int j = (int)(idx % NTOK);
  const float2* pp = part + zh * NTILE * NTOK + j;
  float2 v[NTILE];
  float M = -INFINITY;
#pragma unroll
  for (int b = 0; b < NTILE; ++b) {
    v[b] = pp[(long long)b * NTOK];
    M = fmaxf(M, v[b].x);
  }
  float L = 0.f;
#pragma unroll
  for (int b = 0; b < NTILE; ++b) L += v[b].y * fast_exp2(v[b].x - M);
  clse[idx] = M * RP_LN2 + logf(L);
}

}  // namespace

// Row and column log-sum-exp of the EMM's score matrix S_z = scale q_{z^1} k_z^T (vision_transformer.py:205-206, the dual softmax's
// two normalisers): rlse[z][h][i] over keys j, clse[z][h][j] over queries i.  fp32: ONE pass over S (rows online, columns from per-block
// partials in `workspace`, rp_emm_stats_workspace_bytes) + a finalize launch; bf16 != 0: the two stats_only passes of rp_attn_fwd.
extern "C" size_t rp_emm_stats_workspace_bytes(int Z, int H) { return (size_t)Z * H * NTILE * NTOK * 2 * sizeof(float); }
// s_out (fp32 only, NULL = off): [Z][H][18 query blocks][18 key tiles][1024] -- the score tiles themselves (log2 units: scale log2(e) q.k),
// element (query i, key j) of a tile at float ((j >> 2) * 32 + i) * 4 + (j & 3) (store_tile_runs): with 288 GB of HBM the three later
// passes over S (rp_emm_apply forward and swap, rp_emm_grad_ds) read these 4 MB per image instead of recomputing q k^T.
extern "C" int rp_emm_stats(const float* q, const float* k, float* rlse, float* clse, void* workspace, float* s_out, int Z, int H,
                            int ldq, int ldk, float scale, int bf16, void* stream) {
  if (Z <= 0 || H <= 0 || (Z & 1) || !q || !k || !rlse || !clse) return RP_EBADSHAPE;
  if ((ldq | ldk) & 3) return RP_EALIGN;
  hipStream_t st = (hipStream_t)stream;
  const dim3 g3(xcd_grid(NTILE / 3, Z * H));
  if (bf16) {
    if (s_out) return RP_EUNSUPPORTED;
    AttnP a{q, k, nullptr, nullptr, rlse, H, ldq, ldk, 4, 4, 1, 0, scale, Z * H, nullptr};
    hipLaunchKernelGGL((attn_fwd_kernel<3, true, 2, true>), g3, dim3(192), 0, st, a);
    AttnP b{k, q, nullptr, nullptr, clse, H, ldk, ldq, 4, 4, 0, 1, scale, Z * H, nullptr};
    hipLaunchKernelGGL((attn_fwd_kernel<3, true, 2, true>), g3, dim3(192), 0, st, b);
    RP_CHECK_LAUNCH();
    return RP_OK;
  }
  if (!workspace) return RP_EBADSHAPE;
  AttnP a{q, k, nullptr, nullptr, rlse, H, ldq, ldk, 4, 4, 1, 0, scale, Z * H, (float*)workspace, s_out, nullptr};
  if (s_out) hipLaunchKernelGGL((attn_fwd_kernel<3, true, 2, false, true, true>), g3, dim3(192), 0, st, a);
  else hipLaunchKernelGGL((attn_fwd_kernel<3, true, 2, false, true>), g3, dim3(192), 0, st, a);
  RP_CHECK_LAUNCH();
  const long long total = (long long)Z * H * NTOK;
  hipLaunchKernelGGL(colstats_finalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, (const float2*)workspace, clse, total);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int Z, int H, int ldq,
                           int ldk, int ldv, int ldo, int q_xor, int k_xor, float scale, int stats_only, int bf16, void* stream) {
  if (Z <= 0 || H <= 0) return RP_EBADSHAPE;
  if ((q_xor & ~1) || (k_xor & ~3)) return RP_EBADSHAPE;   // k_xor: bit 0 = K from the partner image, bit 1 = V
  if ((q_xor || k_xor) && (Z & 1)) return RP_EBADSHAPE;
  if ((ldq | ldk) & 3) return RP_EALIGN;
  if (!stats_only && ((ldv | ldo) & 3)) return RP_EALIGN;
  AttnP p{q, k, v, o, lse, H, ldq, ldk, ldv, ldo, q_xor, k_xor, scale, Z * H, nullptr};
  hipStream_t st = (hipStream_t)stream;
  static const char* const ov = getenv("RP_ATTN_FWD");   // tuning aid: "<NW><WPS>", e.g. "32"
  // few problems (small batches): one-wave workgroups -- with <= two 2-wave workgroups per CU every workgroup's 18-tile
  // loop runs alone on its SIMDs and the launch takes one loop latency; twice as many half-size workgroups interleave (12 images:
  // 78 -> 52 us, dK/dV pass 156 -> 104 us)
  const bool few = Z * H * (NTILE / 2) <= 512;        // (32 images, 864 two-wave workgroups: the two-wave form is 12-20 % faster again)
  const int nw = ov ? ov[0] - '0' : (few ? 1 : 2), wps = ov ? ov[1] - '0' : 2;
  const dim3 g3(xcd_grid(NTILE / 3, Z * H)), g2(xcd_grid(NTILE / 2, Z * H));
  if (bf16) {
    if (stats_only) hipLaunchKernelGGL((attn_fwd_kernel<3, true, 2, true>), g3, dim3(192), 0, st, p);
    else hipLaunchKernelGGL((attn_fwd_kernel<2, false, 2, true>), g2, dim3(128), 0, st, p);
  } else if (stats_only) hipLaunchKernelGGL((attn_fwd_kernel<3, true, 2, false>), g3, dim3(192), 0, st, p);
  else if (nw == 1) hipLaunchKernelGGL((attn_fwd_kernel<1, false, 2, false>), dim3(xcd_grid(NTILE, Z * H)), dim3(64), 0, st, p);
  else if (nw == 2) hipLaunchKernelGGL((attn_fwd_kernel<2, false, 2, false>), g2, dim3(128), 0, st, p);
  else if (wps == 3) hipLaunchKernelGGL((attn_fwd_kernel<3, false, 3, false>), g3, dim3(192), 0, st, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<3, false, 2, false>), g3, dim3(192), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// Training forward of the stored-P backward (exact fp32): rp_attn_fwd plus pst [Z][H][18][18][32][32] (un-normalised probabilities of every
// 32 x 32 tile, store_tile_runs' layout) and mrun [Z][H][18][576] (the running maxima they are relative to)
extern "C" int rp_attn_fwd_savep(const float* q, const float* k, const float* v, float* o, float* lse, float* pst, float* mrun, int Z,
                                 int H, int ldq, int ldk, int ldv, int ldo, float scale, void* stream) {
  if (Z <= 0 || H <= 0 || !q || !k || !v || !o || !lse || !pst || !mrun) return RP_EBADSHAPE;
  if ((ldq | ldk | ldv | ldo) & 3) return RP_EALIGN;
  AttnP p{q, k, v, o, lse, H, ldq, ldk, ldv, ldo, 0, 0, scale, Z * H, nullptr, pst, mrun};
  hipStream_t st = (hipStream_t)stream;
  if (Z * H * (NTILE / 2) <= 512) hipLaunchKernelGGL((attn_fwd_kernel<1, false, 2, false, false, true>), dim3(xcd_grid(NTILE, Z * H)), dim3(64), 0, st, p);
  else hipLaunchKernelGGL((attn_fwd_kernel<2, false, 2, false, false, true>), dim3(xcd_grid(NTILE / 2, Z * H)), dim3(128), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// The dK/dV pass over what rp_attn_fwd_savep stored (attn_bwd_dkdv_p_kernel): dk, dv, and scale * dS in rp_ds_matmul_t's tiles
extern "C" int rp_attn_bwd_dkdv_p(const float* q, const float* v, const float* dout, const float* lse, const float* delta,
                                  const float* pst, const float* mrun, float* dk, float* dv, float* ds, int Z, int H, int ldq, int ldv,
                                  int lddo, int lddk, int lddv, float scale, float* dk_colpart, float* dv_colpart, int ldp, void* stream) {
  if (Z <= 0 || H <= 0 || !q || !v || !dout || !lse || !delta || !pst || !mrun || !dk || !dv || !ds) return RP_EBADSHAPE;
  if ((ldq | ldv | lddo | lddk | lddv) & 3) return RP_EALIGN;
  if ((dk_colpart == nullptr) != (dv_colpart == nullptr) || (dk_colpart && ldp < H * 64)) return RP_EBADSHAPE;
  AttnBwdP p{q, nullptr, v, dout, lse, delta, nullptr, dk, dv, H, ldq, 4, ldv, lddo, 4, lddk, lddv, scale, Z * H, 0, ds,
             dk_colpart, dv_colpart, ldp, pst, mrun};
  hipStream_t st = (hipStream_t)stream;
  if (Z * H * (NTILE / 2) <= 512) hipLaunchKernelGGL((attn_bwd_dkdv_p_kernel<1, 2>), dim3(xcd_grid(NTILE, Z * H)), dim3(64), 0, st, p);
  else hipLaunchKernelGGL((attn_bwd_dkdv_p_kernel<2, 2>), dim3(xcd_grid(NTILE / 2, Z * H)), dim3(128), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// which: 1 = dK,dV pass, 2 = dQ pass, 3 = both (the passes are independent: callers may put them on different streams)
template <int NW, bool BF>
static int launch_bwd(const AttnBwdP& p, int Z, int H, int which, hipStream_t st) {
  dim3 grid(xcd_grid(NTILE / NW, Z * H));
  if (which & 1) {
    hipLaunchKernelGGL((attn_bwd_dkdv_kernel<NW, 2, BF>), grid, dim3(NW * 64), 0, st, p);
    RP_CHECK_LAUNCH();
  }
  if (which & 2) {
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NW, 2, BF>), grid, dim3(NW * 64), 0, st, p);
    RP_CHECK_LAUNCH();
  }
  return RP_OK;
}

static int attn_bwd_impl(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                         const float* delta, float* dq, float* dk, float* dv, int Z, int H, int ldq, int ldk, int ldv,
                         int lddo, int lddq, int lddk, int lddv, float scale, int which, int bf16, void* stream, int kv_xor = 0,
                         float* ds = nullptr, float* dk_colpart = nullptr, float* dv_colpart = nullptr, int ldp = 0) {
  if (Z <= 0 || H <= 0 || (kv_xor & ~1) || (kv_xor && (Z & 1))) return RP_EBADSHAPE;
  if ((ldq | ldk | ldv | lddo | lddq | lddk | lddv) & 3) return RP_EALIGN;
  if ((dk_colpart == nullptr) != (dv_colpart == nullptr) || (dk_colpart && ldp < H * 64)) return RP_EBADSHAPE;
  AttnBwdP p{q, k, v, dout, lse, delta, dq, dk, dv, H, ldq, ldk, ldv, lddo, lddq, lddk, lddv, scale, Z * H, kv_xor, ds,
             dk_colpart, dv_colpart, ldp};
  // both passes need ~190-240 VGPRs (2 waves/SIMD = 8 wave slots per CU): 2-wave workgroups pack 4 per CU, 3-wave ones only 2
  if (bf16) return launch_bwd<2, true>(p, Z, H, which, (hipStream_t)stream);
  static const char* const ov = getenv("RP_ATTN_NW");
  if (ov && ov[0] == '3') return launch_bwd<3, false>(p, Z, H, which, (hipStream_t)stream);
  if ((ov && ov[0] == '1') || (!ov && Z * H * (NTILE / 2) <= 512)) return launch_bwd<1, false>(p, Z, H, which, (hipStream_t)stream);
  return launch_bwd<2, false>(p, Z, H, which, (hipStream_t)stream);
}

extern "C" int rp_attn_bwd(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                           const float* delta, float* dq, float* dk, float* dv, int Z, int H, int ldq, int ldk, int ldv,
                           int lddo, int lddq, int lddk, int lddv, float scale, int bf16, void* stream) {
  return attn_bwd_impl(q, k, v, dout, lse, delta, dq, dk, dv, Z, H, ldq, ldk, ldv, lddo, lddq, lddk, lddv, scale, 3, bf16, stream);
}
extern "C" int rp_attn_bwd_cross(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                                 const float* delta, float* dq, float* dk, float* dv, int Z, int H, int ldq, int ldk, int ldv,
                                 int lddo, int lddq, int lddk, int lddv, float scale, int kv_xor, int bf16, void* stream) {
  return attn_bwd_impl(q, k, v, dout, lse, delta, dq, dk, dv, Z, H, ldq, ldk, ldv, lddo, lddq, lddk, lddv, scale, 3, bf16, stream,
                       kv_xor);
}
extern "C" int rp_attn_bwd_dkdv(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                                const float* delta, float* dk, float* dv, int Z, int H, int ldq, int ldk, int ldv, int lddo,
                                int lddk, int lddv, float scale, int bf16, void* stream) {
  return attn_bwd_impl(q, k, v, dout, lse, delta, nullptr, dk, dv, Z, H, ldq, ldk, ldv, lddo, 4, lddk, lddv, scale, 1, bf16, stream);
}
extern "C" int rp_attn_bwd_dkdv_ds(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                                   const float* delta, float* dk, float* dv, float* ds, int Z, int H, int ldq, int ldk, int ldv,
                                   int lddo, int lddk, int lddv, float scale, int bf16, float* dk_colpart, float* dv_colpart,
                                   int ldp, void* stream) {
  if (!ds) return RP_EBADSHAPE;
  return attn_bwd_impl(q, k, v, dout, lse, delta, nullptr, dk, dv, Z, H, ldq, ldk, ldv, lddo, 4, lddk, lddv, scale, 1, bf16, stream,
                       0, ds, dk_colpart, dv_colpart, ldp);
}
extern "C" int rp_attn_bwd_dq(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                              const float* delta, float* dq, int Z, int H, int ldq, int ldk, int ldv, int lddo, int lddq,
                              float scale, int bf16, void* stream) {
  return attn_bwd_impl(q, k, v, dout, lse, delta, dq, nullptr, nullptr, Z, H, ldq, ldk, ldv, lddo, lddq, 4, 4, scale, 2, bf16, stream);
}

// out[z][i][h*64 + d] = sum_j ds[z*H + h][i][j] * b[z ^ b_xor][j][h*64 + d]: the product that follows a stored-dS pass
// (rp_attn_bwd_dkdv_ds: dQ = ds K;  rp_emm_grad_ds: dK = ds Q of the partner image, b_xor = 1), one launch for all Z*H problems
extern "C" int rp_ds_matmul(const float* ds, const float* b, float* out, int Z, int H, int ldb, int ldo, int b_xor, int ds_bf16,
                            float* colpart, int ldp, void* stream) {
  if (Z <= 0 || H <= 0 || !ds || !b || !out || (b_xor & ~1) || (b_xor && (Z & 1))) return RP_EBADSHAPE;
  if ((ldb | ldo) & 3) return RP_EALIGN;
  if (colpart && ldp < H * 64) return RP_EBADSHAPE;
  static const char* const rv = getenv("RP_DSMM_REV");
  DsMmP p{ds, b, out, H, ldb, ldo, b_xor, Z * H, rv ? rv[0] - '0' : 1, colpart, ldp};
  // RP_DSMM=16: the v_mfma_f32_16x16x4_f32 form (A/B aid; both run ~200 us per 128 images: the stream of fragment-shaped dS reads, not
  // the MFMA form, is what bounds this kernel -- profiles/r3_ds_matmul.txt)
  static const char* const ov = getenv("RP_DSMM");
  hipStream_t st = (hipStream_t)stream;
  if (ds_bf16) hipLaunchKernelGGL((ds_matmul_bf16_kernel<3>), dim3(xcd_grid(NTILE / 3, Z * H)), dim3(192), 0, st, p);
  else if (ov && ov[0] == '1' && !colpart) hipLaunchKernelGGL((ds_matmul16_kernel<3>), dim3(xcd_grid(NTILE / 3, Z * H)), dim3(192), 0, st, p);
  else hipLaunchKernelGGL((ds_matmul_kernel<3>), dim3(xcd_grid(NTILE / 3, Z * H)), dim3(192), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// rp_ds_matmul for tiles stored by store_tile_runs (what rp_attn_bwd_dkdv_p writes)
extern "C" int rp_ds_matmul_t(const float* ds, const float* b, float* out, int Z, int H, int ldb, int ldo, int b_xor, float* colpart,
                              int ldp, void* stream) {
  if (Z <= 0 || H <= 0 || !ds || !b || !out || (b_xor & ~1) || (b_xor && (Z & 1))) return RP_EBADSHAPE;
  if ((ldb | ldo) & 3) return RP_EALIGN;
  if (colpart && ldp < H * 64) return RP_EBADSHAPE;
  static const char* const rv = getenv("RP_DSMM_REV");
  DsMmP p{ds, b, out, H, ldb, ldo, b_xor, Z * H, rv ? rv[0] - '0' : 1, colpart, ldp};
  // RP_DSMM_T=32: the v_mfma_f32_32x32x2_f32 form (A/B aid)
  static const char* const ov = getenv("RP_DSMM_T");
  if (ov && ov[0] == '3') hipLaunchKernelGGL((ds_matmul_t_kernel<3>), dim3(xcd_grid(NTILE / 3, Z * H)), dim3(192), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((ds_matmul_t16_kernel<3>), dim3(xcd_grid(NTILE / 3, Z * H)), dim3(192), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

#ifdef RP_DKDV_PROBE
extern "C" int rp_debug_probe(unsigned long long* out, int reset) {
  unsigned long long z[16] = {0};
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_probe), sizeof(z)) != hipSuccess) return 1;
  if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_probe), z, sizeof(z)) != hipSuccess) return 2;
  return 0;
}
#endif
