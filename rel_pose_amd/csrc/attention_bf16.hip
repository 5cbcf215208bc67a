// attention_bf16.hip -- the bf16 DATA PATH of the fused attention (BASELINE.json configs[4]: "bf16 with MFMA bf16 attention GEMMs").
//
// Same operation as attention.hip (reference Attention.forward, src/modules/vision_transformer.py:325-329, and its autograd), but q / k / v
// / o / dO and the gradients live in HBM as bf16 and reach the matrix pipe without a conversion instruction:
//   * K / V (forward), Q / dO (backward) tiles go global -> LDS by LDS-DMA (global_load_lds_dwordx4: 16 bytes per lane, 8 rows of 64 bf16
//     per wave instruction), two stages, one barrier per stage.  The DMA writes LDS lane-linearly, so a row is 8 unpadded 16-byte chunks
//     whose index is XOR-swizzled on the DMA's SOURCE address and on the reading address.
//   * an operand that is contracted along its contiguous index (K in S = K Q^T: lane = key, 8 consecutive d) is ONE ds_read_b128 =
//     one v_mfma_f32_32x32x16_bf16 operand;
//   * an operand that is contracted along its ROW index (V in O^T = V^T P: lane = d, 8 keys) comes from ds_read_b64_tr_b16, the gfx950
//     transpose read (within a 16-lane group lane t points at 4 contiguous elements of row t >> 2, columns 4 (t & 3)..; it receives
//     column t of the four rows -- tools/lab/tr_probe.hip): two reads per operand, no packing;
//   * P is packed once per tile with v_cvt_pk_bf16_f32 straight from the score accumulators (the S^T accumulators ARE the B operand
//     of the second product: register r of half-wave hi is key acc_row(r, hi), and the k-slot <-> key assignment of that MFMA is chosen
//     to match: slot j of half hi in step c = key 16 c + 8 (j >> 2) + 4 hi + (j & 3));
//   * the softmax row sum is a third MFMA against a ones operand (the matrix pipe has the slack, the VALU does not: at bf16 rates a tile's
//     32 x 32 exponentials cost as much issue time as its 8 MFMAs), scores are scaled inside the exponent's FMA (exp2(s c - m)), and the
//     running maximum is only raised -- and the accumulators rescaled -- when some row's maximum grows by more than 2^8 (wave-uniform
//     branch; P stays <= 256, exact in the fp32 accumulation).
// A workgroup of NW waves (32 query rows each) shares every K / V stage, so the L2 -> LDS traffic of a launch is (18 / NW) x the K / V
// bytes: NW = 6 (192 queries per workgroup, 96-key stages) is the product form; NW = 2 serves launches too small to fill the chip.
#include "bf16_path.h"
#include "../../include/relpose_hip.h"
#include <stdlib.h>

namespace {
using namespace bf16path;

struct AttnBfP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v;
  bf16_t* o; float* lse;
  int H, ldq, ldk, ldv, ldo, q_xor, k_xor;
  float scale;
  int ZH;
};

// GR > 1: a workgroup carries GR independent NW-wave groups (consecutive row-block problems; separate K / V stages, shared barriers).
// The hardware places the waves of a workgroup on the four SIMDs round-robin starting from the same SIMD for every workgroup, so
// two resident 6-wave workgroups load the SIMDs 4 / 4 / 2 / 2 (and do not fit at all at 3 waves per SIMD): ONE 12-wave workgroup of
// two groups loads them 3 / 3 / 3 / 3 (PMC: 1.34 resident waves per SIMD with GR = 1 at 130 VGPRs).
template <int NW, bool STATS, int GR>
__global__ __launch_bounds__(NW * 64 * GR, (NW * GR + 3) / 4 >= 3 ? 3 : 4) void attn_fwd_bf16_kernel(AttnBfP p) {
  constexpr int SK = 16 * NW;                // keys per stage: every wave moves 2 K pieces + 2 V pieces (8 rows each) per stage
  constexpr int NSTAGE = NTOK / SK;
  constexpr int ST_EL = SK * 64;             // bf16 elements of one operand's stage
  static_assert(NTOK % SK == 0 && SK % 32 == 0, "stage must be whole 32-key tiles");
  __shared__ __attribute__((aligned(16))) bf16_t Ks_[GR][2][ST_EL];
  __shared__ __attribute__((aligned(16))) bf16_t Vs_[GR][STATS ? 1 : 2][STATS ? 8 : ST_EL];
  const int tid = threadIdx.x, lane = tid & 63, wave_wg = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int grp = wave_wg / NW, wave = wave_wg % NW;
  bf16_t (*Ks)[ST_EL] = Ks_[grp];
  bf16_t (*Vs)[STATS ? 8 : ST_EL] = Vs_[grp];
  int zh, qblk;
  {
    // XCD-aware order over (problem, row block) units, GR consecutive units per workgroup (see common.h: xcd_problem)
    constexpr int NQ = 18 / NW;
    const int j = (blockIdx.x >> 3) * GR + grp;
    zh = (j / NQ) * 8 + (blockIdx.x & 7);
    qblk = j % NQ;
    if (zh >= p.ZH) {                               // padding group: still has to meet the workgroup's barriers
      if (GR == 1) return;
      zh = p.ZH - 1;
    }
  }
  const int h = zh % p.H, z = zh / p.H;
  const int q0 = (qblk * NW + wave) * 32;
  const bf16_t* qb = p.q + (long long)(z ^ p.q_xor) * NTOK * p.ldq + h * 64;
  const bf16_t* kb = p.k + (long long)(z ^ (p.k_xor & 1)) * NTOK * p.ldk + h * 64;
  const bf16_t* vb = STATS ? nullptr : p.v + (long long)(z ^ (p.k_xor >> 1)) * NTOK * p.ldv + h * 64;

  // DMA plan: piece i (0, 1) of this wave covers stage rows (2 wave + i) * 8 + (lane >> 3); lane -> LDS slot lane & 7 of that row
  unsigned kvoff[2], vvoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (2 * wave + i) * 8 + (lane >> 3);
    kvoff[i] = (unsigned)(r * p.ldk + (((lane & 7) ^ swz_k(r)) << 3)) * 2u;
    vvoff[i] = (unsigned)(r * p.ldv + (((lane & 7) ^ swz_v(r)) << 3)) * 2u;
  }
  const unsigned ks0 = lds_addr_of(&Ks[0][0]) + wave * 2048, vs0 = lds_addr_of(&Vs[0][0]) + wave * 2048;
  auto issue = [&](int s, int buf) {
    const void* kk = uniform_vptr(kb + (long long)s * SK * p.ldk);
    glds16b(kk, kvoff[0], ks0 + buf * (ST_EL * 2));
    glds16b(kk, kvoff[1], ks0 + buf * (ST_EL * 2) + 1024);
    if (!STATS) {
      const void* vv = uniform_vptr(vb + (long long)s * SK * p.ldv);
      glds16b(vv, vvoff[0], vs0 + buf * (ST_EL * 2));
      glds16b(vv, vvoff[1], vs0 + buf * (ST_EL * 2) + 1024);
    }
  };

  issue(0, 0);

  // Q as the B operand of S^T = K Q^T: lane (query l31, half hi) holds d = 16 c + 8 hi .. + 7 for c = 0..3 (raw bf16, unscaled)
  bf16x8 qpk[4];
  {
    const bf16_t* qr = qb + (long long)(q0 + l31) * p.ldq + 8 * hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) qpk[c] = *reinterpret_cast<const bf16x8*>(qr + 16 * c);
  }
  const float cs = p.scale * RP_LOG2E;            // scores enter the exponent as s * cs (log2 units)

  // per-lane LDS offsets (elements).  K: row l31 of a tile, chunk (2 c + hi) ^ swz_k(l31).  V: transpose-read blocks, see header.
  int koff[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) koff[c] = l31 * 64 + (((2 * c + hi) ^ swz_k(l31)) << 3);
  const int t16 = lane & 15, g = (lane >> 4) & 1;
  // row (4 hi + (t16 >> 2)) of an 8-row group, columns 16 g + 4 (t16 & 3) (+ 32 dblk): chunk 2 g + ((t16 & 3) >> 1) (+ 4 dblk), the
  // V swizzle of that row is ((t16 >> 3) & 1) << 2 (all other row terms are multiples of 4)
  int voff[2];
#pragma unroll
  for (int db = 0; db < 2; ++db)
    voff[db] = (4 * hi + (t16 >> 2)) * 64 + (((2 * g + ((t16 & 3) >> 1) + 4 * db) ^ (((t16 >> 3) & 1) << 2)) << 3) + 4 * (t16 & 1);

  f32x16 o0 = zero16(), o1 = zero16(), ls = zero16();
  float m = -1e30f;
  const bf16x8 one = ones8();

  // (A software-pipelined form of this loop -- S(T+1) issued under the softmax of tile T, K / V fragments read a tile ahead, a
  // three-slot LDS ring with the DMA a whole stage ahead -- was built and measured SLOWER, 129-147 us against 104-120 us at 256 images:
  // on gfx950 a wave's VALU issue time ADDS to its MFMA time (PMC: VALU-active 397 + MFMA-busy 320 cycles per tile and SIMD of the
  // 1000-1300 spent), so hiding latencies inside one wave buys nothing that 3-4 resident waves do not already hide, and the extra
  // accumulator set costs occupancy.  What bounds this kernel is the sum of the two pipes' work, not their overlap.)
  for (int s = 0; s < NSTAGE; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of stage s have landed (and its LDS reads returned) ...
    __builtin_amdgcn_s_barrier();                         // ... and everybody's; everybody is also done reading the other buffer
    if (s + 1 < NSTAGE) issue(s + 1, buf ^ 1);
    const bf16_t* Kt = Ks[buf];
    const bf16_t* Vt = STATS ? nullptr : Vs[buf];
#pragma unroll
    for (int j = 0; j < SK / 32; ++j) {
      f32x16 sc = zero16();
#pragma unroll
      for (int c = 0; c < 4; ++c) sc = mfma_bf(ld_bf16x8_lds(Kt + j * 2048 + koff[c]), qpk[c], sc);
      // row maximum (raw units), both halves
      float mx = fmaxf(fmaxf(sc[0], sc[1]), sc[2]);
#pragma unroll
      for (int r = 3; r < 15; r += 2) mx = fmaxf(fmaxf(mx, sc[r]), sc[r + 1]);
      mx = fmaxf(mx, sc[15]);
      {
        const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, mx), __builtin_bit_cast(unsigned, mx), false, false);
        mx = fmaxf(__builtin_bit_cast(float, sw[0]), __builtin_bit_cast(float, sw[1])) * cs;      // both halves: max over the 32 keys
      }
      if (__builtin_amdgcn_ballot_w64(mx > m + 8.0f) != 0) {            // wave-uniform: somebody's maximum outgrew the headroom
        const float mn = fmaxf(m, mx);
        const float alpha = fast_exp2(m - mn);
        m = mn;
        ls[0] *= alpha;
        if (!STATS) {
#pragma unroll
          for (int r = 0; r < 16; ++r) { o0[r] *= alpha; o1[r] *= alpha; }
        }
      }
      const float nm = -m;
#pragma unroll
      for (int r = 0; r < 16; ++r) sc[r] = fast_exp2(fmaf(sc[r], cs, nm));
      const bf16x8 p0 = pack8(sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[7]);
      const bf16x8 p1 = pack8(sc[8], sc[9], sc[10], sc[11], sc[12], sc[13], sc[14], sc[15]);
      ls = mfma_bf(one, p0, ls);
      ls = mfma_bf(one, p1, ls);
      if (!STATS) {
        const bf16_t* vt = Vt + j * 2048;
        o0 = mfma_bf(tr_operand(vt + voff[0], vt + 512 + voff[0]), p0, o0);
        o1 = mfma_bf(tr_operand(vt + voff[1], vt + 512 + voff[1]), p0, o1);
        o0 = mfma_bf(tr_operand(vt + 1024 + voff[0], vt + 1536 + voff[0]), p1, o0);
        o1 = mfma_bf(tr_operand(vt + 1024 + voff[1], vt + 1536 + voff[1]), p1, o1);
      }
    }
  }
  const float lt = ls[0];                          // every row of the ones-product is the full sum over the 16 k-slots of each step
  // lse2 = log2 sum_j exp2(s_ij cs): the LOG2-domain normaliser (natural-log lse = lse2 ln 2); the backward kernels below consume it as is
  if (hi == 0) p.lse[((long long)z * p.H + h) * NTOK + q0 + l31] = m + __builtin_amdgcn_logf(lt);
  if (STATS) return;
  // O^T accumulators (lane = query, register = d) -> [32 q][64 d] bf16 through this wave's 4 KB of LDS -> whole 128-byte rows
  __builtin_amdgcn_s_barrier();                    // every wave is done with the stage buffers
  store_ownerT_bf16(&Ks[0][0] + wave * 2048, p.o + ((long long)z * NTOK + q0) * p.ldo + h * 64, p.ldo, lane, o0, o1, 1.0f / lt);
}

template <int NW, bool STATS, int GR = 1>
int launch_fwd(const AttnBfP& p, hipStream_t st) {
  const int units = (18 / NW) * ((p.ZH + 7) / 8);            // per XCD
  hipLaunchKernelGGL((attn_fwd_bf16_kernel<NW, STATS, GR>), dim3(((units + GR - 1) / GR) * 8), dim3(NW * 64 * GR), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}


// ------------------------------------------------------------------------------------------------
// backward on the bf16 data path: RECOMPUTE form, two deterministic kernels, no stored dS (at bf16 MFMA rates the two extra score
// products of the second pass cost less than writing and re-reading 576 x 576 x 2 bytes per head: ~1 GB per layer at 128 pairs).
//   P = exp2(S cs - lse2);  dP = dO V^T;  dS = P o (dP - delta);  dV = P^T dO;  dK = scale dS^T Q;  dQ = scale dS K
//   dkdv: a wave owns 32 keys (K, V rows as MFMA B operands in VGPRs) and streams {Q, dO} stages; S, dP: A = Q / dO rows (ds_read_b128);
//         dV^T += dO^T P, dK^T += Q^T dS: A = the SAME LDS tiles read by ds_read_b64_tr_b16, B = P / dS packed from the accumulators.
//   dq  : a wave owns 32 queries (Q, dO rows in VGPRs, lse2 / delta lane-local) and streams {K, V} stages; S^T, dP^T: A = K / V rows;
//         dQ^T += K^T dS^T: A = the K tile by transpose reads.
// Tiles that are read both ways use the swizzle swz_d: (row >> 1) & 7 bit-reversed, so that the four rows of a transpose-read block
// differ in the 64-byte-half bit as well as in row parity.
// ------------------------------------------------------------------------------------------------
struct AttnBwdBfP {
  const bf16_t* q; const bf16_t* k; const bf16_t* v; const bf16_t* dout;
  const float* lse2; const float* delta;
  bf16_t* dq; bf16_t* dk; bf16_t* dv;
  int H, ldq, ldk, ldv, lddo, lddq, lddk, lddv;
  float scale;
  int ZH, kv_xor;
  float *dq_colpart, *dk_colpart, *dv_colpart;   // optional [Z * 18][ldp] column sums per 32-row block (first of the H*64 columns)
  int ldp;
};

template <int NW>
__global__ __launch_bounds__(NW * 64, NW == 2 ? 3 : 2) void attn_bwd_dkdv_bf16_kernel(AttnBwdBfP p) {
  constexpr int SQ = 16 * NW;                // queries per stage
  constexpr int NSTAGE = NTOK / SQ;
  constexpr int ST_EL = SQ * 64;
  __shared__ __attribute__((aligned(16))) bf16_t Qs[2][ST_EL];
  __shared__ __attribute__((aligned(16))) bf16_t Ds[2][ST_EL];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  int zh, kblk;
  if (!xcd_problem(18 / NW, p.ZH, zh, kblk)) return;
  const int h = zh % p.H, z = zh / p.H;          // z indexes the key / value image
  const int k0 = (kblk * NW + wave) * 32;
  const int zq = z ^ p.kv_xor;
  const bf16_t* qb = p.q + (long long)zq * NTOK * p.ldq + h * 64;
  const bf16_t* dob = p.dout + (long long)zq * NTOK * p.lddo + h * 64;
  const float* lseb = p.lse2 + ((long long)zq * p.H + h) * NTOK;
  const float* delb = p.delta + ((long long)zq * p.H + h) * NTOK;

  unsigned qvoff[2], dvoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (2 * wave + i) * 8 + (lane >> 3);
    qvoff[i] = (unsigned)(r * p.ldq + (((lane & 7) ^ swz_d(r)) << 3)) * 2u;
    dvoff[i] = (unsigned)(r * p.lddo + (((lane & 7) ^ swz_d(r)) << 3)) * 2u;
  }
  const unsigned qs0 = lds_addr_of(&Qs[0][0]) + wave * 2048, ds0 = lds_addr_of(&Ds[0][0]) + wave * 2048;
  auto issue = [&](int s, int buf) {
    const void* qq = uniform_vptr(qb + (long long)s * SQ * p.ldq);
    const void* dd = uniform_vptr(dob + (long long)s * SQ * p.lddo);
    glds16b(qq, qvoff[0], qs0 + buf * (ST_EL * 2));
    glds16b(qq, qvoff[1], qs0 + buf * (ST_EL * 2) + 1024);
    glds16b(dd, dvoff[0], ds0 + buf * (ST_EL * 2));
    glds16b(dd, dvoff[1], ds0 + buf * (ST_EL * 2) + 1024);
  };
  issue(0, 0);

  // K, V of the owner keys as B operands: lane (key l31, half hi) holds d = 16 c + 8 hi .. + 7
  bf16x8 kpk[4], vpk[4];
  {
    const bf16_t* kr = p.k + ((long long)z * NTOK + k0 + l31) * p.ldk + h * 64 + 8 * hi;
    const bf16_t* vr = p.v + ((long long)z * NTOK + k0 + l31) * p.ldv + h * 64 + 8 * hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      kpk[c] = *reinterpret_cast<const bf16x8*>(kr + 16 * c);
      vpk[c] = *reinterpret_cast<const bf16x8*>(vr + 16 * c);
    }
  }
  const float cs = p.scale * RP_LOG2E;
  int roff[4];                                   // b128 rows: query row l31 of a tile, chunk (2 c + hi) ^ swz_d(l31)
#pragma unroll
  for (int c = 0; c < 4; ++c) roff[c] = l31 * 64 + (((2 * c + hi) ^ swz_d(l31)) << 3);
  int toff[2][2];
  tr_offsets_d(lane, toff);

  f32x16 dv0 = zero16(), dv1 = zero16(), dk0 = zero16(), dk1 = zero16();
  for (int s = 0; s < NSTAGE; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < NSTAGE) issue(s + 1, buf ^ 1);
#pragma unroll
    for (int j = 0; j < SQ / 32; ++j) {
      const bf16_t* Qt = Qs[buf] + j * 2048;
      const bf16_t* Dt = Ds[buf] + j * 2048;
      // lse2 / delta of the tile's queries: register r of half hi is query acc_row(r, hi) -> four runs of 4 consecutive queries
      const float* lq = lseb + s * SQ + j * 32 + 4 * hi;
      const float* dq_ = delb + s * SQ + j * 32 + 4 * hi;
      float4 l4[4], d4[4];
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) { l4[g4] = ld4(lq + 8 * g4); d4[g4] = ld4(dq_ + 8 * g4); }
      f32x16 sc = zero16(), dp = zero16();
#pragma unroll
      for (int c = 0; c < 4; ++c) sc = mfma_bf(ld_bf16x8_lds(Qt + roff[c]), kpk[c], sc);        // S[q][kv]
#pragma unroll
      for (int c = 0; c < 4; ++c) dp = mfma_bf(ld_bf16x8_lds(Dt + roff[c]), vpk[c], dp);        // dP[q][kv]
#pragma unroll
      for (int g4 = 0; g4 < 4; ++g4) {
        const float lv[4] = {l4[g4].x, l4[g4].y, l4[g4].z, l4[g4].w};
        const float dl[4] = {d4[g4].x, d4[g4].y, d4[g4].z, d4[g4].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = 4 * g4 + e;
          const float pr = fast_exp2(fmaf(sc[r], cs, -lv[e]));
          sc[r] = pr;
          dp[r] = pr * (dp[r] - dl[e]);
        }
      }
      const bf16x8 p0 = pack8(sc[0], sc[1], sc[2], sc[3], sc[4], sc[5], sc[6], sc[7]);
      const bf16x8 p1 = pack8(sc[8], sc[9], sc[10], sc[11], sc[12], sc[13], sc[14], sc[15]);
      const bf16x8 s0 = pack8(dp[0], dp[1], dp[2], dp[3], dp[4], dp[5], dp[6], dp[7]);
      const bf16x8 s1 = pack8(dp[8], dp[9], dp[10], dp[11], dp[12], dp[13], dp[14], dp[15]);
      // dV^T[d][kv] += sum_q dO[q][d] P[q][kv];  dK^T[d][kv] += sum_q Q[q][d] dS[q][kv]   (A = transpose reads of the same tiles)
      dv0 = mfma_bf(tr_operand(Dt + toff[0][0], Dt + toff[0][1]), p0, dv0);
      dv1 = mfma_bf(tr_operand(Dt + toff[1][0], Dt + toff[1][1]), p0, dv1);
      dk0 = mfma_bf(tr_operand(Qt + toff[0][0], Qt + toff[0][1]), s0, dk0);
      dk1 = mfma_bf(tr_operand(Qt + toff[1][0], Qt + toff[1][1]), s0, dk1);
      dv0 = mfma_bf(tr_operand(Dt + 1024 + toff[0][0], Dt + 1024 + toff[0][1]), p1, dv0);
      dv1 = mfma_bf(tr_operand(Dt + 1024 + toff[1][0], Dt + 1024 + toff[1][1]), p1, dv1);
      dk0 = mfma_bf(tr_operand(Qt + 1024 + toff[0][0], Qt + 1024 + toff[0][1]), s1, dk0);
      dk1 = mfma_bf(tr_operand(Qt + 1024 + toff[1][0], Qt + 1024 + toff[1][1]), s1, dk1);
    }
  }
  if (p.dk_colpart) {
    const long long prow = ((long long)z * 18 + (k0 >> 5)) * p.ldp + h * 64;
    colsum_ownerT_bf(p.dv_colpart + prow, l31, hi, dv0, dv1, 1.0f);
    colsum_ownerT_bf(p.dk_colpart + prow, l31, hi, dk0, dk1, p.scale);
  }
  __builtin_amdgcn_s_barrier();                  // every wave is done with the stage buffers
  bf16_t* Os = &Qs[0][0] + wave * 2048;
  store_ownerT_bf16(Os, p.dv + ((long long)z * NTOK + k0) * p.lddv + h * 64, p.lddv, lane, dv0, dv1, 1.0f);
  store_ownerT_bf16(Os, p.dk + ((long long)z * NTOK + k0) * p.lddk + h * 64, p.lddk, lane, dk0, dk1, p.scale);
}

template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void attn_bwd_dq_bf16_kernel(AttnBwdBfP p) {
  constexpr int SK = 16 * NW;
  constexpr int NSTAGE = NTOK / SK;
  constexpr int ST_EL = SK * 64;
  __shared__ __attribute__((aligned(16))) bf16_t Ks[2][ST_EL];
  __shared__ __attribute__((aligned(16))) bf16_t Vs[2][ST_EL];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  int zh, qblk;
  if (!xcd_problem(18 / NW, p.ZH, zh, qblk)) return;
  const int h = zh % p.H, z = zh / p.H;          // z indexes the query image
  const int q0 = (qblk * NW + wave) * 32;
  const bf16_t* kb = p.k + (long long)(z ^ p.kv_xor) * NTOK * p.ldk + h * 64;
  const bf16_t* vb = p.v + (long long)(z ^ p.kv_xor) * NTOK * p.ldv + h * 64;

  unsigned kvoff[2], vvoff[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int r = (2 * wave + i) * 8 + (lane >> 3);
    kvoff[i] = (unsigned)(r * p.ldk + (((lane & 7) ^ swz_d(r)) << 3)) * 2u;      // K is read both ways
    vvoff[i] = (unsigned)(r * p.ldv + (((lane & 7) ^ swz_k(r)) << 3)) * 2u;      // V by rows only
  }
  const unsigned ks0 = lds_addr_of(&Ks[0][0]) + wave * 2048, vs0 = lds_addr_of(&Vs[0][0]) + wave * 2048;
  auto issue = [&](int s, int buf) {
    const void* kk = uniform_vptr(kb + (long long)s * SK * p.ldk);
    const void* vv = uniform_vptr(vb + (long long)s * SK * p.ldv);
    glds16b(kk, kvoff[0], ks0 + buf * (ST_EL * 2));
    glds16b(kk, kvoff[1], ks0 + buf * (ST_EL * 2) + 1024);
    glds16b(vv, vvoff[0], vs0 + buf * (ST_EL * 2));
    glds16b(vv, vvoff[1], vs0 + buf * (ST_EL * 2) + 1024);
  };
  issue(0, 0);

  bf16x8 qpk[4], dpk[4];
  {
    const bf16_t* qr = p.q + ((long long)z * NTOK + q0 + l31) * p.ldq + h * 64 + 8 * hi;
    const bf16_t* dr = p.dout + ((long long)z * NTOK + q0 + l31) * p.lddo + h * 64 + 8 * hi;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      qpk[c] = *reinterpret_cast<const bf16x8*>(qr + 16 * c);
      dpk[c] = *reinterpret_cast<const bf16x8*>(dr + 16 * c);
    }
  }
  const float nlse = -p.lse2[((long long)z * p.H + h) * NTOK + q0 + l31];
  const float del = p.delta[((long long)z * p.H + h) * NTOK + q0 + l31];
  const float cs = p.scale * RP_LOG2E;
  int koff[4], vofr[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    koff[c] = l31 * 64 + (((2 * c + hi) ^ swz_d(l31)) << 3);
    vofr[c] = l31 * 64 + (((2 * c + hi) ^ swz_k(l31)) << 3);
  }
  int toff[2][2];
  tr_offsets_d(lane, toff);

  f32x16 dq0 = zero16(), dq1 = zero16();
  for (int s = 0; s < NSTAGE; ++s) {
    const int buf = s & 1;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (s + 1 < NSTAGE) issue(s + 1, buf ^ 1);
#pragma unroll
    for (int j = 0; j < SK / 32; ++j) {
      const bf16_t* Kt = Ks[buf] + j * 2048;
      const bf16_t* Vt = Vs[buf] + j * 2048;
      f32x16 sc = zero16(), dp = zero16();
#pragma unroll
      for (int c = 0; c < 4; ++c) sc = mfma_bf(ld_bf16x8_lds(Kt + koff[c]), qpk[c], sc);        // S^T[kv][q]
#pragma unroll
      for (int c = 0; c < 4; ++c) dp = mfma_bf(ld_bf16x8_lds(Vt + vofr[c]), dpk[c], dp);        // dP^T[kv][q]
#pragma unroll
      for (int r = 0; r < 16; ++r) dp[r] = fast_exp2(fmaf(sc[r], cs, nlse)) * (dp[r] - del);
      const bf16x8 s0 = pack8(dp[0], dp[1], dp[2], dp[3], dp[4], dp[5], dp[6], dp[7]);
      const bf16x8 s1 = pack8(dp[8], dp[9], dp[10], dp[11], dp[12], dp[13], dp[14], dp[15]);
      dq0 = mfma_bf(tr_operand(Kt + toff[0][0], Kt + toff[0][1]), s0, dq0);                      // dQ^T[d][q] += K^T dS^T
      dq1 = mfma_bf(tr_operand(Kt + toff[1][0], Kt + toff[1][1]), s0, dq1);
      dq0 = mfma_bf(tr_operand(Kt + 1024 + toff[0][0], Kt + 1024 + toff[0][1]), s1, dq0);
      dq1 = mfma_bf(tr_operand(Kt + 1024 + toff[1][0], Kt + 1024 + toff[1][1]), s1, dq1);
    }
  }
  if (p.dq_colpart) colsum_ownerT_bf(p.dq_colpart + ((long long)z * 18 + (q0 >> 5)) * p.ldp + h * 64, l31, hi, dq0, dq1, p.scale);
  __builtin_amdgcn_s_barrier();
  store_ownerT_bf16(&Ks[0][0] + wave * 2048, p.dq + ((long long)z * NTOK + q0) * p.lddq + h * 64, p.lddq, lane, dq0, dq1, p.scale);
}

// delta[z][h][i] = sum_e dO[z][i][h*64+e] O[z][i][h*64+e] on bf16 rows (fp32 products and sums): one wave per token row, 3 heads x 64
__global__ __launch_bounds__(256) void attn_delta_bf16_kernel(const bf16_t* dout, const bf16_t* o, float* delta, int H, int ld, long long rows) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (row >= rows) return;
  // 192 columns = 24 lanes x 8 bf16; lanes 24.. idle (a row is 384 bytes)
  float acc = 0.f;
  if (lane < 24) {
    const uint4 a = *reinterpret_cast<const uint4*>(dout + row * ld + 8 * lane), b = *reinterpret_cast<const uint4*>(o + row * ld + 8 * lane);
    const unsigned aw[4] = {a.x, a.y, a.z, a.w}, bw[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      acc = fmaf(__uint_as_float(aw[i] << 16), __uint_as_float(bw[i] << 16), acc);
      acc = fmaf(__uint_as_float(aw[i] & 0xffff0000u), __uint_as_float(bw[i] & 0xffff0000u), acc);
    }
  }
  // head hh = lanes 8 hh .. 8 hh + 7
  acc += __shfl_xor(acc, 1, 64);
  acc += __shfl_xor(acc, 2, 64);
  acc += __shfl_xor(acc, 4, 64);
  if (lane < 24 && (lane & 7) == 0) {
    const long long z = row / NTOK, i = row % NTOK;
    delta[(z * H + (lane >> 3)) * NTOK + i] = acc;
  }
}

template <int NW>
int launch_bwd(const AttnBwdBfP& p, hipStream_t st, int which) {
  if (which & 1) hipLaunchKernelGGL((attn_bwd_dkdv_bf16_kernel<NW>), dim3(xcd_grid(18 / NW, p.ZH)), dim3(NW * 64), 0, st, p);
  if (which & 2) hipLaunchKernelGGL((attn_bwd_dq_bf16_kernel<NW>), dim3(xcd_grid(18 / NW, p.ZH)), dim3(NW * 64), 0, st, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

}  // namespace

extern "C" int rp_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int Z, int H, int ldq, int ldk,
                                int ldv, int ldo, int q_xor, int k_xor, float scale, int stats_only, void* stream) {
  if (!q || !k || !lse || Z <= 0 || H <= 0) return RP_EBADSHAPE;
  if (!stats_only && (!v || !o)) return RP_EBADSHAPE;
  if ((ldq | ldk | ldv | ldo) & 7) return RP_EALIGN;                       // 16-byte row segments
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)o) & 15) return RP_EALIGN;
  if ((q_xor & ~1) || (k_xor & ~3)) return RP_EBADSHAPE;                   // as rp_attn_fwd: k_xor bit 0 = K from the partner image, bit 1 = V
  if ((q_xor | k_xor) && (Z & 1)) return RP_EBADSHAPE;
  AttnBfP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.o = (bf16_t*)o; p.lse = lse;
  p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.ldo = ldo; p.q_xor = q_xor; p.k_xor = k_xor; p.scale = scale; p.ZH = Z * H;
  hipStream_t st = (hipStream_t)stream;
  static const char* const ov = getenv("RP_ATTN_BF16_NW");                 // tuning aid: "2" / "6"
  const bool small = ov ? ov[0] == '2' : true;                             // measured (256 images): 2-wave workgroups 104-108 us, 6-wave 118-120
  const bool one = ov && ov[0] == '6' && ov[1] == '1';                    // "61": 6-wave workgroups of ONE group (A/B aid)
  if (stats_only) return small ? launch_fwd<2, true>(p, st) : one ? launch_fwd<6, true>(p, st) : launch_fwd<6, true, 2>(p, st);
  return small ? launch_fwd<2, false>(p, st) : one ? launch_fwd<6, false>(p, st) : launch_fwd<6, false, 2>(p, st);
}

extern "C" int rp_attn_bwd_delta_bf16(const void* dout, const void* o, float* delta, int Z, int H, int ld, void* stream) {
  if (!dout || !o || !delta || Z <= 0 || H != 3 || (ld & 7)) return RP_EBADSHAPE;
  const long long rows = (long long)Z * NTOK;
  hipLaunchKernelGGL(attn_delta_bf16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const bf16_t*)dout,
                     (const bf16_t*)o, delta, H, ld, rows);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dout, const float* lse2, const float* delta,
                                void* dq, void* dk, void* dv, int Z, int H, int ldq, int ldk, int ldv, int lddo, int lddq, int lddk,
                                int lddv, float scale, int kv_xor, float* dq_colpart, float* dk_colpart, float* dv_colpart, int ldp,
                                void* stream) {
  if (!q || !k || !v || !dout || !lse2 || !delta || !dq || !dk || !dv || Z <= 0 || H <= 0) return RP_EBADSHAPE;
  if ((ldq | ldk | ldv | lddo | lddq | lddk | lddv) & 7) return RP_EALIGN;
  if (((uintptr_t)q | (uintptr_t)k | (uintptr_t)v | (uintptr_t)dout | (uintptr_t)dq | (uintptr_t)dk | (uintptr_t)dv) & 15) return RP_EALIGN;
  if ((kv_xor & ~1) || (kv_xor && (Z & 1))) return RP_EBADSHAPE;       // as rp_attn_bwd: an image index z ^ kv_xor must stay inside the pair
  if ((dk_colpart == nullptr) != (dv_colpart == nullptr)) return RP_EBADSHAPE;
  AttnBwdBfP p;
  p.q = (const bf16_t*)q; p.k = (const bf16_t*)k; p.v = (const bf16_t*)v; p.dout = (const bf16_t*)dout; p.lse2 = lse2; p.delta = delta;
  p.dq = (bf16_t*)dq; p.dk = (bf16_t*)dk; p.dv = (bf16_t*)dv;
  p.H = H; p.ldq = ldq; p.ldk = ldk; p.ldv = ldv; p.lddo = lddo; p.lddq = lddq; p.lddk = lddk; p.lddv = lddv;
  p.scale = scale; p.ZH = Z * H; p.kv_xor = kv_xor;
  p.dq_colpart = dq_colpart; p.dk_colpart = dk_colpart; p.dv_colpart = dv_colpart; p.ldp = ldp;
  static const char* const ov = getenv("RP_ATTN_BF16_BWD_NW");          // tuning aid: "2" / "6"
  if (ov && ov[0] == '6') return launch_bwd<6>(p, (hipStream_t)stream, 3);
  return launch_bwd<2>(p, (hipStream_t)stream, 3);
}
