// gemm.hip -- fp32 MFMA contraction with fused epilogue (rp_gemm).
//
// Replaces nn.Linear forward / dX / dW of the reference's ViT blocks, EMM projection and pose regressor
// (reference src/modules/vision_transformer.py:323,331,191-195,233-234; vit_layers/mlp.py:20-26;
// src/model.py:91-98).  One kernel template, 4 operand layouts x 6 tile shapes.
//
// Tiling (gfx950): 256 threads = 4 waves in a 2x2 grid; each wave owns (32*TM) x (32*TN) of C as TM*TN
// v_mfma_f32_32x32x2_f32 accumulators; BK = 32.  Global -> registers (float4, issued one k-tile ahead so
// HBM/L2 latency hides under the MFMAs of the current tile) -> LDS -> MFMA operands.
// k-step pairing inside a k-tile: step t in [0,16): half-wave 0 supplies k = t, half-wave 1 supplies
// k = 16 + t, so a K-contiguous operand row is read as 4 x ds_read_b128 (row stride 36 floats: the 16
// lanes of a b128 group land on 16 distinct 16-byte slots); an MN-contiguous operand is read as
// conflict-free ds_read_b32 rows.
#include "gemm_common.h"
#include "../../include/relpose_hip.h"
#include <stdlib.h>

namespace {
using namespace rpgemm;

// ---- split-bf16 operand path (NL = 1..3 limbs) ---------------------------------------------------------------------
// x = x1 + x2 + x3 exactly, each limb a bf16 (8 significant bits): x1 = top 8 bits of x (truncation), x2 = top 8 bits of
// x - x1, x3 = the rest.  With NL = 3 the product a*b is accumulated (fp32, inside v_mfma_f32_32x32x16_bf16) as the six
// limb products whose weight is >= 2^-16: a1b1 + a1b2 + a2b1 + a2b2 + a1b3 + a3b1; the three dropped ones are <= 2^-24
// relative -- the size of one fp32 rounding -- so the result is fp32-grade while the contraction runs on the bf16
// matrix pipe (16 k per 32-cycle MFMA instead of 2 k per 64-cycle fp32 MFMA: 16/6 = 2.7x fewer pipe cycles).
// NL = 2 (3 products, ~2^-17 operand truncation) and NL = 1 (plain bf16 operands, the bf16 configuration) share the code.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// packed bf16 pair (lo half = a's limb, hi half = b's limb) by truncation: v_perm_b32 picks the two high halves
RP_DEV unsigned pack_hi16(float a, float b) {
  return __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
}
// single-limb (plain bf16) operands are rounded to nearest-even instead (v_cvt_pk_bf16_f32): no truncation bias
RP_DEV unsigned pack_rne(float a, float b) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
template <int NL>
RP_DEV unsigned pack_limb(float a, float b) { return NL == 1 ? pack_rne(a, b) : pack_hi16(a, b); }
RP_DEV float drop_hi16(float x) {   // x - (top 16 bits of x), exact
  return x - __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & 0xffff0000u);
}

// split NV staged float4 of one operand tile into NL bf16 limb images in LDS (layouts: see gemm_kernel)
template <int LAY, int EXT, int NL, int NV>
RP_DEV void limb_store(unsigned* w, int tid, const float4 (&r)[NV]) {
  constexpr int RSW = 20;
  constexpr int WORDS = LAY == 0 ? EXT * RSW : 16 * (EXT + 4);
  if (LAY == 0) {
#pragma unroll
    for (int j = 0; j < NV; ++j) {
      const int f = tid + 256 * j;
      unsigned* dst = w + (f >> 3) * RSW + (f & 7) * 2;
      float4 v = r[j];
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        *reinterpret_cast<uint2*>(dst + l * WORDS) = make_uint2(pack_limb<NL>(v.x, v.y), pack_limb<NL>(v.z, v.w));
        if (l + 1 < NL) { v.x = drop_hi16(v.x); v.y = drop_hi16(v.y); v.z = drop_hi16(v.z); v.w = drop_hi16(v.w); }
      }
    }
  } else {
#pragma unroll
    for (int jp = 0; jp < NV / 2; ++jp) {
      const int f = tid + 256 * jp;
      unsigned* dst = w + (f / (EXT / 4)) * (EXT + 4) + (f % (EXT / 4)) * 4;
      float4 v0 = r[2 * jp], v1 = r[2 * jp + 1];          // k = 2 kp and 2 kp + 1 of the same four rows
#pragma unroll
      for (int l = 0; l < NL; ++l) {
        u32x4 o;
        o[0] = pack_limb<NL>(v0.x, v1.x); o[1] = pack_limb<NL>(v0.y, v1.y); o[2] = pack_limb<NL>(v0.z, v1.z); o[3] = pack_limb<NL>(v0.w, v1.w);
        *reinterpret_cast<u32x4*>(dst + l * WORDS) = o;
        if (l + 1 < NL) {
          v0.x = drop_hi16(v0.x); v0.y = drop_hi16(v0.y); v0.z = drop_hi16(v0.z); v0.w = drop_hi16(v0.w);
          v1.x = drop_hi16(v1.x); v1.y = drop_hi16(v1.y); v1.z = drop_hi16(v1.z); v1.w = drop_hi16(v1.w);
        }
      }
    }
  }
}

// MFMA operand (8 bf16: k = 16 s + 8 hb .. +7 of tile row `row`) of one limb image
template <int LAY, int EXT>
RP_DEV bf16x8 limb_frag(const unsigned* w, int row, int s, int hb) {
  constexpr int RSW = 20;
  u32x4 v;
  if (LAY == 0) {
    v = *reinterpret_cast<const u32x4*>(w + row * RSW + 8 * s + 4 * hb);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = w[(8 * s + 4 * hb + q) * (EXT + 4) + row];
  }
  return __builtin_bit_cast(bf16x8, v);
}

// k row / first column of the j-th float4 a thread stages from an MN-contiguous operand tile [32 k][EXT].  fp32 path: plain
// round-robin.  bf16 path: consecutive j are the two k of one k-pair, so the thread can pack (k, k+1) into one word.
template <int EXT, int NL>
RP_DEV int mnk(int tid, int j) {
  return NL == 0 ? (tid + 256 * j) / (EXT / 4) : 2 * ((tid + 256 * (j >> 1)) / (EXT / 4)) + (j & 1);
}
template <int EXT, int NL>
RP_DEV int mnc(int tid, int j) {
  return NL == 0 ? ((tid + 256 * j) % (EXT / 4)) * 4 : ((tid + 256 * (j >> 1)) % (EXT / 4)) * 4;
}

template <int ALAY, int BLAY, int TM, int TN, int NL>
__global__ __launch_bounds__(256, NL > 0 ? 2 : 1) void gemm_kernel(GemmP p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  constexpr int KST = BK + 4;                       // padded row stride of a K-contiguous tile
  // bf16 limb images (NL > 0), per limb, in 32-bit words: a K-contiguous operand is [rows][RSW = 20 words = 32 bf16 + pad]
  // (MFMA operand = one ds_read_b128: row l&31, k = 16 s + 8 (l>>5) .. +7); an MN-contiguous operand is k-pair-major
  // [16 k-pairs][rows + 4] words (word = bf16 of k, k+1 for one row; MFMA operand = 4 conflict-free ds_read_b32)
  constexpr int RSW = 20;
  constexpr int A_WORDS = ALAY == 0 ? BM * RSW : 16 * (BM + 4);
  constexpr int B_WORDS = BLAY == 0 ? BN * RSW : 16 * (BN + 4);
  constexpr int A_FLOATS = NL > 0 ? NL * A_WORDS : (ALAY == 0 ? BM * KST : BK * BM);
  constexpr int B_FLOATS = NL > 0 ? NL * B_WORDS : (BLAY == 0 ? BN * KST : BK * BN);
  constexpr int NA = BM / 32, NB = BN / 32;         // float4 per thread per k-tile
  constexpr int STAGE = A_FLOATS + B_FLOATS;
  constexpr int CST = 32 * TN + 4;                  // row stride of a wave's C tile staged in LDS for the epilogue
  constexpr int C_FLOATS = 4 * 32 * TM * CST;
  // the weight-gradient layout writes small split-K slabs: staging them would only cost LDS (51 KB vs 32 KB -> 3 instead
  // of 5 workgroups per CU, measured 199 -> 278 us), so it keeps the direct dword epilogue
  constexpr bool STAGED = !(ALAY == 1 && BLAY == 1);
  constexpr int LDS_FLOATS = (STAGED && C_FLOATS > STAGE) ? C_FLOATS : STAGE;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FLOATS];
  float* As = lds;
  float* Bs = lds + A_FLOATS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  // XCD-aware tile order (workgroup b runs on XCD b % 8, each XCD has a private L2): XCD x owns the row
  // panels mt = x (mod 8) and walks all N tiles of one panel back to back, so an A panel is pulled from
  // HBM/MALL into ONE L2 once and the weight panel stays L2-resident everywhere.
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  int mt, nt;
  int zid = blockIdx.z;
  if (p.split_k > 1) {
    // split-K launch (1-D grid over tiles x splits): split z runs on XCD z % 8 and walks all its tiles back to back,
    // so the (M-tile, N-tile) workgroups that read the same K chunk of A and B hit the same L2.
    const int T = ntm * ntn;
    const int j = blockIdx.x >> 3;
    zid = (blockIdx.x & 7) + 8 * (j / T);
    if (zid >= p.split_k) return;
    const int t = j % T;
    mt = t / ntn;
    nt = t % ntn;
  } else if (ntm >= 16) {
    const int xs = blockIdx.x >> 3;
    mt = (xs / ntn) * 8 + (blockIdx.x & 7);
    nt = xs % ntn;
    if (mt >= ntm) return;
  } else {                       // few row panels: plain order (the remap would park whole XCDs)
    mt = blockIdx.x / ntn;
    nt = blockIdx.x % ntn;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int zb = p.split_k > 1 ? 0 : zid, zs = p.split_k > 1 ? zid : 0;

  const float* A = p.A + zb * p.sa;
  const float* B = p.B + zb * p.sb;
  const int kbeg = zs * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nkt = (kend - kbeg + BK - 1) / BK;

  float4 ra[NA], rb[NB];
  const bool mn_inside = (m0 + BM <= p.M) && (n0 + BN <= p.N);
  const bool a_bf = NL == 1 && (p.io_bf16 & 1);           // A is bf16 in memory (the bf16 configuration's activations)
  auto lda4 = [&](long long off) { return NL == 1 ? ld4_io(A, off, a_bf) : ld4(A + off); };
  auto gload = [&](int kt) {
    const int k0 = kbeg + kt * BK;
    if (mn_inside && k0 + BK <= kend) {      // wave-uniform fast path: no per-lane guards
#pragma unroll
      for (int j = 0; j < NA; ++j) {
        const int f = tid + 256 * j;
        if (ALAY == 0) ra[j] = lda4((long long)(m0 + (f >> 3)) * p.lda + k0 + (f & 7) * 4);
        else ra[j] = lda4((long long)(k0 + mnk<BM, NL>(tid, j)) * p.lda + m0 + mnc<BM, NL>(tid, j));
      }
#pragma unroll
      for (int j = 0; j < NB; ++j) {
        const int f = tid + 256 * j;
        if (BLAY == 0) rb[j] = ld4(B + (long long)(n0 + (f >> 3)) * p.ldb + k0 + (f & 7) * 4);
        else rb[j] = ld4(B + (long long)(k0 + mnk<BN, NL>(tid, j)) * p.ldb + n0 + mnc<BN, NL>(tid, j));
      }
      return;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int f = tid + 256 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (ALAY == 0) {
        const int gm = m0 + (f >> 3), gk = k0 + (f & 7) * 4;
        if (gm < p.M && gk < kend) v = lda4((long long)gm * p.lda + gk);
      } else {
        const int gk = k0 + mnk<BM, NL>(tid, j), gm = m0 + mnc<BM, NL>(tid, j);
        if (gk < kend && gm < p.M) v = lda4((long long)gk * p.lda + gm);
      }
      ra[j] = v;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int f = tid + 256 * j;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (BLAY == 0) {
        const int gn = n0 + (f >> 3), gk = k0 + (f & 7) * 4;
        if (gn < p.N && gk < kend) v = ld4(B + (long long)gn * p.ldb + gk);
      } else {
        const int gk = k0 + mnk<BN, NL>(tid, j), gn = n0 + mnc<BN, NL>(tid, j);
        if (gk < kend && gn < p.N) v = ld4(B + (long long)gk * p.ldb + gn);
      }
      rb[j] = v;
    }
  };
  auto sstore = [&](int stage = 0) {
    float* as = As + stage * STAGE;
    float* bs = Bs + stage * STAGE;
    if (NL > 0) {
      limb_store<ALAY, BM, NL, NA>(reinterpret_cast<unsigned*>(as), tid, ra);
      limb_store<BLAY, BN, NL, NB>(reinterpret_cast<unsigned*>(bs), tid, rb);
      return;
    }
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int f = tid + 256 * j;
      if (ALAY == 0) st4(as + (f >> 3) * KST + (f & 7) * 4, ra[j]);
      else st4(as + (f / (BM / 4)) * BM + (f % (BM / 4)) * 4, ra[j]);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int f = tid + 256 * j;
      if (BLAY == 0) st4(bs + (f >> 3) * KST + (f & 7) * 4, rb[j]);
      else st4(bs + (f / (BN / 4)) * BN + (f % (BN / 4)) * 4, rb[j]);
    }
  };
  // MFMA operand fragments of one half k-tile (8 k-steps) from LDS stage `stage`
  auto fload = [&](int stage, int half, float (&a)[TM][8], float (&b)[TN][8]) {
    const float* as = As + stage * STAGE;
    const float* bs = Bs + stage * STAGE;
#pragma unroll
    for (int i = 0; i < TM; ++i) {
      if (ALAY == 0) {
        const float* s_ = as + (wm0 + 32 * i + l31) * KST + 16 * hi + 8 * half;
        const float4 x = ld4(s_), y = ld4(s_ + 4);
        a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w;
        a[i][4] = y.x; a[i][5] = y.y; a[i][6] = y.z; a[i][7] = y.w;
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) a[i][t] = as[(16 * hi + 8 * half + t) * BM + wm0 + 32 * i + l31];
      }
    }
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      if (BLAY == 0) {
        const float* s_ = bs + (wn0 + 32 * j + l31) * KST + 16 * hi + 8 * half;
        const float4 x = ld4(s_), y = ld4(s_ + 4);
        b[j][0] = x.x; b[j][1] = x.y; b[j][2] = x.z; b[j][3] = x.w;
        b[j][4] = y.x; b[j][5] = y.y; b[j][6] = y.z; b[j][7] = y.w;
      } else {
#pragma unroll
        for (int t = 0; t < 8; ++t) b[j][t] = bs[(16 * hi + 8 * half + t) * BN + wn0 + 32 * j + l31];
      }
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero16();

  if (nkt > 0) {
    gload(0);
    sstore();
  }
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
    if (NL > 0) {
      constexpr int L = NL > 0 ? NL : 1;
      const unsigned* aw = reinterpret_cast<const unsigned*>(As);
      const unsigned* bw = reinterpret_cast<const unsigned*>(Bs);
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 a[TM][L], b[TN][L];
#pragma unroll
        for (int l = 0; l < L; ++l) {
#pragma unroll
          for (int i = 0; i < TM; ++i) a[i][l] = limb_frag<ALAY, BM>(aw + l * A_WORDS, wm0 + 32 * i + l31, s2, hi);
#pragma unroll
          for (int j = 0; j < TN; ++j) b[j][l] = limb_frag<BLAY, BN>(bw + l * B_WORDS, wn0 + 32 * j + l31, s2, hi);
        }
        // limb products, smallest weight first; la + lb <= L - 1 keeps every term >= 2^-8(L-1) of the leading one
#pragma unroll
        for (int w = L - 1; w >= 0; --w)
#pragma unroll
          for (int la = 0; la <= w; ++la)
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
              for (int j = 0; j < TN; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i][la], b[j][w - la], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float a[TM][8], b[TN][8];
      fload(0, half, a, b);
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i][t], b[j][t], acc[i][j]);
    }
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      sstore();
      __syncthreads();
    }
  }

  tile_epilogue<TM, TN, STAGED, NL == 1>(p, acc, lds, m0, n0, mt, zb, zid);
}

// Fixed-order reduction of the split-K slabs [split][M][N] + epilogue (+ optional transposed store).  A weight gradient is
// small (<= 147 456 elements): with one thread per float4 walking all 64-128 slabs the launch had ~100 workgroups of serial,
// dependent-looking loads and took 18-34 us for 28-57 MB.  Now 4 thread groups per float4 column each sum every fourth slab
// (slab z = g, g + 4, ...; independent loads, unrolled), and their partials are combined in the fixed order g = 0, 1, 2, 3
// through LDS: 4x the workgroups, the same result on every run.
__global__ __launch_bounds__(256) void splitk_reduce_kernel(GemmP p, const float* ws) {
  __shared__ float4 part[4][64];
  const long long total = (long long)p.M * p.N;
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long idx = ((long long)blockIdx.x * 64 + col) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < total) {
    int z = grp;
#pragma unroll 4
    for (; z < p.split_k; z += 4) {
      const float4 t = ld4(ws + z * total + idx);
      s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
    }
  }
  part[grp][col] = s;
  __syncthreads();
  if (grp != 0 || idx >= total) return;
#pragma unroll
  for (int g = 1; g < 4; ++g) {
    const float4 t = part[g][col];
    s.x += t.x; s.y += t.y; s.z += t.z; s.w += t.w;
  }
  const int m = (int)(idx / p.N), n = (int)(idx % p.N);   // N % 4 == 0: the float4 stays in one row
  float o[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const float v = epilogue(o[e], m, n + e, p);
    if (p.trans_c) p.C[(long long)(n + e) * p.ldc + m] = v;
    else p.C[(long long)m * p.ldc + n + e] = v;
  }
}

template <int ALAY, int BLAY, int TM, int TN, int NL>
int launch(const GemmP& p, int nz, hipStream_t st) {
  const int ntn = (p.N + 64 * TN - 1) / (64 * TN), ntm = (p.M + 64 * TM - 1) / (64 * TM);
  dim3 grid(ntm >= 16 ? ntn * ((ntm + 7) / 8) * 8 : ntn * ntm, 1, nz);
  if (p.split_k > 1) grid = dim3(ntn * ntm * ((p.split_k + 7) / 8) * 8, 1, 1);
  hipLaunchKernelGGL((gemm_kernel<ALAY, BLAY, TM, TN, NL>), grid, dim3(256), 0, st, p);
  return 0;
}

template <int ALAY, int BLAY, int NL>
int launch_tiles(const GemmP& p, int nz, int tm, int tn, hipStream_t st) {
  if (tm == 1) {
    if (tn == 1) return launch<ALAY, BLAY, 1, 1, NL>(p, nz, st);
    if (tn == 2) return launch<ALAY, BLAY, 1, 2, NL>(p, nz, st);
    return launch<ALAY, BLAY, 1, 3, NL>(p, nz, st);
  }
  if (tn == 1) return launch<ALAY, BLAY, 2, 1, NL>(p, nz, st);
  if (tn == 2) return launch<ALAY, BLAY, 2, 2, NL>(p, nz, st);
  return launch<ALAY, BLAY, 2, 3, NL>(p, nz, st);
}

template <int NL>
int launch_layouts(const GemmP& p, int nz, int al, int bl, int tm, int tn, hipStream_t st) {
  if (al == 0 && bl == 0) return launch_tiles<0, 0, NL>(p, nz, tm, tn, st);
  if (al == 0 && bl == 1) return launch_tiles<0, 1, NL>(p, nz, tm, tn, st);
  if (al == 1 && bl == 0) return launch_tiles<1, 0, NL>(p, nz, tm, tn, st);
  return launch_tiles<1, 1, NL>(p, nz, tm, tn, st);
}

}  // namespace

extern "C" size_t rp_gemm_workspace_bytes(int M, int N, int split_k) {
  return split_k > 1 ? (size_t)split_k * (size_t)M * (size_t)N * sizeof(float) : 0;
}

extern "C" int rp_gemm(const RpGemm* g, void* stream) {
  if (!g || g->M <= 0 || g->N <= 0 || g->K <= 0) return RP_EBADSHAPE;
  if (g->a_layout < 0 || g->a_layout > 1 || g->b_layout < 0 || g->b_layout > 1) return RP_EBADSHAPE;
  if ((g->lda | g->ldb | g->ldc) & 3) return RP_EALIGN;
  if ((g->a_layout == 0 || g->b_layout == 0) && (g->K & 3)) return RP_EALIGN;
  if (g->a_layout == 1 && (g->M & 3)) return RP_EALIGN;
  if (g->b_layout == 1 && (g->N & 3)) return RP_EALIGN;
  if (((uintptr_t)g->A | (uintptr_t)g->B | (uintptr_t)g->C) & 15) return RP_EALIGN;
  const int batch = g->batch > 0 ? g->batch : 1;
  const int split = g->split_k > 0 ? g->split_k : 1;
  if (batch > 1 && split > 1) return RP_EUNSUPPORTED;
  if (split > 1) {
    if (g->N & 3) return RP_EALIGN;
    if (!g->workspace || g->workspace_bytes < rp_gemm_workspace_bytes(g->M, g->N, split)) return RP_EWORKSPACE;
  }
  GemmP p;
  p.A = g->A; p.B = g->B; p.C = split > 1 ? g->workspace : g->C;
  p.M = g->M; p.N = g->N; p.K = g->K;
  p.lda = g->lda; p.ldb = g->ldb; p.ldc = g->ldc;
  p.sa = g->stride_a; p.sb = g->stride_b; p.sc = g->stride_c;
  p.split_k = split;
  const int ktiles = (g->K + 31) / 32;
  p.k_per_split = ((ktiles + split - 1) / split) * 32;
  p.bias = g->bias; p.pre_out = g->pre_out; p.act = g->act; p.dact = g->dact; p.aux = g->aux; p.residual = g->residual;
  {
    const bool b = g->bias, pr = g->pre_out, rs = g->residual;
    int m = EPI_GENERIC;
    if (g->dact == 0) {
      if (g->act == 0 && !pr) m = b ? (rs ? EPI_BIAS_RES : EPI_BIAS) : (rs ? EPI_RES : EPI_RAW);
      else if (g->act == 1 && b && !rs) m = pr ? EPI_BIAS_GELU_PRE : EPI_BIAS_GELU;
      else if (g->act == 2 && b && !rs && !pr) m = EPI_BIAS_RELU;
    } else if (!b && !pr && !rs && g->act == 0 && g->aux) {
      m = g->dact == 1 ? EPI_DGELU : EPI_DRELU;
    }
    p.epi_mode = m;
  }
  p.trans_c = g->trans_c;
  p.limbs = g->precision;
  p.colsum_part = g->colsum_part;
  p.ln_x = g->ln_x; p.ln_mean = g->ln_mean; p.ln_rstd = g->ln_rstd; p.ln_gamma = g->ln_gamma; p.ln_part = g->ln_part;
  p.io_bf16 = g->io_bf16;
  if (p.io_bf16) {
    // bf16 storage: only with bf16 operand precision, single batch, 4-element-aligned everything; C / pre_out in bf16 need the
    // LDS-staged epilogue of a plain, bias, GELU or GELU' mode (not the LayerNorm-backward or transposed-slab forms)
    if (g->precision != 1 || batch > 1 || (p.io_bf16 & ~7)) return RP_EUNSUPPORTED;
    if ((p.io_bf16 & 2) && (split > 1 || g->ln_x || g->residual || (g->N & 3) || (g->a_layout == 1 && g->b_layout == 1))) return RP_EUNSUPPORTED;
    if ((p.io_bf16 & 4) && !g->aux) return RP_EBADSHAPE;
    // bits 1 (bf16 C / pre_out) and 2 (bf16 aux) are only honoured by the LDS-staged epilogue: a combination that falls to the
    // generic per-element epilogue (act without bias, dact with bias, ...) or to the split-K reduce kernel would write 4-byte
    // floats into a 2-byte-per-element C, or read a bf16 aux as fp32
    if ((p.io_bf16 & 6) && (p.epi_mode == EPI_GENERIC || (g->a_layout == 1 && g->b_layout == 1))) return RP_EUNSUPPORTED;
    if ((p.io_bf16 & 4) && split > 1) return RP_EUNSUPPORTED;
  }
  const bool lnbwd = g->ln_x != nullptr;
  if (lnbwd) {
    if (!g->ln_mean || !g->ln_rstd || !g->ln_gamma || !g->ln_part) return RP_EBADSHAPE;
    if (g->N != 192 || g->ldc != 192 || split > 1 || batch > 1 || g->bias || g->pre_out || g->act || g->dact || g->aux ||
        g->colsum_part || g->trans_c || (g->a_layout == 1 && g->b_layout == 1))
      return RP_EUNSUPPORTED;
    p.epi_mode = EPI_LNBWD;
  }
  if (g->colsum_part && (split > 1 || batch > 1 || (g->N & 3) || (g->a_layout == 1 && g->b_layout == 1))) return RP_EUNSUPPORTED;
  if (p.limbs != 0 && p.limbs != 1 && p.limbs != 3) return RP_EUNSUPPORTED;
  if (g->trans_c && (split == 1 || g->bias || g->pre_out || g->aux || g->residual)) return RP_EUNSUPPORTED;
  // tile shape (TM,TN) = wave tile in 32x32 units; measured on MI355X (tools/gemm_tiles.py): with fp32 MFMA (64
  // cycles per 32x32x2) operand reuse is cheap and occupancy wins -- 128x64 / 64x192 tiles beat 128x192.
  int tm = g->M <= 64 ? 1 : 2;
  int tn = g->N <= 64 ? 1 : 2;
  if (g->M > 64 && p.limbs == 0) {
    const bool reads_mn = g->aux || g->residual;
    if (g->a_layout == 0 && g->b_layout == 0) { tm = 2; tn = 1; }
    else if (g->N % 192 == 0 && !reads_mn) { tm = 1; tn = 3; }
    else { tm = 2; tn = 1; }
  }
  // bf16 limb paths (measured, tools/split_probe.py): 128x64 tiles (3 workgroups per CU) for the forward and input-gradient
  // layouts, 64x192 for the split-K weight gradient; 128x128 would halve the LDS operand reads per MFMA but only fits
  // 2 workgroups per CU and quantises worse (2592 workgroups on 512 slots)
  if (g->M > 64 && p.limbs != 0) {
    const bool reads_mn = g->aux || g->residual;
    if (g->a_layout == 1 && g->b_layout == 1 && g->N % 192 == 0 && !reads_mn) { tm = 1; tn = 3; }
    else { tm = 2; tn = 1; }
  }
  // (64-row tiles for the ragged batched 576-row products -- dQ = dS K, the EMM's 576x96 blocks -- were measured and are
  // NOT faster: those launches are bound by the 170 MB dS read (4.9 TB/s), the half-empty fifth row panel is free)
  if (lnbwd) { tm = 1; tn = 3; }                   // whole 192-wide rows in one workgroup
  else if (static const char* const ov = getenv("RP_GEMM_TILE"); ov) {   // tuning aid only: "TM,TN" (read once per process)
    if (ov[0] >= '1' && ov[0] <= '2' && ov[1] == ',' && ov[2] >= '1' && ov[2] <= '3') { tm = ov[0] - '0'; tn = ov[2] - '0'; }
  }
  if (p.colsum_part && (tn == 3 || p.epi_mode == EPI_GENERIC)) return RP_EUNSUPPORTED;   // needs the staged epilogue, 64 % (8 TN) == 0
  hipStream_t st = (hipStream_t)stream;
  const int nz = batch * split;
  static const bool no_dma = getenv("RP_GEMM_NO_DMA") != nullptr;      // A/B aid: force the register-staged main loop
  if (g->ev_start) (void)hipEventRecord((hipEvent_t)g->ev_start, st);
  if (!no_dma && dma_eligible(p)) launch_dma(p, nz, g->a_layout, g->b_layout, tm, tn, st);
  else if (p.limbs == 3) launch_layouts<3>(p, nz, g->a_layout, g->b_layout, tm, tn, st);
  else if (p.limbs == 1) launch_layouts<1>(p, nz, g->a_layout, g->b_layout, tm, tn, st);
  else launch_layouts<0>(p, nz, g->a_layout, g->b_layout, tm, tn, st);
  if (g->ev_stop) (void)hipEventRecord((hipEvent_t)g->ev_stop, st);
  RP_CHECK_LAUNCH();
  if (split > 1 && g->defer_reduce) {
    if (g->bias || g->pre_out || g->act || g->dact || g->aux || g->residual) return RP_EUNSUPPORTED;
    return RP_OK;                   // the caller finishes C with rp_splitk_reduce_multi
  }
  if (split > 1) {
    GemmP r = p;
    r.C = g->C;
    const long long total = (long long)g->M * g->N;
    const int blocks = (int)((total / 4 + 63) / 64);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3(blocks), dim3(256), 0, st, r, (const float*)g->workspace);
    RP_CHECK_LAUNCH();
  }
  return RP_OK;
}

// ---- deferred split-K reduces of several products in one launch (same arithmetic and order as splitk_reduce_kernel) ----------
namespace {
struct SplitkMultiP {
  const float* ws[RP_SPLITK_MAX];
  float* C[RP_SPLITK_MAX];
  int M[RP_SPLITK_MAX], N[RP_SPLITK_MAX], ldc[RP_SPLITK_MAX], split[RP_SPLITK_MAX], trans[RP_SPLITK_MAX];
  int blk_end[RP_SPLITK_MAX];
  int n;
};
__global__ __launch_bounds__(256) void splitk_reduce_multi_kernel(SplitkMultiP p) {
  __shared__ float4 part[4][64];
  int t = 0;
  while (t + 1 < p.n && (int)blockIdx.x >= p.blk_end[t]) ++t;
  const int local = blockIdx.x - (t ? p.blk_end[t - 1] : 0);
  const float* ws = p.ws[t];
  const int N = p.N[t], split = p.split[t];
  const long long total = (long long)p.M[t] * N;
  const int col = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long idx = ((long long)local * 64 + col) * 4;
  float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
  if (idx < total) {
    int z = grp;
#pragma unroll 4
    for (; z < split; z += 4) {
      const float4 v = ld4(ws + z * total + idx);
      s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
    }
  }
  part[grp][col] = s;
  __syncthreads();
  if (grp != 0 || idx >= total) return;
#pragma unroll
  for (int g = 1; g < 4; ++g) {
    const float4 v = part[g][col];
    s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w;
  }
  const int m = (int)(idx / N), n = (int)(idx % N);
  float* C = p.C[t];
  const int ldc = p.ldc[t];
  if (p.trans[t]) {
    C[(long long)n * ldc + m] = s.x; C[(long long)(n + 1) * ldc + m] = s.y;
    C[(long long)(n + 2) * ldc + m] = s.z; C[(long long)(n + 3) * ldc + m] = s.w;
  } else {
    st4(C + (long long)m * ldc + n, s);
  }
}
}  // namespace

extern "C" int rp_splitk_reduce_multi(const RpSplitkTask* tasks, int n, void* stream) {
  if (!tasks || n <= 0 || n > RP_SPLITK_MAX) return RP_EBADSHAPE;
  SplitkMultiP p{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const RpSplitkTask& t = tasks[i];
    if (!t.ws || !t.C || t.M <= 0 || t.N <= 0 || (t.N & 3) || t.split_k < 2) return RP_EBADSHAPE;
    if ((t.ldc & 3) || (((uintptr_t)t.ws | (uintptr_t)t.C) & 15)) return RP_EALIGN;
    p.ws[i] = t.ws; p.C[i] = t.C; p.M[i] = t.M; p.N[i] = t.N; p.ldc[i] = t.ldc; p.split[i] = t.split_k; p.trans[i] = t.trans_c ? 1 : 0;
    blocks += (int)(((long long)t.M * t.N / 4 + 63) / 64);
    p.blk_end[i] = blocks;
  }
  p.n = n;
  hipLaunchKernelGGL(splitk_reduce_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// ---- several small transposes in one launch (32x32 tiles through LDS) -------------------------------------------------------
namespace {
struct TransposeMultiP {
  const float* src[RP_TRANSPOSE_MAX];
  float* dst[RP_TRANSPOSE_MAX];
  int rows[RP_TRANSPOSE_MAX], cols[RP_TRANSPOSE_MAX], tx[RP_TRANSPOSE_MAX], blk_end[RP_TRANSPOSE_MAX];
  int n;
};
__global__ __launch_bounds__(256) void transpose_multi_kernel(TransposeMultiP p) {
  __shared__ float tile[32][33];
  int t = 0;
  while (t + 1 < p.n && (int)blockIdx.x >= p.blk_end[t]) ++t;
  const int local = blockIdx.x - (t ? p.blk_end[t - 1] : 0);
  const int bx = local % p.tx[t], by = local / p.tx[t];
  const int rows = p.rows[t], cols = p.cols[t];
  const int lx = threadIdx.x & 31, ly = threadIdx.x >> 5;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int r = by * 32 + ly + 8 * k, c = bx * 32 + lx;
    if (r < rows && c < cols) tile[ly + 8 * k][lx] = p.src[t][(long long)r * cols + c];
  }
  __syncthreads();
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int c = bx * 32 + ly + 8 * k, r = by * 32 + lx;
    if (r < rows && c < cols) p.dst[t][(long long)c * rows + r] = tile[lx][ly + 8 * k];
  }
}
}  // namespace

extern "C" int rp_transpose_multi(const RpTransposeTask* tasks, int n, void* stream) {
  if (!tasks || n <= 0 || n > RP_TRANSPOSE_MAX) return RP_EBADSHAPE;
  TransposeMultiP p{};
  int blocks = 0;
  for (int i = 0; i < n; ++i) {
    const RpTransposeTask& t = tasks[i];
    if (!t.src || !t.dst || t.rows <= 0 || t.cols <= 0) return RP_EBADSHAPE;
    p.src[i] = t.src; p.dst[i] = t.dst; p.rows[i] = t.rows; p.cols[i] = t.cols;
    p.tx[i] = (t.cols + 31) / 32;
    blocks += p.tx[i] * ((t.rows + 31) / 32);
    p.blk_end[i] = blocks;
  }
  p.n = n;
  hipLaunchKernelGGL(transpose_multi_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" void* rp_event_create(void) {
  hipEvent_t e = nullptr;
  return hipEventCreate(&e) == hipSuccess ? (void*)e : nullptr;
}
extern "C" void rp_event_destroy(void* ev) {
  if (ev) (void)hipEventDestroy((hipEvent_t)ev);
}
extern "C" float rp_event_elapsed_ms(void* start, void* stop) {
  float ms = -1.f;
  if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return -1.f;
  return hipEventElapsedTime(&ms, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? ms : -1.f;
}
