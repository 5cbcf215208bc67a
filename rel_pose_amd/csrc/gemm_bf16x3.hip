// gemm_bf16x3.hip -- EXPERIMENTAL fp32-in / fp32-out GEMM on the bf16 MFMA pipe ("3xBF16 split"), NT layout only.
//
// Not on the default path: the shipped contraction is the exact-fp32 v_mfma_f32_32x32x2_f32 kernel
// in gemm.hip.  This file measures what the 16x faster bf16 pipe buys when fp32 accuracy is emulated:
//   a = a_hi + a_lo  (a_hi = bf16(a), a_lo = bf16(a - a_hi));   a*b ~= a_hi*b_hi + a_hi*b_lo + a_lo*b_hi
// i.e. three v_mfma_f32_32x32x16_bf16 (K = 16, 32 cycles) per 16 k instead of eight fp32 MFMAs (K = 2, 64 cycles): 5.3x
// fewer matrix-pipe cycles; the dropped a_lo*b_lo term is ~2^-18 relative, accumulation stays fp32.
// Operands are split while they are staged global -> VGPR -> LDS (hi and lo images, k-contiguous rows of 32 bf16 padded
// to 40); a lane's MFMA operand is one ds_read_b128 (8 bf16: row l&31, k = 16*step + 8*(l>>5) ..+7).
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct XP {
  const float* A; const float* B; float* C;
  int M, N, K, lda, ldb, ldc;
  const float* bias; const float* residual; float* pre_out; int act;
};

RP_DEV unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }
RP_DEV float bf16_val(float x) { return (float)(__bf16)x; }

// split 4 floats into packed hi / lo bf16 (2 x 32-bit words each)
RP_DEV void split4(const float4 v, uint2& hi, uint2& lo) {
  const float hx = bf16_val(v.x), hy = bf16_val(v.y), hz = bf16_val(v.z), hw = bf16_val(v.w);
  hi.x = (unsigned)bf16_bits(v.x) | ((unsigned)bf16_bits(v.y) << 16);
  hi.y = (unsigned)bf16_bits(v.z) | ((unsigned)bf16_bits(v.w) << 16);
  lo.x = (unsigned)bf16_bits(v.x - hx) | ((unsigned)bf16_bits(v.y - hy) << 16);
  lo.y = (unsigned)bf16_bits(v.z - hz) | ((unsigned)bf16_bits(v.w - hw) << 16);
}

template <int TM, int TN>
__global__ __launch_bounds__(256) void gemm_nt_bf16x3_kernel(XP p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32, RS = 40;        // RS: row stride in bf16 (80 bytes)
  constexpr int NA = BM / 32, NB = BN / 32;
  constexpr int CST = 32 * TN + 4;
  constexpr int STAGE_BYTES = (BM + BN) * RS * 2 * 2;                 // hi + lo images
  constexpr int C_BYTES = 4 * 32 * TM * CST * 4;
  __shared__ __attribute__((aligned(16))) unsigned char lds_raw[STAGE_BYTES > C_BYTES ? STAGE_BYTES : C_BYTES];
  unsigned short* Ah = reinterpret_cast<unsigned short*>(lds_raw);
  unsigned short* Al = Ah + BM * RS;
  unsigned short* Bh = Al + BM * RS;
  unsigned short* Bl = Bh + BN * RS;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, hb = lane >> 5;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  int mt, nt;
  if (ntm >= 16) {
    const int xs = blockIdx.x >> 3;
    mt = (xs / ntn) * 8 + (blockIdx.x & 7);
    nt = xs % ntn;
    if (mt >= ntm) return;
  } else {
    mt = blockIdx.x / ntn;
    nt = blockIdx.x % ntn;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int nkt = (p.K + BK - 1) / BK;

  float4 ra[NA], rb[NB];
  auto gload = [&](int kt) {
    const int k0 = kt * BK;
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int f = tid + 256 * j, gm = min(m0 + (f >> 3), p.M - 1), gk = k0 + (f & 7) * 4;
      ra[j] = gk < p.K ? ld4(p.A + (long long)gm * p.lda + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int f = tid + 256 * j, gn = min(n0 + (f >> 3), p.N - 1), gk = k0 + (f & 7) * 4;
      rb[j] = gk < p.K ? ld4(p.B + (long long)gn * p.ldb + gk) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int j = 0; j < NA; ++j) {
      const int f = tid + 256 * j;
      uint2 h, l;
      split4(ra[j], h, l);
      *reinterpret_cast<uint2*>(Ah + (f >> 3) * RS + (f & 7) * 4) = h;
      *reinterpret_cast<uint2*>(Al + (f >> 3) * RS + (f & 7) * 4) = l;
    }
#pragma unroll
    for (int j = 0; j < NB; ++j) {
      const int f = tid + 256 * j;
      uint2 h, l;
      split4(rb[j], h, l);
      *reinterpret_cast<uint2*>(Bh + (f >> 3) * RS + (f & 7) * 4) = h;
      *reinterpret_cast<uint2*>(Bl + (f >> 3) * RS + (f & 7) * 4) = l;
    }
  };

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero16();

  gload(0);
  sstore();
  __syncthreads();
  for (int kt = 0; kt < nkt; ++kt) {
    if (kt + 1 < nkt) gload(kt + 1);
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const int o = (wm0 + 32 * i + l31) * RS + 16 * s + 8 * hb;
        ah[i] = *reinterpret_cast<const bf16x8*>(Ah + o);
        al[i] = *reinterpret_cast<const bf16x8*>(Al + o);
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const int o = (wn0 + 32 * j + l31) * RS + 16 * s + 8 * hb;
        bh[j] = *reinterpret_cast<const bf16x8*>(Bh + o);
        bl[j] = *reinterpret_cast<const bf16x8*>(Bl + o);
      }
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    }
    __syncthreads();
    if (kt + 1 < nkt) {
      sstore();
      __syncthreads();
    }
  }

  // LDS-staged float4 epilogue (same scheme as gemm.hip)
  float* cs = reinterpret_cast<float*>(lds_raw) + wave * (32 * TM * CST);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[(32 * i + acc_row(r, hb)) * CST + 32 * j + l31] = acc[i][j][r];
  __syncthreads();
  constexpr int C4 = 8 * TN;
#pragma unroll
  for (int it = 0; it < (32 * TM * C4) / 64; ++it) {
    const int idx = lane + 64 * it, row = idx / C4, c4 = idx % C4;
    const int m = m0 + wm0 + row, n = n0 + wn0 + 4 * c4;
    if (m < p.M && n < p.N) {
      float4 v = ld4(cs + row * CST + 4 * c4);
      const long long off = (long long)m * p.ldc + n;
      if (p.bias) {
        const float4 b4 = ld4(p.bias + n);
        v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
      }
      if (p.pre_out) st4(p.pre_out + off, v);
      if (p.act == 1) { v.x = gelu_exact(v.x); v.y = gelu_exact(v.y); v.z = gelu_exact(v.z); v.w = gelu_exact(v.w); }
      if (p.residual) {
        const float4 r4 = ld4(p.residual + off);
        v.x += r4.x; v.y += r4.y; v.z += r4.z; v.w += r4.w;
      }
      st4(p.C + off, v);
    }
  }
}

}  // namespace

extern "C" int rp_gemm_nt_bf16x3(const float* A, const float* B, float* C, int M, int N, int K, int lda, int ldb, int ldc,
                                 const float* bias, const float* residual, float* pre_out, int act, void* stream) {
  if (M <= 0 || N <= 0 || K <= 0 || (N & 3) || (K & 3)) return RP_EBADSHAPE;
  if ((lda | ldb | ldc) & 3) return RP_EALIGN;
  XP p{A, B, C, M, N, K, lda, ldb, ldc, bias, residual, pre_out, act};
  const int ntn = (N + 63) / 64, ntm = (M + 127) / 128;
  dim3 grid(ntm >= 16 ? ntn * ((ntm + 7) / 8) * 8 : ntn * ntm);
  hipLaunchKernelGGL((gemm_nt_bf16x3_kernel<2, 1>), grid, dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
