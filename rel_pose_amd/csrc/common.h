// common.h -- device helpers shared by the gfx950 kernels of librelpose_hip.so.
//
// MFMA primitive used everywhere: v_mfma_f32_32x32x2_f32 (exact fp32, 64 FLOP/clk/SIMD, 157 TF chip peak).
//   A operand: lane l holds A[i = l&31][k = l>>5]      (one fp32 VGPR)
//   B operand: lane l holds B[k = l>>5][j = l&31]
//   C/D      : 16 fp32 per lane; reg r of lane l is D[row = (r&3) + 8*(r>>2) + 4*(l>>5)][col = l&31]
// Because the contraction order is free, every kernel here pairs "k-step t" with the two half-waves
// (hi = l>>5) however its data happens to be laid out; see the per-kernel comments.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

#define RP_DEV __device__ __forceinline__

RP_DEV f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

// bf16 operand mode (BASELINE.json configs[4]: "bf16 with MFMA bf16 attention GEMMs"): v_mfma_f32_32x32x16_bf16, fp32
// accumulate, SAME 32x32 accumulator layout as above.  A operand: lane l holds A[i = l&31][k-slots 8*(l>>5) .. +7] as 8 bf16,
// B likewise -- so a run of 8 fp32 MFMA steps whose per-lane operands are a[0..7], b[0..7] (step t pairs a[t] with b[t] in
// each half-wave) is ONE bf16 MFMA on pack8(a), pack8(b): every kernel keeps its fp32 data layout, LDS tiles and k-step
// pairing and only regroups its operands.  Values are rounded to nearest-even (v_cvt_pk_bf16_f32).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
RP_DEV unsigned pk_bf16(float a, float b) {
  bf16x2_t v;
  v[0] = (__bf16)a;
  v[1] = (__bf16)b;
  return __builtin_bit_cast(unsigned, v);
}
RP_DEV bf16x8 pack8(float a0, float a1, float a2, float a3, float a4, float a5, float a6, float a7) {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t w;
  w[0] = pk_bf16(a0, a1); w[1] = pk_bf16(a2, a3); w[2] = pk_bf16(a4, a5); w[3] = pk_bf16(a6, a7);
  return __builtin_bit_cast(bf16x8, w);
}
RP_DEV bf16x8 pack8(const float* v) { return pack8(v[0], v[1], v[2], v[3], v[4], v[5], v[6], v[7]); }
RP_DEV f32x16 mfma_bf(bf16x8 a, bf16x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }

// row of accumulator register r for half-wave hi
RP_DEV constexpr int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

RP_DEV f32x16 zero16() {
  f32x16 z;
#pragma unroll
  for (int i = 0; i < 16; ++i) z[i] = 0.f;
  return z;
}

RP_DEV float gelu_exact(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
RP_DEV float gelu_grad(float x) {
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  const float pdf = 0.39894228040143267794f * expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// Branch-free GELU for kernels whose VALU time matters (the fused MLP): Phi(x) = 1/2 erfc(-x / sqrt 2) with
// 1/2 erfc(t / sqrt 2) = exp2(-t q(t) - 1) for t = |x| <= 5.7 (beyond: 1e-8, below fp32 resolution of 1), q a degree-7 fit
// weighted by its effect on Phi.  |gelu_fast - exact| <= 4.0e-7 over [-9, 9] (at |x| ~ 4.5, i.e. <= 1 ulp of the result;
// the libm erff formulation measures 4.5e-7 there), relative to |x| <= 1.1e-7; 13 VALU ops + one v_exp_f32, no divergence
// (libm's erff is ~31 instructions on each side of a branch).
RP_DEV float gelu_fast(float x) {
  const float t = fminf(fabsf(x), 5.7f);
  float p = 2.796787612e-06f;
  p = fmaf(p, t, -3.893709072e-05f);
  p = fmaf(p, t, 1.841904013e-04f);
  p = fmaf(p, t, 1.414295839e-04f);
  p = fmaf(p, t, -7.068802603e-03f);
  p = fmaf(p, t, 5.249951407e-02f);
  p = fmaf(p, t, 4.592072368e-01f);
  p = fmaf(p, t, 1.151105285e+00f);
  const float e = __builtin_amdgcn_exp2f(fmaf(-t, p, -1.0f));
  return x * (x < 0.f ? e : 1.0f - e);
}

// d/dx gelu(x) = Phi(x) + x phi(x) with the same Phi as gelu_fast and phi(x) = exp2(-x^2 log2(e) / 2) / sqrt(2 pi)
RP_DEV float gelu_grad_fast(float x) {
  const float t = fminf(fabsf(x), 5.7f);
  float p = 2.796787612e-06f;
  p = fmaf(p, t, -3.893709072e-05f);
  p = fmaf(p, t, 1.841904013e-04f);
  p = fmaf(p, t, 1.414295839e-04f);
  p = fmaf(p, t, -7.068802603e-03f);
  p = fmaf(p, t, 5.249951407e-02f);
  p = fmaf(p, t, 4.592072368e-01f);
  p = fmaf(p, t, 1.151105285e+00f);
  const float e = __builtin_amdgcn_exp2f(fmaf(-t, p, -1.0f));
  const float cdf = x < 0.f ? e : 1.0f - e;
  const float pdf = 0.39894228040143267794f * __builtin_amdgcn_exp2f(-0.72134752044448170368f * x * x);
  return fmaf(x, pdf, cdf);
}

// GELU of the bf16 configuration (BASELINE.json configs[4]; the BF instantiations of mlp_fused.hip / linear_rows.hip only): the
// sigmoid ("tanh") form  x / (1 + exp2(-x (K0 + K1 x^2))),  K0 = 2 sqrt(2/pi) log2(e), K1 = 0.044715 K0 -- 5 VALU + v_exp + v_rcp
// against the 14 + v_exp of gelu_fast: torch's GELU(approximate='tanh').  |gelu_bf - exact-erf GELU| <= 4.8e-4 ABSOLUTE over the reals
// (at x = +-2.7; 2e-4 of the value for x > 0.5, but up to 5 % of the value in the negative tail where |GELU| ~ 1e-2): the size of the
// bf16 storage rounding (2e-3 of the value; h is stored as bf16 in this configuration) of an element of magnitude 0.24 -- small against
// the rounding noise of the typical |h| ~ 1 elements it is summed with, NOT below the rounding of every element.  It buys epilogue VALU time, which ADDS to
// the MFMA time on gfx950 and dominates these kernels (150 VALU per 24 MFMAs and chunk), drops by ~40 %.  gelu_bf_grad is the exact
// derivative of gelu_bf (forward and backward stay consistent): s + x s (1 - s)(A + B x^2), s the sigmoid, 9 VALU + 2 transcendentals.
RP_DEV float gelu_bf(float x) {
  const float u = x * fmaf(x * x, 0.10294324f, 2.3022082f);
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u));
}
RP_DEV float gelu_bf_grad(float x) {
  const float x2 = x * x;
  const float u = x * fmaf(x2, 0.10294324f, 2.3022082f);
  const float s = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-u));
  const float dv = fmaf(x2, 0.21406445f, 1.5957692f);
  return fmaf(x * fmaf(-s, s, s), dv, s);
}

// exp / softmax arithmetic runs in the log2 domain: scores are produced pre-multiplied by log2(e) (folded into the
// operand prescale), so every probability is ONE v_exp_f32 instead of libm's ~25-instruction expf (which was ~half of
// the non-MFMA time per attention tile).  v_exp_f32 is accurate to ~1 ulp; measured pose error stays ~1e-6.
#define RP_LOG2E 1.4426950408889634f
#define RP_LN2 0.6931471805599453f
RP_DEV float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }

RP_DEV float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// sum over the 16 lanes of a DPP row (lanes 16q .. 16q+15), result in every lane of the row; fixed order, VALU only
RP_DEV float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
  return v;
}

RP_DEV float row16_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)));
  return v;
}

// 4 x 4 transpose across the four lanes of a DPP quad: before, lane q (= lane & 3) holds a[0..3]; after, it holds b[e] = (a of lane e)[q].
// Two 2 x 2 stages (quad_perm [1,0,3,2], then [2,3,0,1]), 16 VALU operations, no LDS.  Used to turn four 4-byte-per-lane stores of an
// accumulator image into one 16-byte-per-lane store (the stored-dS tiles): lanes 4Q..4Q+3 each end up with four CONSECUTIVE lanes' values
// of one register.
RP_DEV void quad_transpose4(float (&a)[4], int q) {
  const bool o1 = q & 1, o2 = q & 2;
  const float s01 = o1 ? a[0] : a[1], s23 = o1 ? a[2] : a[3];
  const float r01 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s01), 0xB1, 0xF, 0xF, true));
  const float r23 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s23), 0xB1, 0xF, 0xF, true));
  if (o1) { a[0] = r01; a[2] = r23; } else { a[1] = r01; a[3] = r23; }
  const float s02 = o2 ? a[0] : a[2], s13 = o2 ? a[1] : a[3];
  const float r02 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s02), 0x4E, 0xF, 0xF, true));
  const float r13 = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, s13), 0x4E, 0xF, 0xF, true));
  if (o2) { a[0] = r02; a[1] = r13; } else { a[2] = r02; a[3] = r13; }
}

// store a wave's 32 x 32 accumulator tile (16 registers per lane) as its [16 r][64 lanes] image (4 KB at `tile`), values scaled by `mul`:
// four 16-byte stores per lane (1 KiB per wave instruction, contiguous) instead of sixteen 4-byte ones -- the dword form spent ~2700
// cycles per tile issuing stores in attn_bwd_dkdv and backed the load queue up behind them (tools/dkdv_probe.py)
RP_DEV void store_acc_image(float* tile, const f32x16& v, float mul, int lane) {
  const int q = lane & 3;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float a[4] = {v[4 * g] * mul, v[4 * g + 1] * mul, v[4 * g + 2] * mul, v[4 * g + 3] * mul};
    quad_transpose4(a, q);
    *reinterpret_cast<float4*>(tile + (4 * g + q) * 64 + (lane & ~3)) = make_float4(a[0], a[1], a[2], a[3]);
  }
}

// The same image staged through 2 KB of wave-private LDS (T): sixteen lane-contiguous ds_write_b32, four ds_read_b128 of
// consecutive 16-byte chunks, four 1-KiB stores -- and no quad transposes.  An fp32 MFMA runs on the vector ALU's multipliers: every
// VALU instruction of a SIMD, whichever wave issues it, takes its issue time (4.5 - 14 cycles) out of the matrix rate, LDS and scalar
// instructions do not (profiles/r5_shadow_lab.txt).  The transposes were ~130 of the ~350 VALU instructions per 128 MFMAs of the attention
// backward's tile.  A wave's LDS instructions execute in order: the second half may overwrite T right behind the first half's reads.
template <int RP = 8>      // registers per pass: T holds RP * 64 floats (2 KB; 1 KB with RP = 4 where LDS is what bounds the occupancy)
RP_DEV void store_acc_image_lds(float* tile, float* T, const f32x16& v, float mul, int lane) {
#pragma unroll
  for (int h = 0; h < 16 / RP; ++h) {
#pragma unroll
    for (int r = 0; r < RP; ++r) T[r * 64 + lane] = v[RP * h + r] * mul;
#pragma unroll
    for (int g = 0; g < RP / 4; ++g) {
      const float4 x = *reinterpret_cast<const float4*>(T + 4 * (g * 64 + lane));
      *reinterpret_cast<float4*>(tile + 64 * RP * h + 4 * (g * 64 + lane)) = x;
    }
  }
}

// the same image with bf16 elements (2 KB per tile; the bf16 configuration's stored dS): after the quad transpose a lane holds four
// consecutive lanes' values of one register row = 8 bytes; every store instruction still covers four whole 128-byte rows
RP_DEV void store_acc_image_bf16(unsigned short* tile, const f32x16& v, float mul, int lane) {
  const int q = lane & 3;
#pragma unroll
  for (int g = 0; g < 4; ++g) {
    float a[4] = {v[4 * g] * mul, v[4 * g + 1] * mul, v[4 * g + 2] * mul, v[4 * g + 3] * mul};
    quad_transpose4(a, q);
    *reinterpret_cast<uint2*>(tile + (4 * g + q) * 64 + (lane & ~3)) = make_uint2(pk_bf16(a[0], a[1]), pk_bf16(a[2], a[3]));
  }
}

// ---- bf16 rows of the row-resident kernels (linear_rows.hip, mlp_fused.hip; accumulator layout of v_mfma_f32_16x16x*) ----
RP_DEV float4 widen4(uint2 w) {          // 4 bf16 -> fp32 (exact)
  return make_float4(__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u), __uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u));
}
// lane (j, q) holds units 4q..4q+3 of a 32-unit chunk's first 16 (v0) and second 16 (v1); after swapping halves with lane q ^ 1
// (lane ^ 16) it stores 8 consecutive bf16: even q -> [own v0 | partner's v0], odd q -> [partner's v1 | own v1]
template <bool NTS = false>      // NTS: non-temporal store (a LATER kernel reads the row)
RP_DEV void st_bf16x8(unsigned short* dst, float4 v0, float4 v1, int q) {
  const unsigned a0 = pk_bf16(v0.x, v0.y), a1 = pk_bf16(v0.z, v0.w), b0 = pk_bf16(v1.x, v1.y), b1 = pk_bf16(v1.z, v1.w);
  const bool odd = q & 1;
  const unsigned s0 = odd ? a0 : b0, s1 = odd ? a1 : b1;
  const unsigned r0 = __shfl_xor(s0, 16, 64), r1 = __shfl_xor(s1, 16, 64);
  typedef unsigned u4v __attribute__((ext_vector_type(4)));
  const u4v w = odd ? u4v{r0, r1, b0, b1} : u4v{a0, a1, r0, r1};
  if (NTS) __builtin_nontemporal_store(w, reinterpret_cast<u4v*>(dst));
  else *reinterpret_cast<u4v*>(dst) = w;
}
// the inverse for a bf16 aux row: one 16-byte load of 8 consecutive units per lane, halves swapped back into (v0, v1)
RP_DEV void ld_bf16x8(const unsigned short* src, float4& v0, float4& v1, int q) {
  const uint4 w = *reinterpret_cast<const uint4*>(src);
  const bool odd = q & 1;
  const unsigned s0 = odd ? w.x : w.z, s1 = odd ? w.y : w.w;          // even q sends the partner's v0 half, odd q the partner's v1 half
  const unsigned r0 = __shfl_xor(s0, 16, 64), r1 = __shfl_xor(s1, 16, 64);
  v0 = odd ? widen4(make_uint2(r0, r1)) : widen4(make_uint2(w.x, w.y));
  v1 = odd ? widen4(make_uint2(w.z, w.w)) : widen4(make_uint2(r0, r1));
}

// XCD-aware work order for the (image, head) x row-block kernels: workgroup b runs on XCD b % 8 (private 4 MB L2), so
// XCD x takes the (image, head) problems zh = x (mod 8) and runs all NQ row-block workgroups of one problem back to
// back: the K/V (or Q/dO) tiles every one of them streams are then fetched into ONE L2 once.  Measured before: the
// attention forward pulled 736 MB per launch through the fabric for 170 MB of compulsory input.
// Returns false for the padding workgroups of the last group of 8.
RP_DEV bool xcd_problem(int nq, int ZH, int& zh, int& qb) {
  const int j = blockIdx.x >> 3;
  zh = (j / nq) * 8 + (blockIdx.x & 7);
  qb = j % nq;
  return zh < ZH;
}
inline int xcd_grid(int nq, int ZH) { return nq * ((ZH + 7) / 8) * 8; }

// ---- LDS-DMA (global_load_lds_dwordx4): LDS[lds_byte_addr + 16 * lane] <- *(sbase + voff bytes); sbase wave-uniform.
// Inline asm: hipcc drains vmcnt(0) before the next ds_read when it sees the builtin form in flight (the LDS write sits on
// the VM counter); completion is counted by hand -- s_waitcnt vmcnt(N) by every wave, then s_barrier, then the reads.
typedef __attribute__((address_space(3))) void* rp_lds_ptr_t;
RP_DEV void glds16(const float* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
RP_DEV unsigned lds_byte_addr(const float* p) { return (unsigned)(size_t)(rp_lds_ptr_t)(p); }
RP_DEV const float* uniform_ptr(const float* p) {      // SGPR pair for the DMA base where hipcc cannot prove uniformity
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const float*)(((unsigned long long)hi << 32) | lo);
}

RP_DEV float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
RP_DEV void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
RP_DEV void st4_nt(float* p, float4 v) {      // non-temporal: for data a LATER kernel reads (it would only displace what this one re-reads)
  typedef float f4v __attribute__((ext_vector_type(4)));
  __builtin_nontemporal_store(f4v{v.x, v.y, v.z, v.w}, reinterpret_cast<f4v*>(p));
}

#define RP_CHECK_LAUNCH()                         \
  do {                                            \
    hipError_t e__ = hipGetLastError();           \
    if (e__ != hipSuccess) return (int)e__;       \
  } while (0)
