// dw192_split3.hip -- the weight gradients of the transformer's Linear layers with fp32 operands and fp32-GRADE results on the BF16 matrix
// pipe (the gate experiment for the only route past the exact-fp32 MFMA ceiling):
//
//     C[n][k] = sum_m A[m][n] * B[m][k],    A [M, N] fp32 (N = 192, 576, 768), B [M, 192] fp32,   C fp32
//
// = the autograd dW = dY^T X of nn.Linear (vision_transformer.py:323,330; vit_layers/mlp.py:22,24) -- the product rp_dw192_f32 computes on
// v_mfma_f32_32x32x2_f32 (157 TF).  Here every fp32 operand is split ON CHIP into three bf16 limbs by an error-free transformation
//     x = x0 + x1 + x2,   x0 = rne_bf16(x), x1 = rne_bf16(x - x0), x2 = x - x0 - x1   (exact: 3 x 8 significant bits = fp32's 24)
// and the product is accumulated in fp32 from SIX of the nine limb products on v_mfma_f32_32x32x16_bf16 (2.5 PF):
//     a b ~ a0 b0 + a0 b1 + a1 b0 + a1 b1 + a0 b2 + a2 b0          dropped: a1 b2 + a2 b1 + a2 b2 <= 2^-26 |a b|
// (|x1| <= 2^-9 |x|, |x2| <= 2^-18 |x| with round-to-nearest limbs): a quarter of the 2^-24 an fp32 product-accumulate rounds by, and a
// 32x32x16 MFMA rounds its accumulator once per 16 products where 32x32x2 rounds it once per 2.  6 bf16 MFMAs of 32 cycles replace 8 fp32
// MFMAs of 64 cycles per 16 token rows: 2.67x fewer matrix-pipe cycles.
//   * the split costs 5.5 VALU per element (v_cvt_pk_bf16_f32 rounds a pair, shift / mask + v_sub_f32 form x - x0) = 264 VALU for the 54
//     MFMAs of a 16-row step: 4.9 per MFMA, about what the bf16 matrix unit co-issues (profiles/r5_shadow_lab.txt: 4-5 per 32-cycle
//     MFMA) -- unlike next to an fp32 MFMA, where a VALU instruction costs its full issue time;
//   * structure = dw192_f32.hip: a workgroup owns a [192 x 192] output tile for a slab of token rows, 4 waves x 3 x 3 accumulator tiles,
//     one wave per SIMD; 32-row fp32 stages of A and B go global -> LDS by LDS-DMA into a THREE-slot ring (144 KB), the DMA two stages
//     ahead with a counted vmcnt (at 2.67x the matrix rate the stream needs ~8 TB/s for N = 192: the kernel is HBM-bound there);
//   * software pipeline over 16-row steps: the raw ds_read_b32 + the split of step u + 1 run under the MFMAs of step u; the one barrier
//     per stage sits where the step after it needs the next stage;
//   * same split-K slabs / fixed-order reduce as rp_dw192_f32 (rp_dw192_f32_splits / _workspace_bytes apply): deterministic.
// Not handled (documented, not needed for gradients): an Inf operand gives NaN (Inf - Inf in the split); limbs below the bf16 normal
// range are flushed, i.e. elements below ~2^-108 lose their low limbs (absolute error < 2^-126 per product).
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

RP_DEV void glds16s(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
RP_DEV const void* uniform_vps(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

constexpr int W = 192;            // tile extent both ways
constexpr int SR = 32;            // token rows per stage (two 16-row MFMA steps)
constexpr int ST_FL = SR * W;     // floats of one operand's stage (24 KB)
constexpr int NSLOT = 3;
#ifndef SPLIT3_MODE
#define SPLIT3_MODE 0
#endif

struct DwS {
  const float* a; const float* b; float* ws;
  int M, N, lda, rows_per_split, nsplit;
};

// three bf16 limbs of eight fp32 values (the eight token rows a lane feeds into one 32x32x16 operand), packed as MFMA operands
struct Limbs { bf16x8 l0, l1, l2; };

// MODE 0 (shipped): round-to-nearest limbs, the residual by shift / mask + v_sub_f32 (5.5 VALU per element).  The other modes exist for
// tools/lab/dw_split3_lab.sh (-DSPLIT3_MODE=): 1 = residual by v_dot2c_f32_bf16 with the constants (-1, 0) / (0, -1) in SGPRs (3.5 VALU per
// element, but SLOWER: the dot unit is not free beside MFMAs; and NEVER as inline constants: hipcc encodes the pair (-1, 0) as the inline
// operand -1.0, which the instruction does not read as that pair -- tools/lab/split3_lab.hip: every residual wrong); 2 = truncated limbs
// (v_and_b32 / v_perm_b32: same count, 6-8 % faster than the v_cvt_pk form, dropped terms up to 2^-23 instead of 2^-26); 3 = no split
// (timing floor, wrong numbers).
template <int MODE>
RP_DEV Limbs split3(const float (&x)[8]) {
  unsigned c10 = 0x0000BF80u, c01 = 0xBF800000u;                 // (-1, 0) picks the low half of a packed pair, (0, -1) the high half
  if (MODE == 1) { asm volatile("" : "+s"(c10)); asm volatile("" : "+s"(c01)); }
  const bf16x2 m10 = __builtin_bit_cast(bf16x2, c10), m01 = __builtin_bit_cast(bf16x2, c01);
  u32x4 w0, w1, w2;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float a = x[2 * q], b = x[2 * q + 1];
    if (MODE == 0) {
      const unsigned p0 = pk_bf16(a, b);                                          // v_cvt_pk_bf16_f32 (round to nearest even)
      const float ra = a - __builtin_bit_cast(float, p0 << 16);                   // a - a0, exact
      const float rb = b - __builtin_bit_cast(float, p0 & 0xffff0000u);
      const unsigned p1 = pk_bf16(ra, rb);
      const float sa = ra - __builtin_bit_cast(float, p1 << 16);                  // a - a0 - a1, exact, fits 8 bits
      const float sb = rb - __builtin_bit_cast(float, p1 & 0xffff0000u);
      w0[q] = p0; w1[q] = p1; w2[q] = pk_bf16(sa, sb);
    } else if (MODE == 1) {
      const bf16x2 p0 = {(__bf16)a, (__bf16)b};
      const float ra = __builtin_amdgcn_fdot2_f32_bf16(p0, m10, a, false);
      const float rb = __builtin_amdgcn_fdot2_f32_bf16(p0, m01, b, false);
      const bf16x2 p1 = {(__bf16)ra, (__bf16)rb};
      const float sa = __builtin_amdgcn_fdot2_f32_bf16(p1, m10, ra, false);
      const float sb = __builtin_amdgcn_fdot2_f32_bf16(p1, m01, rb, false);
      const bf16x2 p2 = {(__bf16)sa, (__bf16)sb};
      w0[q] = __builtin_bit_cast(unsigned, p0); w1[q] = __builtin_bit_cast(unsigned, p1); w2[q] = __builtin_bit_cast(unsigned, p2);
    } else if (MODE == 2) {
      const unsigned ua = __builtin_bit_cast(unsigned, a), ub = __builtin_bit_cast(unsigned, b);
      const float ra = a - __builtin_bit_cast(float, ua & 0xffff0000u), rb = b - __builtin_bit_cast(float, ub & 0xffff0000u);
      const unsigned va = __builtin_bit_cast(unsigned, ra), vb = __builtin_bit_cast(unsigned, rb);
      const float sa = ra - __builtin_bit_cast(float, va & 0xffff0000u), sb = rb - __builtin_bit_cast(float, vb & 0xffff0000u);
      w0[q] = __builtin_amdgcn_perm(ub, ua, 0x07060302u);
      w1[q] = __builtin_amdgcn_perm(vb, va, 0x07060302u);
      w2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, sb), __builtin_bit_cast(unsigned, sa), 0x07060302u);
    } else {
      w0[q] = w1[q] = w2[q] = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, b), __builtin_bit_cast(unsigned, a), 0x07060302u);
    }
  }
  Limbs r;
  r.l0 = __builtin_bit_cast(bf16x8, w0);
  r.l1 = __builtin_bit_cast(bf16x8, w1);
  r.l2 = __builtin_bit_cast(bf16x8, w2);
  return r;
}

template <int MODE>
__global__ __launch_bounds__(256, 1) void dw192_split3_kernel(DwS p) {
  __shared__ __attribute__((aligned(16))) float As[NSLOT][ST_FL];
  __shared__ __attribute__((aligned(16))) float Bs[NSLOT][ST_FL];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int ntile = p.N / W;
  // workgroup b runs on XCD b % 8: the N / 192 tiles of one slab share its B rows through ONE XCD's L2 (dw192_bf16.hip)
  const int nt = (blockIdx.x >> 3) % ntile, sp = ((blockIdx.x >> 3) / ntile) * 8 + (blockIdx.x & 7);
  if (sp >= p.nsplit) return;
  const int m0 = sp * p.rows_per_split;
  const int m1 = min(p.M, m0 + p.rows_per_split);
  const int nst = (m1 - m0) / SR;                                    // (rows_per_split and M are multiples of SR)
  if (nst <= 0) return;
  const float* ab = p.a + (long long)m0 * p.lda + nt * W;
  const float* bb = p.b + (long long)m0 * W;

  // DMA plan: a stage image is [32 rows][768 B] = 24 pieces of 1 KB; wave w moves pieces w, w + 4, ... (6 per operand), copied as they lie
  unsigned aoff[6], boff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int byte = (wave + 4 * i) * 1024 + lane * 16;
    const int r = byte / 768, c = byte % 768;
    aoff[i] = (unsigned)(r * p.lda * 4 + c);
    boff[i] = (unsigned)(r * W * 4 + c);
  }
  const unsigned as0 = (unsigned)(size_t)(rp_lds_ptr_t)(&As[0][0]) + wave * 1024, bs0 = (unsigned)(size_t)(rp_lds_ptr_t)(&Bs[0][0]) + wave * 1024;
  // stage s -> slot; stages past the end re-fetch the last one into a dead slot (no branch, constant vmcnt)
  auto issue = [&](int s, int slot) {
    const int sc = min(s, nst - 1);
    const void* sa = uniform_vps(ab + (long long)sc * SR * p.lda);
    const void* sb = uniform_vps(bb + (long long)sc * SR * W);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16s(sa, aoff[i], as0 + slot * (ST_FL * 4) + i * 4096);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16s(sb, boff[i], bs0 + slot * (ST_FL * 4) + i * 4096);
  };

  const int wr = wave >> 1, wc = wave & 1;                           // this wave's 96 x 96 quadrant of the tile
  // operand element e (0..7) of 16-row step kk of a stage: row 16 kk + 8 hi + e, column 96 w + 32 i + l31
  const int ao = 8 * hi * W + 96 * wr + l31, bo = 8 * hi * W + 96 * wc + l31;
  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = zero16();

  // Software pipeline over 16-row steps u = 2 s + kk, three deep: during step u the wave issues the 54 MFMAs of step u (limbs L[u & 1]),
  // splits the raw values of step u + 1 (read one step ago) into L[(u + 1) & 1] and reads the raw values of step u + 2 from LDS.  A step
  // is cut into SIX SLICES, one per operand block g (0-2: A blocks, 3-5: B blocks) and per limb product: slice g = the nine MFMAs of
  // product g + the split of block g (44 VALU) + the eight LDS reads of block g + one DMA piece, and inside a slice the instruction order
  // is PINNED (sched_group_barrier: MFMA, LDS read, 5 VALU, MFMA, ...): one wave per SIMD issues in order, so a VALU instruction is only
  // hidden if it stands between two MFMAs in the program text, and hipcc on its own clumps the reads and their waits (N = 768 at 128
  // images, isolated: 139 -> 118 us; ablations in profiles/r6_split3_gate.txt: no split 100, no DMA pieces 100, neither nor reads 84 = the
  // bare 3888 MFMAs + epilogue at the ~1.9 GHz this matrix load sustains).
  Limbs L[2][6];
  float raw[2][6][8];
  auto read_block = [&](const float* at, const float* bt, int kk, auto G, float (&x)[8]) {
    constexpr int g = G;
    const float* t = (g < 3 ? at : bt) + 16 * kk * W + 32 * (g % 3);
#pragma unroll
    for (int e = 0; e < 8; ++e) x[e] = t[e * W];
  };
  auto limb = [](const Limbs& l, int k) -> const bf16x8& { return k == 0 ? l.l0 : k == 1 ? l.l1 : l.l2; };
  // slice g of a step: Lc = this step's limbs, rc -> Ln = the next step's block g, (at, bt, kk) -> rn = block g of the step after next;
  // q = DMA piece of the stage being fetched (0-5: A, 6-11: B)
  auto slice = [&](auto G, const Limbs (&Lc)[6], Limbs (&Ln)[6], const float (&rc)[6][8], float (&rn)[6][8], const float* at, const float* bt,
                   int kk, const void* sa, const void* sb, unsigned adst, unsigned bdst) {
    constexpr int g = G;
    constexpr int pa = g == 0 ? 2 : (g == 2 || g == 3) ? 1 : 0;      // the small partial products first: (a2 b0) (a0 b2) (a1 b1) (a1 b0) (a0 b1) (a0 b0)
    constexpr int pb = g == 1 ? 2 : (g == 2 || g == 4) ? 1 : 0;
    read_block(at, bt, kk, G, rn[g]);
    Ln[g] = split3<MODE>(rc[g]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int j = 0; j < 3; ++j) acc[i][j] = mfma_bf(limb(Lc[i], pa), limb(Lc[3 + j], pb), acc[i][j]);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
      __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
      if (k < 8) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
      __builtin_amdgcn_sched_group_barrier(0x002, 5, 0);
    }
    __builtin_amdgcn_sched_barrier(0);
    const int q = 6 * kk + g;
    if (q < 6) glds16s(sa, aoff[q < 6 ? q : 0], adst + q * 4096);
    else glds16s(sb, boff[q >= 6 ? q - 6 : 0], bdst + (q - 6) * 4096);
    __builtin_amdgcn_sched_barrier(0);
  };
  using I0 = std::integral_constant<int, 0>; using I1 = std::integral_constant<int, 1>; using I2 = std::integral_constant<int, 2>;
  using I3 = std::integral_constant<int, 3>; using I4 = std::integral_constant<int, 4>; using I5 = std::integral_constant<int, 5>;

  issue(0, 0);
  issue(1, 1);
  issue(2, 2);
  asm volatile("s_waitcnt vmcnt(24)" ::: "memory");                  // stage 0 has landed (mine) ...
  __builtin_amdgcn_s_barrier();                                      // ... and everybody's
  {
    const float* at = As[0] + ao;
    const float* bt = Bs[0] + bo;
    read_block(at, bt, 0, I0{}, raw[0][0]); read_block(at, bt, 0, I1{}, raw[0][1]); read_block(at, bt, 0, I2{}, raw[0][2]);
    read_block(at, bt, 0, I3{}, raw[0][3]); read_block(at, bt, 0, I4{}, raw[0][4]); read_block(at, bt, 0, I5{}, raw[0][5]);
#pragma unroll
    for (int g = 0; g < 6; ++g) L[0][g] = split3<MODE>(raw[0][g]);
    read_block(at, bt, 1, I0{}, raw[1][0]); read_block(at, bt, 1, I1{}, raw[1][1]); read_block(at, bt, 1, I2{}, raw[1][2]);
    read_block(at, bt, 1, I3{}, raw[1][3]); read_block(at, bt, 1, I4{}, raw[1][4]); read_block(at, bt, 1, I5{}, raw[1][5]);
  }
  int slot = 0;
  for (int s = 0; s < nst; ++s) {
    const int nslot = slot == 2 ? 0 : slot + 1;
    // stage s is in registers (raw / limbs) -- for everybody behind the barrier -- so its slot takes stage s + 3, one piece per slice;
    // stage s + 1, which the twelve slices read, has landed when at most the 12 pieces of stage s + 2 are in flight
    asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    const int sc = min(s + 3, nst - 1);                               // (past the end: a re-fetch into a dead slot, constant vmcnt)
    const void* sa = uniform_vps(ab + (long long)sc * SR * p.lda);
    const void* sb = uniform_vps(bb + (long long)sc * SR * W);
    const unsigned adst = as0 + slot * (ST_FL * 4), bdst = bs0 + slot * (ST_FL * 4);
    const float* at = As[nslot] + ao;
    const float* bt = Bs[nslot] + bo;
    // step 2 s: MFMAs on L[0]; raw[1] = (s, 1) -> L[1]; raw[0] <- (s + 1, 0)
    slice(I0{}, L[0], L[1], raw[1], raw[0], at, bt, 0, sa, sb, adst, bdst);
    slice(I1{}, L[0], L[1], raw[1], raw[0], at, bt, 0, sa, sb, adst, bdst);
    slice(I2{}, L[0], L[1], raw[1], raw[0], at, bt, 0, sa, sb, adst, bdst);
    slice(I3{}, L[0], L[1], raw[1], raw[0], at, bt, 0, sa, sb, adst, bdst);
    slice(I4{}, L[0], L[1], raw[1], raw[0], at, bt, 0, sa, sb, adst, bdst);
    slice(I5{}, L[0], L[1], raw[1], raw[0], at, bt, 0, sa, sb, adst, bdst);
    // step 2 s + 1: MFMAs on L[1]; raw[0] = (s + 1, 0) -> L[0]; raw[1] <- (s + 1, 1)
    slice(I0{}, L[1], L[0], raw[0], raw[1], at, bt, 1, sa, sb, adst, bdst);
    slice(I1{}, L[1], L[0], raw[0], raw[1], at, bt, 1, sa, sb, adst, bdst);
    slice(I2{}, L[1], L[0], raw[0], raw[1], at, bt, 1, sa, sb, adst, bdst);
    slice(I3{}, L[1], L[0], raw[0], raw[1], at, bt, 1, sa, sb, adst, bdst);
    slice(I4{}, L[1], L[0], raw[0], raw[1], at, bt, 1, sa, sb, adst, bdst);
    slice(I5{}, L[1], L[0], raw[0], raw[1], at, bt, 1, sa, sb, adst, bdst);
    slot = nslot;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                   // no DMA may land in the LDS of a workgroup that has left
  // slab [N][192] of this split: rows n = nt 192 + 96 wr + 32 i + acc_row(r, hi), columns 96 wc + 32 j + l31
  float* slab = p.ws + (long long)sp * p.N * W + (long long)(nt * W + 96 * wr) * W + 96 * wc + l31;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(long long)(32 * i + acc_row(r, hi)) * W + 32 * j] = acc[i][j][r];
}

}  // namespace

extern "C" int rp_dw192_split3(const float* a, int lda, const float* b, int M, int N, void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !b || !workspace || M <= 0 || N <= 0) return RP_EBADSHAPE;
  if (N % W || M % SR || M < 2 * SR || (lda & 3) || lda < N) return RP_EBADSHAPE;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)workspace) & 15) return RP_EALIGN;
  if (workspace_bytes < rp_dw192_f32_workspace_bytes(M, N)) return RP_EWORKSPACE;
  DwS p;
  p.a = a; p.b = b; p.ws = (float*)workspace; p.M = M; p.N = N; p.lda = lda;
  p.nsplit = rp_dw192_f32_splits(M, N);
  const int stages = M / SR;
  p.rows_per_split = ((stages + p.nsplit - 1) / p.nsplit) * SR;
  const dim3 grid((N / W) * ((p.nsplit + 7) / 8) * 8);
  hipLaunchKernelGGL(dw192_split3_kernel<SPLIT3_MODE>, grid, dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
