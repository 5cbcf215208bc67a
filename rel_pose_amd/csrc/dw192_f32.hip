// dw192_f32.hip -- weight gradients of the transformer's Linear layers, exact fp32 (the parity path), as an OUTPUT-STATIONARY stream:
//
//     C[n][k] = sum_m A[m][n] * B[m][k],    A [M, N] fp32 (N = 192, 576, 768), B [M, 192] fp32,   C fp32
//
// = the autograd dW = dY^T X of nn.Linear (vision_transformer.py:323,330; vit_layers/mlp.py:22,24); one operand of every such product
// in a Block is 192 wide.  rp_gemm runs these as 64 x 192 tiles with split-K >= 8 (gemm_dma_kernel<1,1,1,3>: 3 workgroups per CU, 0.72
// of the fp32 matrix peak, the dominant symbol of the step).  Here a workgroup OWNS a [192 x 192] output tile for a slab of token rows:
//   * 4 waves x 3 x 3 accumulator tiles of 32 x 32 (144 accumulator registers per lane), ONE wave per SIMD -- v_mfma_f32_32x32x2_f32
//     holds its datasheet rate only at one wave per SIMD (profiles/README.md "measured ceilings": 0.65-0.78 with 2-8 waves), and nine
//     independent accumulators per wave keep the pipe fed without any dependent-issue gap;
//   * 32-row stages of A and B go global -> LDS by LDS-DMA (two buffers, 96 KB; rows are 768 B, copied as they lie: both operands are
//     contracted along their ROW index, so an MFMA operand is one conflict-free ds_read_b32 per lane -- lane = output row / column);
//   * per 2-row k-step a wave issues 6 ds_read_b32 for 9 MFMAs (576 matrix-pipe cycles): LDS and issue bandwidth are idle, one barrier
//     per 144 MFMAs;
//   * split-K slabs in rp_gemm's workspace layout, finished by the same fixed-order reduce (rp_splitk_reduce_multi): deterministic.
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

RP_DEV void glds16f(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
template <int OFF> RP_DEV float lds_rd32(unsigned addr) {
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
RP_DEV const void* uniform_vpf(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}

constexpr int W = 192;            // tile extent both ways
constexpr int SR = 32;            // token rows per stage
constexpr int ST_FL = SR * W;     // floats of one operand's stage (24 KB)

struct DwF {
  const float* a; const float* b; float* ws;
  int M, N, lda, rows_per_split, nsplit;
};

__global__ __launch_bounds__(256, 1) void dw192_f32_kernel(DwF p) {
  __shared__ __attribute__((aligned(16))) float As[2][ST_FL];
  __shared__ __attribute__((aligned(16))) float Bs[2][ST_FL];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int ntile = p.N / W;
  // workgroup b runs on XCD b % 8: the N / 192 tiles of one slab share its B rows through ONE XCD's L2 (dw192_bf16.hip)
  const int nt = (blockIdx.x >> 3) % ntile, sp = ((blockIdx.x >> 3) / ntile) * 8 + (blockIdx.x & 7);
  if (sp >= p.nsplit) return;
  const int m0 = sp * p.rows_per_split;
  const int m1 = min(p.M, m0 + p.rows_per_split);
  const int nst = (m1 - m0) / SR;                                    // (rows_per_split and M are multiples of SR)
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
  auto issue = [&](int s, int buf) {
    const void* sa = uniform_vpf(ab + (long long)s * SR * p.lda);
    const void* sb = uniform_vpf(bb + (long long)s * SR * W);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16f(sa, aoff[i], as0 + buf * (ST_FL * 4) + i * 4096);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16f(sb, boff[i], bs0 + buf * (ST_FL * 4) + i * 4096);
  };

  const int wr = wave >> 1, wc = wave & 1;                           // this wave's 96 x 96 quadrant of the tile
  const int ao = hi * W + 96 * wr + l31, bo = hi * W + 96 * wc + l31;  // k-step t pairs row 2 t + hi with the two half-waves
  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = zero16();

  if (nst > 0) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int s = 0; s < nst; ++s) {
    const int buf = s & 1;
    // The next stage's twelve DMA pieces are issued ONE PER K-STEP, each in the shadow of that step's nine MFMAs (576 matrix-pipe cycles):
    // issued as a block at the stage's top they held the in-order wave -- and the idle pipe -- for ~7 % of the launch
    // (tools/lab/dw_ablate.sh, profiles/r5_dw_ablate.txt).  The last stage re-fetches itself into the dead buffer (no branch in the loop).
    const int sn = min(s + 1, nst - 1);
    const void* sa = uniform_vpf(ab + (long long)sn * SR * p.lda);
    const void* sb = uniform_vpf(bb + (long long)sn * SR * W);
    const unsigned an0 = as0 + (buf ^ 1) * (ST_FL * 4), bn0 = bs0 + (buf ^ 1) * (ST_FL * 4);
    // Operands of k-step t + 1 are read while the nine MFMAs of step t run (one wave per SIMD: nobody else hides the LDS latency), as
    // ds_read_b32 with the step's offset as an IMMEDIATE: hipcc's ds_read2_b32 pairs need a fresh base register per step (8-bit offsets),
    // 32 v_add_u32 per stage, and next to an fp32 MFMA a VALU instruction costs its issue time while an LDS instruction is free
    // (profiles/r5_shadow_lab.txt).  The asm reads are invisible to the compiler's counters: the wait is written out, with the six
    // registers passed through it.
    const unsigned at = (unsigned)(size_t)(rp_lds_ptr_t)(As[buf] + ao), bt = (unsigned)(size_t)(rp_lds_ptr_t)(Bs[buf] + bo);
    float af[3], bf[3];
    static_for<3>([&](auto i) {
      af[i] = lds_rd32<128 * i>(at);
      bf[i] = lds_rd32<128 * i>(bt);
    });
    static_for<SR / 2>([&](auto tt) {
      constexpr int t = tt;
      asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(af[0]), "+v"(af[1]), "+v"(af[2]), "+v"(bf[0]), "+v"(bf[1]), "+v"(bf[2]));
      float an[3], bn[3];
      if constexpr (t + 1 < SR / 2) {
        static_for<3>([&](auto i) {
          an[i] = lds_rd32<(2 * (t + 1) * W + 32 * i) * 4>(at);
          bn[i] = lds_rd32<(2 * (t + 1) * W + 32 * i) * 4>(bt);
        });
      }
      if (t < 6) glds16f(sa, aoff[t < 6 ? t : 0], an0 + t * 4096);
      else if (t < 12) glds16f(sb, boff[t >= 6 && t < 12 ? t - 6 : 0], bn0 + (t - 6) * 4096);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
      if constexpr (t + 1 < SR / 2) {
#pragma unroll
        for (int i = 0; i < 3; ++i) {
          af[i] = an[i];
          bf[i] = bn[i];
        }
      }
    });
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
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

// number of token-row slabs (= split-K factor, >= 2): one workgroup per CU, at most 32 / (N / 192) slabs per XCD (dw192_bf16.hip)
extern "C" int rp_dw192_f32_splits(int M, int N) {
  if (M <= 0 || N <= 0 || N % W) return 0;
  const int ntile = N / W, stages = M / SR;
  int sp = 8 * (32 / ntile);
  if (sp > stages) sp = stages;
  if (sp < 2) sp = 2;
  const int per = (stages + sp - 1) / sp;
  sp = (stages + per - 1) / per;
  return sp < 2 ? 2 : sp;
}

extern "C" size_t rp_dw192_f32_workspace_bytes(int M, int N) {
  return (size_t)rp_dw192_f32_splits(M, N) * (size_t)N * W * sizeof(float);
}

extern "C" int rp_dw192_f32(const float* a, int lda, const float* b, int M, int N, void* workspace, size_t workspace_bytes, void* stream) {
  if (!a || !b || !workspace || M <= 0 || N <= 0) return RP_EBADSHAPE;
  if (N % W || M % SR || M < 2 * SR || (lda & 3) || lda < N) return RP_EBADSHAPE;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)workspace) & 15) return RP_EALIGN;
  if (workspace_bytes < rp_dw192_f32_workspace_bytes(M, N)) return RP_EWORKSPACE;
  DwF p;
  p.a = a; p.b = b; p.ws = (float*)workspace; p.M = M; p.N = N; p.lda = lda;
  p.nsplit = rp_dw192_f32_splits(M, N);
  const int stages = M / SR;
  p.rows_per_split = ((stages + p.nsplit - 1) / p.nsplit) * SR;
  const dim3 grid((N / W) * ((p.nsplit + 7) / 8) * 8);
  hipLaunchKernelGGL(dw192_f32_kernel, grid, dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
