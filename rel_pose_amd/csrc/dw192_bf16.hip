// dw192_bf16.hip -- weight gradients of the transformer's Linear layers on the bf16 data path (BASELINE.json configs[4]):
//
//     C[n][k] = sum_m A[m][n] * B[m][k],    A [M, N] bf16 (N = 192, 576, 768), B [M, 192] bf16 or fp32,   C fp32
//
// = the autograd dW = dY^T X of nn.Linear (vision_transformer.py:323,330; vit_layers/mlp.py:22,24): one of the two operands of every
// such product in a Block is 192 wide (xn1, o, xn2, or the 192-wide residual-stream gradient).  Both operands are contracted along
// their ROW index (the token), which is the slow index in memory: the case gfx950's transpose read is for.  At bf16 rates the product
// is HBM-bound (154 flop per byte against a machine balance of ~400), so the kernel is a STREAM:
//   * a workgroup owns a [192 x 192] output tile for a contiguous slab of token rows (split-K over the tokens, one slab per workgroup,
//     256 workgroups = one per CU) and keeps it in registers (4 waves x 3 x 3 accumulator tiles = 144 VGPRs);
//   * 64-row stages of A and B go global -> LDS by LDS-DMA (16 bytes per lane) into two buffers; an fp32 B (the residual-stream
//     gradients are fp32) is staged through registers instead and rounded to bf16 once per stage;
//   * every MFMA operand is two ds_read_b64_tr_b16 (lane = output row / column, 8 consecutive tokens per lane), 12 reads per 9 MFMAs;
//     rows are 24 chunks of 16 bytes, chunk ^ (((row >> 1) & 1) << 2) puts the four rows of a transpose-read block into four
//     different 64-byte quarters of the bank row;
//   * partial tiles are written as split-K slabs in rp_gemm's workspace layout ([split][N][192] fp32) and finished by the same
//     fixed-order reduce (rp_splitk_reduce_multi, optionally transposing): deterministic, no atomics.
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

typedef unsigned short bf16_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

RP_DEV void glds16w(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
RP_DEV const void* uniform_vp(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
RP_DEV bf16x8 tr_op(const bf16_t* a0, const bf16_t* a1) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a1);
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}

constexpr int W = 192;            // tile extent both ways
constexpr int SR = 64;            // token rows per stage
constexpr int ST_EL = SR * W;     // bf16 elements of one operand's stage (24 KB)

struct DwP {
  const bf16_t* a; const void* b; float* ws;
  int M, N, lda, rows_per_split, nsplit;
};

RP_DEV int swz_w(int r) { return ((r >> 1) & 1) << 2; }

template <bool BF32>
__global__ __launch_bounds__(256, 1) void dw192_bf16_kernel(DwP p) {
  // bf16 B: both operands by DMA into a THREE-slot ring, the DMA two stages ahead with a counted vmcnt (144 KB of LDS).  fp32 B: its
  // stage goes through registers, and hipcc's own s_waitcnt for those loads drains every younger DMA too, so a deeper ring would buy
  // nothing there: two slots.
  constexpr int NB = BF32 ? 2 : 3;
  __shared__ __attribute__((aligned(16))) bf16_t As[NB][ST_EL];
  __shared__ __attribute__((aligned(16))) bf16_t Bs[NB][ST_EL];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l31 = lane & 31, hi = lane >> 5;
  const int ntile = p.N / W;
  // workgroup b runs on XCD b % 8: the N / 192 tiles of one slab share its B rows, so they take consecutive slots of ONE XCD's L2
  const int nt = (blockIdx.x >> 3) % ntile, sp = ((blockIdx.x >> 3) / ntile) * 8 + (blockIdx.x & 7);
  if (sp >= p.nsplit) return;
  const int m0 = sp * p.rows_per_split;
  const int m1 = min(p.M, m0 + p.rows_per_split);
  const int nst = (m1 - m0 + SR - 1) / SR;                           // (the launcher makes rows_per_split a multiple of SR; M % 64 == 0)
  const bf16_t* ab = p.a + (long long)m0 * p.lda + nt * W;
  const bf16_t* bb16 = reinterpret_cast<const bf16_t*>(p.b) + (long long)m0 * W;
  const float* bb32 = reinterpret_cast<const float*>(p.b) + (long long)m0 * W;

  // DMA plan: a stage image is [64 rows][24 chunks of 16 B] = 24 pieces of 1 KB; wave w moves pieces w, w + 4, ... (6 per operand)
  unsigned aoff[6], boff[6];
#pragma unroll
  for (int i = 0; i < 6; ++i) {
    const int byte = (wave + 4 * i) * 1024 + lane * 16;
    const int r = byte / 384, slot = (byte % 384) >> 4;
    const int ch = slot ^ swz_w(r);
    aoff[i] = (unsigned)(r * p.lda + ch * 8) * 2u;
    boff[i] = (unsigned)(r * W + ch * 8) * 2u;
  }
  const unsigned as0 = (unsigned)(size_t)(rp_lds_ptr_t)(&As[0][0]) + wave * 1024, bs0 = (unsigned)(size_t)(rp_lds_ptr_t)(&Bs[0][0]) + wave * 1024;
  auto issue_a = [&](int s, int buf) {
    const void* src = uniform_vp(ab + (long long)s * SR * p.lda);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16w(src, aoff[i], as0 + buf * (ST_EL * 2) + i * 4096);
  };
  auto issue_b = [&](int s, int buf) {
    const void* src = uniform_vp(bb16 + (long long)s * SR * W);
#pragma unroll
    for (int i = 0; i < 6; ++i) glds16w(src, boff[i], bs0 + buf * (ST_EL * 2) + i * 4096);
  };
  // fp32 B: thread t takes float4 number t + 256 i (i = 0..11) of the [64][192] fp32 stage = row (t + 256 i) / 48, columns 4 ((t + 256 i) % 48)
  float4 breg[BF32 ? 12 : 1];
  auto load_b32 = [&](int s) {
    const float* src = bb32 + (long long)s * SR * W;
#pragma unroll
    for (int i = 0; i < 12; ++i) breg[i] = ld4(src + (long long)(tid + 256 * i) * 4);
  };
  auto store_b32 = [&](int buf) {
#pragma unroll
    for (int i = 0; i < 12; ++i) {
      const int f = tid + 256 * i, r = f / 48, c4 = f % 48;            // 4 elements = half a chunk
      const int slot = (c4 >> 1) ^ swz_w(r);
      *reinterpret_cast<uint2*>(&Bs[buf][0] + r * W + slot * 8 + 4 * (c4 & 1)) =
          make_uint2(pk_bf16(breg[i].x, breg[i].y), pk_bf16(breg[i].z, breg[i].w));
    }
  };

  // transpose-read offsets (elements) of a [64][192] image: operand = column block cb (32 columns), 16-row step st
  //   rows 16 st + 8 hi + 4 half + (t16 >> 2), columns 32 cb + 16 g + 4 (t16 & 3): chunk 4 cb + 2 g + ((t16 & 3) >> 1), swizzle ((t16 >> 3) & 1) << 2
  const int t16 = lane & 15, g = (lane >> 4) & 1;
  const int wr = wave >> 1, wc = wave & 1;                           // this wave's 96 x 96 quadrant of the tile
  const int trow = (8 * hi + (t16 >> 2)) * W + 4 * (t16 & 1);
  int aofs[3], bofs[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    const int sw = ((t16 >> 3) & 1) << 2, x = 2 * g + ((t16 & 3) >> 1);
    aofs[i] = trow + (((12 * wr + 4 * i + x) ^ sw) << 3);
    bofs[i] = trow + (((12 * wc + 4 * i + x) ^ sw) << 3);
  }

  f32x16 acc[3][3];
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = zero16();

  auto compute = [&](int buf) {
    const bf16_t* At = As[buf];
    const bf16_t* Bt = Bs[buf];
#pragma unroll
    for (int st = 0; st < SR / 16; ++st) {
      bf16x8 af[3], bfr[3];
#pragma unroll
      for (int i = 0; i < 3; ++i) {
        af[i] = tr_op(At + st * 16 * W + aofs[i], At + st * 16 * W + 4 * W + aofs[i]);
        bfr[i] = tr_op(Bt + st * 16 * W + bofs[i], Bt + st * 16 * W + 4 * W + bofs[i]);
      }
#pragma unroll
      for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc[i][j] = mfma_bf(af[i], bfr[j], acc[i][j]);
    }
  };
  if constexpr (BF32) {
    if (nst > 0) {
      issue_a(0, 0);
      load_b32(0);
      store_b32(0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    for (int s = 0; s < nst; ++s) {
      const int buf = s & 1;
      if (s + 1 < nst) {
        issue_a(s + 1, buf ^ 1);
        load_b32(s + 1);
      }
      compute(buf);
      if (s + 1 < nst) store_b32(buf ^ 1);                            // (the other buffer: last read in iteration s - 1, behind a barrier)
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
  } else {
    if (nst > 0) { issue_a(0, 0); issue_b(0, 0); }
    if (nst > 1) { issue_a(1, 1); issue_b(1, 1); }
    int buf = 0;
    for (int s = 0; s < nst; ++s) {
      // stage s has landed (12 DMA instructions per wave and stage; stage s + 1 may stay in flight) -- for everybody -- and everybody
      // is past iteration s - 1, whose slot the DMA of stage s + 2 refills
      if (s + 1 < nst) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      if (s + 2 < nst) {
        const int nb = buf == 0 ? 2 : buf - 1;                        // (s + 2) % 3
        issue_a(s + 2, nb);
        issue_b(s + 2, nb);
      }
      compute(buf);
      buf = buf == 2 ? 0 : buf + 1;
    }
  }
  // slab [N][192] of this split: rows n = nt 192 + 96 wr + 32 i + acc_row(r, hi), columns 96 wc + 32 j + l31 (128-byte runs per half-wave)
  float* slab = p.ws + (long long)sp * p.N * W + (long long)(nt * W + 96 * wr) * W + 96 * wc + l31;
#pragma unroll
  for (int i = 0; i < 3; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) slab[(long long)(32 * i + acc_row(r, hi)) * W + 32 * j] = acc[i][j][r];
}

}  // namespace

// number of token-row slabs (= split-K factor, >= 2) for a product with N output rows over M tokens: one workgroup per CU
extern "C" int rp_dw192_bf16_splits(int M, int N) {
  if (M <= 0 || N <= 0 || N % W) return 0;
  const int ntile = N / W, stages = (M + SR - 1) / SR;
  // one workgroup per CU (96 KB of LDS), and workgroup b runs on XCD b % 8 with its 32 CUs: at most 32 / ntile slabs per XCD, or the
  // surplus workgroup of an XCD runs as a second round (measured: 264 workgroups for N = 576 took twice the time of 240)
  int sp = 8 * (32 / ntile);
  if (sp > stages) sp = stages;
  if (sp < 2) sp = 2;
  const int per = (stages + sp - 1) / sp;              // stages per slab; drop the empty tail slabs
  sp = (stages + per - 1) / per;
  return sp < 2 ? 2 : sp;
}

extern "C" size_t rp_dw192_bf16_workspace_bytes(int M, int N) {
  return (size_t)rp_dw192_bf16_splits(M, N) * (size_t)N * W * sizeof(float);
}

extern "C" int rp_dw192_bf16(const void* a, int lda, const void* b, int b_is_f32, int M, int N, void* workspace, size_t workspace_bytes,
                             void* stream) {
  if (!a || !b || !workspace || M <= 0 || N <= 0) return RP_EBADSHAPE;
  if (N % W || M % SR || (lda & 7) || lda < N) return RP_EBADSHAPE;
  if (((uintptr_t)a | (uintptr_t)b | (uintptr_t)workspace) & 15) return RP_EALIGN;
  if (workspace_bytes < rp_dw192_bf16_workspace_bytes(M, N)) return RP_EWORKSPACE;
  DwP p;
  p.a = (const bf16_t*)a; p.b = b; p.ws = (float*)workspace; p.M = M; p.N = N; p.lda = lda;
  p.nsplit = rp_dw192_bf16_splits(M, N);
  const int stages = M / SR;
  p.rows_per_split = ((stages + p.nsplit - 1) / p.nsplit) * SR;
  const dim3 grid((N / W) * ((p.nsplit + 7) / 8) * 8);
  if (b_is_f32) hipLaunchKernelGGL((dw192_bf16_kernel<true>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((dw192_bf16_kernel<false>), grid, dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
