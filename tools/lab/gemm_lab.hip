// gemm_lab.hip -- standalone (no torch) laboratory for the fp32-MFMA GEMM main loop on gfx950.
// Builds with: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o gemm_lab gemm_lab.hip -ldl
// Runs the shipped rp_gemm (dlopen of librelpose_hip.so) next to experimental kernels on the hot shapes, checks the
// experimental results against it, and prints us / TFLOP/s.  Tuning aid only -- nothing here is on the product path.
#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include "../../include/relpose_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define DEV __device__ __forceinline__
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

DEV f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
typedef float f32x4v __attribute__((ext_vector_type(4)));
DEV f32x4v mfma16(float a, float b, f32x4v c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }
DEV constexpr int acc_row(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }
DEV float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }
DEV void st4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }

// ---------------------------------------------------------------------------------------------------------------
// (0) chip-level fp32 MFMA ceiling under power: every SIMD issues back-to-back v_mfma_f32_32x32x2_f32 on NACC
// accumulators with operands taken from memory (random or zero data)
// ---------------------------------------------------------------------------------------------------------------
template <int NACC>
__global__ __launch_bounds__(256) void peak_kernel(const float* src, float* out, int iters) {
  f32x16 c[NACC];
  for (int n = 0; n < NACC; ++n)
    for (int i = 0; i < 16; ++i) c[n][i] = 0.f;
  float a[8], b[8];
  for (int i = 0; i < 8; ++i) { a[i] = src[(threadIdx.x + 256 * i) & 4095]; b[i] = src[(threadIdx.x * 7 + 256 * i + 13) & 4095]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
      for (int n = 0; n < NACC; ++n) c[n] = mfma32(a[t], b[(t + n) & 7], c[n]);
  }
  float s = 0.f;
  for (int n = 0; n < NACC; ++n)
    for (int i = 0; i < 16; ++i) s += c[n][i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

// ---------------------------------------------------------------------------------------------------------------
// (1) G2: NT GEMM (A [M,K] and W [N,K], both K-contiguous), LDS-DMA staging (global_load_lds_dwordx4), NST LDS stages,
// one barrier per k-tile.  LDS image of an operand tile: [rows][8 chunks of 16 B] (row = 32 floats = 128 B, unpadded, as
// the DMA writes lane-linear) with the chunk index XOR-swizzled by ((row >> 1) & 7) -- applied on the SOURCE address of
// the DMA and again on the ds_read_b128 address -- so the 16 lanes of every ds_read_b128 group hit 16 distinct 16-byte
// slots of the 256-byte bank row.
// ABL (ablation bits): 1 = no C store, 2 = no global loads, 4 = no MFMA
// ---------------------------------------------------------------------------------------------------------------
struct G2P {
  const float* A; const float* W; float* C; const float* bias;
  int M, N, K, lda, ldw, ldc;
  int stagger;   // G2: first-generation workgroups of CU slot s sleep s * stagger * 64 cycles
};

typedef __attribute__((address_space(3))) void* lds_ptr_t;

// LDS-DMA of 16 bytes per lane: LDS destination = lds_byte_addr (wave-uniform, via M0) + 16 * lane; source = sbase + voff (bytes).
// Inline asm so that hipcc's s_waitcnt bookkeeping does not see the transfer (the builtin makes it drain vmcnt(0) before
// the next ds_read); completion is counted by hand with s_waitcnt vmcnt(N) + s_barrier.
DEV void glds16(const float* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
DEV unsigned lds_addr(const float* p) { return (unsigned)(size_t)(lds_ptr_t)(p); }

template <int TM, int TN, int NST, int ABL, int WPS>
__global__ __launch_bounds__(256, WPS) void g2_nt(G2P p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  constexpr int A_FL = BM * BK, B_FL = BN * BK, STAGE = A_FL + B_FL;
  constexpr int CST = 32 * TN + 4;
  constexpr int C_FL = 4 * 32 * TM * CST;
  constexpr int LDS_FL = (C_FL > NST * STAGE) ? C_FL : NST * STAGE;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FL];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  const int ntn = p.N / BN, ntm = p.M / BM;
  const int xs = blockIdx.x >> 3;
  const int mt = (xs / ntn) * 8 + (blockIdx.x & 7), nt = xs % ntn;
  if (mt >= ntm) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nkt = p.K / BK;
  if (p.stagger > 0 && blockIdx.x < 1024) {
    const int slot = blockIdx.x >> 8;
    for (int i = 0; i < slot * p.stagger; ++i) __builtin_amdgcn_s_sleep(1);
  }

  // DMA source addresses: instruction i of this wave covers tile rows (4 i + wave) * 8 .. +7; lane -> (row j = lane / 8,
  // LDS chunk position pos = lane % 8) fetches global chunk pos ^ ((row >> 1) & 7)
  constexpr int NA = BM / 32, NB = BN / 32;
  unsigned va[NA], vb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    va[i] = (unsigned)(row * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    vb[i] = (unsigned)(row * p.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
  }
  const float* Ab = p.A + (long long)m0 * p.lda;
  const float* Wb = p.W + (long long)n0 * p.ldw;
  const unsigned lds0 = lds_addr(lds);
  auto issue = [&](int kt, int st) {
    if (ABL & 2) return;
    const unsigned as = lds0 + (st * STAGE + wave * 8 * BK) * 4, bs = as + A_FL * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) glds16(Ab + kt * BK, va[i], as + i * 32 * BK * 4);
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16(Wb + kt * BK, vb[i], bs + i * 32 * BK * 4);
  };

  // operand fragment addresses: lane (l31, hi) reads row l31 of each 32-row block, logical chunks 4 hi + 2 half + {0, 1}
  const int key = (l31 >> 1) & 7;
  int coff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 2; ++c) coff[h][c] = ((4 * hi + 2 * h + c) ^ key) << 2;
  const int arow = (wm0 + l31) * BK, brow = A_FL + (wn0 + l31) * BK;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // prologue: NST - 1 tiles in flight
#pragma unroll
  for (int s = 0; s < NST - 1; ++s)
    if (s < nkt) issue(s, s);
  if (NST == 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  } else {
    if (nkt > 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(NA + NB) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  __builtin_amdgcn_s_barrier();

  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt % NST;
    if (kt + NST - 1 < nkt) issue(kt + NST - 1, (kt + NST - 1) % NST);
    const float* base = lds + st * STAGE;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float a[TM][8], b[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        const float4 x = ld4(base + arow + 32 * i * BK + coff[half][0]), y = ld4(base + arow + 32 * i * BK + coff[half][1]);
        a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w; a[i][4] = y.x; a[i][5] = y.y; a[i][6] = y.z; a[i][7] = y.w;
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        const float4 x = ld4(base + brow + 32 * j * BK + coff[half][0]), y = ld4(base + brow + 32 * j * BK + coff[half][1]);
        b[j][0] = x.x; b[j][1] = x.y; b[j][2] = x.z; b[j][3] = x.w; b[j][4] = y.x; b[j][5] = y.y; b[j][6] = y.z; b[j][7] = y.w;
      }
      if (!(ABL & 4)) {
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i][t], b[j][t], acc[i][j]);
      } else {
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[i][j][t] += a[i][t] * b[j][t];
      }
    }
    // tile kt+1 must have landed (all but the newest NST-2 tiles' DMAs retired), and everyone must be done reading
    // stage st before the next iteration's DMA overwrites it
    if (NST == 2 || kt + NST - 1 >= nkt) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NST - 2) * (NA + NB)) : "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // epilogue: accumulators -> this wave's LDS region -> 16-byte row segments (+ bias)
  if (ABL & 1) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    if (s == 1.2345e-30f) p.C[tid] = s;
    return;
  }
  float* cs = lds + wave * (32 * TM * CST);
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) cs[(32 * i + acc_row(r, hi)) * CST + 32 * j + l31] = acc[i][j][r];
  __syncthreads();
  constexpr int C4 = 8 * TN;
  float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) b4 = ld4(p.bias + n0 + wn0 + 4 * (lane % C4));      // 64 % C4 == 0: a lane keeps its column group
#pragma unroll
  for (int it = 0; it < (32 * TM * C4) / 64; ++it) {
    const int idx = lane + 64 * it;
    const int row = idx / C4, c4 = idx % C4;
    int m = m0 + wm0 + row;
    const int n = n0 + wn0 + 4 * c4;
    if (ABL & 8) m &= 127;
    float4 v = ld4(cs + row * CST + 4 * c4);
    v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    if (ABL & 16) {
      typedef float f4 __attribute__((ext_vector_type(4)));
      f4 w; w[0] = v.x; w[1] = v.y; w[2] = v.z; w[3] = v.w;
      __builtin_nontemporal_store(w, reinterpret_cast<f4*>(p.C + (long long)m * p.ldc + n));
    } else {
      st4(p.C + (long long)m * p.ldc + n, v);
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------
// (2) G3: persistent NT GEMM.  Same LDS-DMA staging as G2, but
//   * a workgroup walks a sequence of output tiles and its DMA pipeline runs across tile boundaries (the first k-tile
//     of the next tile is in flight while the last k-tile of the current one is multiplied): no per-tile prologue;
//   * the epilogue is deferred: a finished tile's accumulators move to a second register set and are stored straight
//     from registers (lane = column: 32 lanes = one 128-byte row segment per store), one slice per k-iteration of the
//     NEXT tile, so the C stream is spread evenly over time instead of one burst per tile.
// Every k-iteration of every workgroup is then the same instruction mix (DMA in, MFMA, C out): whole-chip lockstep of
// the workgroups no longer alternates "everybody loads / multiplies / stores".
// ---------------------------------------------------------------------------------------------------------------
template <int TM, int TN, int WPS, int NSL>
__global__ __launch_bounds__(256, WPS) void g3_nt(G2P p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  constexpr int A_FL = BM * BK, B_FL = BN * BK, STAGE = A_FL + B_FL;
  __shared__ __attribute__((aligned(16))) float lds[2 * STAGE];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  const int ntn = p.N / BN, ntm = p.M / BM;
  const int V = ntn * ((ntm + 7) / 8) * 8;
  const int nkt = p.K / BK;
  const int G = gridDim.x;

  constexpr int NA = BM / 32, NB = BN / 32;
  unsigned va[NA], vb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    va[i] = (unsigned)(row * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    vb[i] = (unsigned)(row * p.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
  }
  const unsigned lds0 = lds_addr(lds);
  auto issue = [&](const float* Ab, const float* Wb, int kt, int st) {
    const unsigned as = lds0 + (st * STAGE + wave * 8 * BK) * 4, bs = as + A_FL * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) glds16(Ab + kt * BK, va[i], as + i * 32 * BK * 4);
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16(Wb + kt * BK, vb[i], bs + i * 32 * BK * 4);
  };
  const int key = (l31 >> 1) & 7;
  int coff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 2; ++c) coff[h][c] = ((4 * hi + 2 * h + c) ^ key) << 2;
  const int arow = (wm0 + l31) * BK, brow = A_FL + (wn0 + l31) * BK;

  // tile sequence of this workgroup: virtual ids v = blockIdx.x + j G (v & 7 = XCD), decoded like the non-persistent grid
  auto decode = [&](int v, int& m0, int& n0) -> bool {
    const int xs = v >> 3;
    const int mt = (xs / ntn) * 8 + (v & 7), nt = xs % ntn;
    m0 = mt * BM; n0 = nt * BN;
    return v < V && mt < ntm;
  };
  auto next_valid = [&](int& v, int& m0, int& n0) -> bool {
    while (v < V) {
      if (decode(v, m0, n0)) return true;
      v += G;
    }
    return false;
  };

  f32x16 acc[TM][TN], accp[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[i][j][r] = 0.f; accp[i][j][r] = 0.f; }

  // deferred epilogue state of the previous tile
  float* cprev = nullptr;          // uniform: &C[(m0 + wm0) * ldc + n0 + wn0] of the previous tile
  const unsigned loff = (unsigned)(4 * hi * p.ldc + l31);     // this lane's offset from a uniform row pointer
  float bprev[TN], bcur[TN];
  constexpr int REGS = 16 * TM * TN, PER = REGS / NSL;     // accumulator registers per slice
  auto store_slice = [&](int s) {
    // registers q = s * PER .. + PER - 1;  q -> (i, j, r) with r fastest
#define G3_SLICE(SS)                                                                              \
    case SS: {                                                                                    \
      _Pragma("unroll") for (int q = SS * PER; q < SS * PER + PER; ++q) {                         \
        const int r = q & 15, ij = q >> 4, i = ij / TN, j = ij % TN;                              \
        const int row = 32 * i + (r & 3) + 8 * (r >> 2);                                          \
        (cprev + (long long)row * p.ldc + 32 * j)[loff] = accp[i][j][r] + bprev[j];               \
      }                                                                                           \
    } break;
    switch (s) {
      G3_SLICE(0) G3_SLICE(1) G3_SLICE(2) G3_SLICE(3) G3_SLICE(4) G3_SLICE(5) G3_SLICE(6) G3_SLICE(7)
      default: break;
    }
#undef G3_SLICE
  };
  static_assert(NSL == 8, "slice switch is written for 8 slices");

  int v = blockIdx.x, m0, n0;
  bool have = next_valid(v, m0, n0);
  if (!have) return;
  const float* Ab = p.A + (long long)m0 * p.lda;
  const float* Wb = p.W + (long long)n0 * p.ldw;
  issue(Ab, Wb, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  int st = 0;
  const int every = nkt >= NSL ? nkt / NSL : 1;
  bool have_prev = false;

  while (have) {
    int vn = v + G, m0n = 0, n0n = 0;
    const bool have_next = next_valid(vn, m0n, n0n);
#pragma unroll
    for (int j = 0; j < TN; ++j) bcur[j] = p.bias ? p.bias[n0 + wn0 + 32 * j + l31] : 0.f;
    const float* Abn = p.A + (long long)m0n * p.lda;
    const float* Wbn = p.W + (long long)n0n * p.ldw;
    int next_slice = 0;
    for (int kt = 0; kt < nkt; ++kt) {
      if (kt + 1 < nkt) issue(Ab, Wb, kt + 1, st ^ 1);
      else if (have_next) issue(Abn, Wbn, 0, st ^ 1);
      const float* base = lds + st * STAGE;
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        float a[TM][8], b[TN][8];
#pragma unroll
        for (int i = 0; i < TM; ++i) {
          const float4 x = ld4(base + arow + 32 * i * BK + coff[half][0]), y = ld4(base + arow + 32 * i * BK + coff[half][1]);
          a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w; a[i][4] = y.x; a[i][5] = y.y; a[i][6] = y.z; a[i][7] = y.w;
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
          const float4 x = ld4(base + brow + 32 * j * BK + coff[half][0]), y = ld4(base + brow + 32 * j * BK + coff[half][1]);
          b[j][0] = x.x; b[j][1] = x.y; b[j][2] = x.z; b[j][3] = x.w; b[j][4] = y.x; b[j][5] = y.y; b[j][6] = y.z; b[j][7] = y.w;
        }
#pragma unroll
        for (int t = 0; t < 8; ++t)
#pragma unroll
          for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i][t], b[j][t], acc[i][j]);
      }
      int nstored = 0;
      if (have_prev) {
        if (kt % every == 0 && next_slice < NSL) { store_slice(next_slice++); ++nstored; }
        if (kt == nkt - 1)
          while (next_slice < NSL) { store_slice(next_slice++); ++nstored; }
      }
      // loads and stores retire in order on the one VM counter: everything older than this iteration's stores (the DMA of the
      // next k-tile, and the previous iteration's stores) must be done; the stores just issued may stay in flight
      if (nstored == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(PER) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      st ^= 1;
    }
    // tile finished: hand its accumulators to the deferred epilogue
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        accp[i][j] = acc[i][j];
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
      }
    cprev = p.C + (long long)(m0 + wm0) * p.ldc + n0 + wn0;
#pragma unroll
    for (int j = 0; j < TN; ++j) bprev[j] = bcur[j];
    have_prev = true;
    v = vn; m0 = m0n; n0 = n0n; Ab = Abn; Wb = Wbn; have = have_next;
  }
  for (int s2 = 0; s2 < NSL; ++s2) store_slice(s2);
}


// ---------------------------------------------------------------------------------------------------------------
// (3) G4: G2's staging (LDS-DMA, 2 stages, one barrier per k-tile, same swizzled LDS image) with
// v_mfma_f32_16x16x4_f32: lane (i = l & 15, g = l >> 4) supplies row i, k = k_g; group g reads chunks g and g + 4 of its
// row (two ds_read_b128 per 16-row block per k-tile) and uses float t of chunk c in step 4 * (c / 4) + t, so
// step s covers k = {4 g + s % 4 + 16 * (s / 4)}.  Accumulator block: D[row = 4 g + r][col = l & 15].
// ---------------------------------------------------------------------------------------------------------------
template <int TM, int TN, int ABL, int WPS>
__global__ __launch_bounds__(256, WPS) void g4_nt(G2P p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  constexpr int A_FL = BM * BK, B_FL = BN * BK, STAGE = A_FL + B_FL;
  constexpr int CST = 32 * TN + 4;
  constexpr int C_FL = 4 * 32 * TM * CST;
  constexpr int LDS_FL = (C_FL > 2 * STAGE) ? C_FL : 2 * STAGE;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FL];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l15 = lane & 15, g = lane >> 4;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  const int ntn = p.N / BN, ntm = p.M / BM;
  const int xs = blockIdx.x >> 3;
  const int mt = (xs / ntn) * 8 + (blockIdx.x & 7), nt = xs % ntn;
  if (mt >= ntm) return;
  const int m0 = mt * BM, n0 = nt * BN;
  const int nkt = p.K / BK;
  constexpr int NA = BM / 32, NB = BN / 32;
  unsigned va[NA], vb[NB];
#pragma unroll
  for (int i = 0; i < NA; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    va[i] = (unsigned)(row * p.lda + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
  }
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    const int row = (4 * i + wave) * 8 + (lane >> 3);
    vb[i] = (unsigned)(row * p.ldw + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
  }
  const float* Ab = p.A + (long long)m0 * p.lda;
  const float* Wb = p.W + (long long)n0 * p.ldw;
  const unsigned lds0 = lds_addr(lds);
  auto issue = [&](int kt, int st) {
    if (ABL & 2) return;
    const unsigned as = lds0 + (st * STAGE + wave * 8 * BK) * 4, bs = as + A_FL * 4;
#pragma unroll
    for (int i = 0; i < NA; ++i) glds16(Ab + kt * BK, va[i], as + i * 32 * BK * 4);
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16(Wb + kt * BK, vb[i], bs + i * 32 * BK * 4);
  };
  const int key = (l15 >> 1) & 7;
  const int c0 = (g ^ key) << 2, c1 = ((g + 4) ^ key) << 2;
  const int arow = (wm0 + l15) * BK, brow = A_FL + (wn0 + l15) * BK;
  constexpr int RB = 2 * TM, CB = 2 * TN;
  f32x4v acc[RB][CB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
  issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nkt) issue(kt + 1, st ^ 1);
    const float* base = lds + st * STAGE;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int co = half ? c1 : c0;
      float a[RB][4], b[CB][4];
#pragma unroll
      for (int i = 0; i < RB; ++i) {
        const float4 x = ld4(base + arow + 16 * i * BK + co);
        a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w;
      }
#pragma unroll
      for (int j = 0; j < CB; ++j) {
        const float4 x = ld4(base + brow + 16 * j * BK + co);
        b[j][0] = x.x; b[j][1] = x.y; b[j][2] = x.z; b[j][3] = x.w;
      }
      if (!(ABL & 4)) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int j = 0; j < CB; ++j) acc[i][j] = mfma16(a[i][t], b[j][t], acc[i][j]);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (ABL & 1) {
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < RB; ++i)
#pragma unroll
      for (int j = 0; j < CB; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += acc[i][j][r];
    if (s == 1.2345e-30f) p.C[tid] = s;
    return;
  }
  float* cs = lds + wave * (32 * TM * CST);
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int j = 0; j < CB; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) cs[(16 * i + 4 * g + r) * CST + 16 * j + l15] = acc[i][j][r];
  __syncthreads();
  constexpr int C4 = 8 * TN;
#pragma unroll
  for (int it = 0; it < (32 * TM * C4) / 64; ++it) {
    const int idx = lane + 64 * it;
    const int row = idx / C4, c4 = idx % C4;
    const int m = m0 + wm0 + row, n = n0 + wn0 + 4 * c4;
    float4 v = ld4(cs + row * CST + 4 * c4);
    if (p.bias) {
      const float4 b4 = ld4(p.bias + n);
      v.x += b4.x; v.y += b4.y; v.z += b4.z; v.w += b4.w;
    }
    st4(p.C + (long long)m * p.ldc + n, v);
  }
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------
typedef int (*rp_gemm_fn)(const RpGemm*, void*);
static rp_gemm_fn rp_gemm_p;

template <class F>
static double time_us(F f, int n = 20, int warm = 3) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int i = 0; i < warm; ++i) f();
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int i = 0; i < n; ++i) f();
  CK(hipEventRecord(e1, 0));
  CK(hipEventSynchronize(e1));
  float ms = 0.f;
  CK(hipEventElapsedTime(&ms, e0, e1));
  return ms * 1e3 / n;
}

static float* dev_random(size_t n, float scale, unsigned seed) {
  std::vector<float> h(n);
  unsigned s = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    s = s * 1664525u + 1013904223u;
    h[i] = scale * (((s >> 8) & 0xffff) / 32768.0f - 1.0f);
  }
  float* d;
  CK(hipMalloc(&d, n * sizeof(float)));
  CK(hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice));
  return d;
}

static double max_rel_diff(const float* d0, const float* d1, size_t n) {
  std::vector<float> a(n), b(n);
  CK(hipMemcpy(a.data(), d0, n * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(b.data(), d1, n * 4, hipMemcpyDeviceToHost));
  double md = 0, mr = 0;
  for (size_t i = 0; i < n; ++i) { md = fmax(md, fabs((double)a[i] - b[i])); mr = fmax(mr, fabs((double)a[i])); }
  return md / (mr + 1e-30);
}

template <int TM, int TN, int NST, int ABL, int WPS>
static void run_g2(const char* tag, const G2P& p, const float* ref, int only_time = 0) {
  const int ntn = p.N / (64 * TN), ntm = p.M / (64 * TM);
  dim3 grid(ntn * ((ntm + 7) / 8) * 8);
  auto f = [&]() { hipLaunchKernelGGL((g2_nt<TM, TN, NST, ABL, WPS>), grid, dim3(256), 0, 0, p); };
  CK(hipMemset(p.C, 0, (size_t)p.M * p.ldc * 4));
  f();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  double err = ((ABL & ~16) || !ref) ? -1.0 : max_rel_diff(p.C, ref, (size_t)p.M * p.N);
  const double us = time_us(f);
  const double usl = time_us(f, 200, 3);
  printf("  %-34s TM%d TN%d NST%d ABL%d WPS%d : %8.1f us %7.1f TF (200: %8.1f us %7.1f TF) err %.2e\n", tag, TM, TN, NST, ABL, WPS, us,
         2.0 * p.M * p.N * p.K / us * 1e-6, usl, 2.0 * p.M * p.N * p.K / usl * 1e-6, err);
  fflush(stdout);
}


template <int TM, int TN, int WPS>
static void run_g3(const char* tag, const G2P& p, const float* ref, int wgs) {
  const int ntn = p.N / (64 * TN), ntm = p.M / (64 * TM);
  const int V = ntn * ((ntm + 7) / 8) * 8;
  dim3 grid(wgs < V ? wgs : V);
  auto f = [&]() { hipLaunchKernelGGL((g3_nt<TM, TN, WPS, 8>), grid, dim3(256), 0, 0, p); };
  CK(hipMemset(p.C, 0, (size_t)p.M * p.ldc * 4));
  f();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  double err = ref ? max_rel_diff(p.C, ref, (size_t)p.M * p.N) : -1.0;
  const double us = time_us(f);
  const double us_long = time_us(f, 200, 3);
  printf("  %-24s TM%d TN%d WPS%d grid %4d : %8.1f us %7.1f TF (200 launches: %8.1f us %7.1f TF) err %.2e\n", tag, TM, TN, WPS, (int)grid.x,
         us, 2.0 * p.M * p.N * p.K / us * 1e-6, us_long, 2.0 * p.M * p.N * p.K / us_long * 1e-6, err);
  fflush(stdout);
}

template <int TM, int TN, int ABL, int WPS>
static void run_g4(const char* tag, const G2P& p, const float* ref) {
  const int ntn = p.N / (64 * TN), ntm = p.M / (64 * TM);
  dim3 grid(ntn * ((ntm + 7) / 8) * 8);
  auto f = [&]() { hipLaunchKernelGGL((g4_nt<TM, TN, ABL, WPS>), grid, dim3(256), 0, 0, p); };
  CK(hipMemset(p.C, 0, (size_t)p.M * p.ldc * 4));
  f();
  CK(hipDeviceSynchronize());
  CK(hipGetLastError());
  double err = (ABL || !ref) ? -1.0 : max_rel_diff(p.C, ref, (size_t)p.M * p.N);
  const double us = time_us(f);
  const double usl = time_us(f, 200, 3);
  printf("  %-34s TM%d TN%d ABL%d WPS%d      : %8.1f us %7.1f TF (200: %8.1f us %7.1f TF) err %.2e\n", tag, TM, TN, ABL, WPS, us,
         2.0 * p.M * p.N * p.K / us * 1e-6, usl, 2.0 * p.M * p.N * p.K / usl * 1e-6, err);
  fflush(stdout);
}

int main(int argc, char** argv) {
  const char* libpath = argc > 1 ? argv[1] : "rel_pose_amd/librelpose_hip.so";
  void* h = dlopen(libpath, RTLD_NOW);
  if (!h) { printf("dlopen failed: %s\n", dlerror()); return 1; }
  rp_gemm_p = (rp_gemm_fn)dlsym(h, "rp_gemm");

  const int Mfull = 128 * 576;
  struct Shape { const char* name; int N, K; } shapes[] = {{"qkv", 576, 192}, {"proj", 192, 192}, {"fc1", 768, 192}, {"fc2", 192, 768}};
  for (auto& s : shapes) {
    const int M = Mfull, N = s.N, K = s.K;
    float* A = dev_random((size_t)M * K, 1.0f, 3);
    float* W = dev_random((size_t)N * K, 0.07f, 5);
    float* bias = dev_random(N, 0.1f, 7);
    float *C0, *C1;
    CK(hipMalloc(&C0, (size_t)M * N * 4)); CK(hipMalloc(&C1, (size_t)M * N * 4));
    printf("%s: M=%d N=%d K=%d  (%.2f GFLOP, ideal %.1f us at 157.3 TF)\n", s.name, M, N, K, 2.0 * M * N * K * 1e-9, 2.0 * M * N * K / 157.3e6);
    RpGemm g; memset(&g, 0, sizeof(g));
    g.A = A; g.B = W; g.C = C0; g.M = M; g.N = N; g.K = K; g.lda = K; g.ldb = K; g.ldc = N; g.batch = 1; g.split_k = 1; g.bias = bias;
    {
      int rc = rp_gemm_p(&g, nullptr);
      CK(hipDeviceSynchronize());
      double us = time_us([&]() { rp_gemm_p(&g, nullptr); });
      double usl = time_us([&]() { rp_gemm_p(&g, nullptr); }, 200, 3);
      printf("  %-34s                          : %8.1f us %7.1f TF (200: %8.1f us %7.1f TF) rc %d\n", "rp_gemm (shipped)", us, 2.0 * M * N * K / us * 1e-6, usl, 2.0 * M * N * K / usl * 1e-6, rc);
    }
    G2P p{A, W, C1, bias, M, N, K, K, K, N, 0};
    run_g2<2, 1, 2, 0, 1>("g2 bias hoisted, stores back to back", p, C0);
    run_g2<1, 1, 2, 0, 1>("g2 bias hoisted, stores back to back", p, C0);
    run_g2<2, 1, 2, 8, 1>("  ablate: C rows folded (L2)", p, nullptr);
    run_g2<2, 1, 2, 1, 1>("  ablate: no C store", p, nullptr);
    CK(hipFree(A)); CK(hipFree(W)); CK(hipFree(bias)); CK(hipFree(C0)); CK(hipFree(C1));
  }
  return 0;
}
