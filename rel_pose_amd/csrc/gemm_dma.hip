// gemm_dma.hip -- rp_gemm's exact-fp32 main loop with LDS-DMA operand staging (precision 0, K a multiple of 32).
//
// Same tiling, operand layouts, k-step pairing, XCD-aware tile order, split-K and epilogue as gemm.hip's register-staged
// kernel (which stays for the bf16-limb precisions and ragged K); what changes is how a k-tile reaches LDS:
//   * global -> LDS directly (global_load_lds_dwordx4: 16 bytes per lane, 1 KB per wave instruction, no VGPR round trip and
//     no ds_write pass), issued one k-tile ahead into the other of TWO LDS stages, so a k-iteration is
//         issue DMA(kt+1) | 12 ds_read_b128 + 32 MFMA on stage kt | s_waitcnt vmcnt(0) ; s_barrier
//     -- one barrier per k-tile instead of two, and the staging costs 6 issue slots per wave instead of 6 loads + 6 ds_write
//     + 24 VGPRs.  Measured (tools/lab/gemm_lab.hip, MI355X, 64 pairs, sustained): qkv 161 -> 147 us, proj 66 -> 57,
//     fc1 212 -> 189, fc2 202 -> 173 (101 -> 111, 83 -> 95, 103 -> 115, 108 -> 126 TFLOP/s).
//   * the DMA writes LDS lane-linearly (wave-uniform base + 16 * lane), so a K-contiguous tile cannot be padded; its image is
//     [rows][8 chunks of 16 B] with the chunk index XOR-swizzled by ((row >> 1) & 7), applied to the SOURCE address of the DMA
//     and to the ds_read_b128 address: the 16 lanes of every ds_read_b128 group then hit 16 distinct 16-byte slots of the
//     256-byte bank row (SQ_LDS_BANK_CONFLICT = 0).  An MN-contiguous tile is [32 k][extent] as it lies in memory; its
//     operands are conflict-free ds_read_b32 rows.
//   * the DMA is inline asm: hipcc drains vmcnt(0) before the next ds_read when it sees the builtin in flight (the LDS write
//     is on the VM counter); completion is counted by hand -- s_waitcnt vmcnt(0) by every wave, then s_barrier, then reads.
// Things measured and NOT adopted (tools/lab): three stages with a counted vmcnt (72 KB LDS -> 2 workgroups per CU: -8 %),
// a persistent tile walk whose DMA pipeline crosses tile boundaries and whose C stores are deferred into the next tile's
// k-loop (+-0), staggered first-generation workgroups (+-0), nontemporal C stores (+1 %), v_mfma_f32_16x16x4_f32 (+1..2 %).
#include "gemm_common.h"

namespace rpgemm {
namespace {

// per-lane source offsets (bytes from the tile's first row / first k) of the NI DMA instructions a wave issues per k-tile
// for one operand tile of EXT rows.  LAY 0 ([EXT][K], K contiguous): instruction i covers rows (4 i + wave) * 8 .. + 7,
// lane -> (row = lane / 8, LDS chunk = lane % 8) fetches chunk ^ ((row >> 1) & 7).  LAY 1 ([K][EXT], EXT contiguous): the
// tile's 8 EXT chunks are taken in memory order, chunk c = (4 i + wave) * 64 + lane -> k = c / (EXT / 4), column 4 (c % (EXT / 4)).
// Rows / columns past the operand's extent are clamped onto its last valid 16-byte chunk (their products are never stored).
template <int LAY, int EXT>
RP_DEV void dma_offsets(unsigned (&v)[EXT / 32], int wave, int lane, int ld, int first, int extent) {
#pragma unroll
  for (int i = 0; i < EXT / 32; ++i) {
    if (LAY == 0) {
      const int row = (4 * i + wave) * 8 + (lane >> 3);
      const int grow = min(first + row, extent - 1) - first;        // may be negative only if first >= extent (never)
      v[i] = (unsigned)(grow * ld + (((lane & 7) ^ ((row >> 1) & 7)) << 2)) * 4u;
    } else {
      const int c = (4 * i + wave) * 64 + lane;
      const int k = c / (EXT / 4), col = (c % (EXT / 4)) * 4;
      const int gcol = min(first + col, extent - 4) - first;
      v[i] = (unsigned)(k * ld + gcol) * 4u;
    }
  }
}

template <int ALAY, int BLAY, int TM, int TN>
__global__ __launch_bounds__(256, (TM == 1 && TN == 3) ? 2 : 1) void gemm_dma_kernel(GemmP p) {
  constexpr int BM = 64 * TM, BN = 64 * TN, BK = 32;
  constexpr int A_FL = BM * BK, B_FL = BN * BK, STAGE = A_FL + B_FL;
  constexpr int CST = 32 * TN + 4;
  constexpr int C_FL = 4 * 32 * TM * CST;
  constexpr bool STAGED = !(ALAY == 1 && BLAY == 1);
  constexpr int LDS_FL = (STAGED && C_FL > 2 * STAGE) ? C_FL : 2 * STAGE;
  __shared__ __attribute__((aligned(16))) float lds[LDS_FL];

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, hi = lane >> 5;
  const int wm0 = (wave >> 1) * 32 * TM, wn0 = (wave & 1) * 32 * TN;
  // tile order: identical to gemm_kernel (XCD x owns row panels mt = x mod 8; split z pinned to XCD z % 8)
  const int ntn = (p.N + BN - 1) / BN, ntm = (p.M + BM - 1) / BM;
  int mt, nt;
  int zid = blockIdx.z;
  if (p.split_k > 1) {
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
  } else {
    mt = blockIdx.x / ntn;
    nt = blockIdx.x % ntn;
  }
  const int m0 = mt * BM, n0 = nt * BN;
  const int zb = p.split_k > 1 ? 0 : zid, zs = p.split_k > 1 ? zid : 0;
  const int kbeg = zs * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nkt = (kend - kbeg) / BK;              // the launcher guarantees whole k-tiles

  // wave-uniform operand bases at (tile origin, kbeg); per-lane DMA offsets
  const float* Ab = p.A + zb * p.sa + (ALAY == 0 ? (long long)m0 * p.lda + kbeg : (long long)kbeg * p.lda + m0);
  const float* Bb = p.B + zb * p.sb + (BLAY == 0 ? (long long)n0 * p.ldb + kbeg : (long long)kbeg * p.ldb + n0);
  const long long a_step = ALAY == 0 ? BK : (long long)BK * p.lda;
  const long long b_step = BLAY == 0 ? BK : (long long)BK * p.ldb;
  constexpr int NA = BM / 32, NB = BN / 32;
  unsigned va[NA], vb[NB];
  dma_offsets<ALAY, BM>(va, wave, lane, p.lda, m0, p.M);
  dma_offsets<BLAY, BN>(vb, wave, lane, p.ldb, n0, p.N);
  const unsigned lds0 = lds_byte_addr(lds);
  auto issue = [&](int kt, int st) {
    const unsigned as = lds0 + (st * STAGE + wave * 256) * 4, bs = as + A_FL * 4;       // 256 floats = 1 KB per wave instruction
    const float* a = uniform_ptr(Ab + kt * a_step);
    const float* b = uniform_ptr(Bb + kt * b_step);
#pragma unroll
    for (int i = 0; i < NA; ++i) glds16(a, va[i], as + i * 4096);
#pragma unroll
    for (int i = 0; i < NB; ++i) glds16(b, vb[i], bs + i * 4096);
  };

  // operand fragment addresses (floats from the stage base)
  const int key = (l31 >> 1) & 7;
  int coff[2][2];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int c = 0; c < 2; ++c) coff[h][c] = ((4 * hi + 2 * h + c) ^ key) << 2;
  const int arow = ALAY == 0 ? (wm0 + l31) * BK : wm0 + l31;
  const int brow = A_FL + (BLAY == 0 ? (wn0 + l31) * BK : wn0 + l31);

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j) acc[i][j] = zero16();

  if (nkt > 0) issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (int kt = 0; kt < nkt; ++kt) {
    const int st = kt & 1;
    if (kt + 1 < nkt) issue(kt + 1, st ^ 1);
    const float* base = lds + st * STAGE;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      float a[TM][8], b[TN][8];
#pragma unroll
      for (int i = 0; i < TM; ++i) {
        if (ALAY == 0) {
          const float4 x = ld4(base + arow + 32 * i * BK + coff[half][0]), y = ld4(base + arow + 32 * i * BK + coff[half][1]);
          a[i][0] = x.x; a[i][1] = x.y; a[i][2] = x.z; a[i][3] = x.w; a[i][4] = y.x; a[i][5] = y.y; a[i][6] = y.z; a[i][7] = y.w;
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) a[i][t] = base[(16 * hi + 8 * half + t) * BM + arow + 32 * i];
        }
      }
#pragma unroll
      for (int j = 0; j < TN; ++j) {
        if (BLAY == 0) {
          const float4 x = ld4(base + brow + 32 * j * BK + coff[half][0]), y = ld4(base + brow + 32 * j * BK + coff[half][1]);
          b[j][0] = x.x; b[j][1] = x.y; b[j][2] = x.z; b[j][3] = x.w; b[j][4] = y.x; b[j][5] = y.y; b[j][6] = y.z; b[j][7] = y.w;
        } else {
#pragma unroll
          for (int t = 0; t < 8; ++t) b[j][t] = base[(16 * hi + 8 * half + t) * BN + brow + 32 * j];
        }
      }
#pragma unroll
      for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
          for (int j = 0; j < TN; ++j) acc[i][j] = mfma32(a[i][t], b[j][t], acc[i][j]);
    }
    // every wave's DMA of k-tile kt+1 has landed, and every wave is done reading stage st, before anyone moves on
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  tile_epilogue<TM, TN, STAGED>(p, acc, lds, m0, n0, mt, zb, zid);
}

template <int ALAY, int BLAY, int TM, int TN>
void launch(const GemmP& p, int nz, hipStream_t st) {
  const int ntn = (p.N + 64 * TN - 1) / (64 * TN), ntm = (p.M + 64 * TM - 1) / (64 * TM);
  dim3 grid(ntm >= 16 ? ntn * ((ntm + 7) / 8) * 8 : ntn * ntm, 1, nz);
  if (p.split_k > 1) grid = dim3(ntn * ntm * ((p.split_k + 7) / 8) * 8, 1, 1);
  hipLaunchKernelGGL((gemm_dma_kernel<ALAY, BLAY, TM, TN>), grid, dim3(256), 0, st, p);
}

template <int ALAY, int BLAY>
void launch_tiles(const GemmP& p, int nz, int tm, int tn, hipStream_t st) {
  if (tm == 1) {
    if (tn == 1) return launch<ALAY, BLAY, 1, 1>(p, nz, st);
    if (tn == 2) return launch<ALAY, BLAY, 1, 2>(p, nz, st);
    return launch<ALAY, BLAY, 1, 3>(p, nz, st);
  }
  if (tn == 1) return launch<ALAY, BLAY, 2, 1>(p, nz, st);
  if (tn == 2) return launch<ALAY, BLAY, 2, 2>(p, nz, st);
  return launch<ALAY, BLAY, 2, 3>(p, nz, st);
}

}  // namespace

bool dma_eligible(const GemmP& p) {
  // whole k-tiles in every split, 16-byte DMA granules inside the operands (the clamps need one full chunk to land on)
  if (p.limbs != 0) return false;
  if (p.K % 32 != 0 || p.k_per_split % 32 != 0) return false;
  if (p.M < 4 || p.N < 4) return false;
  return true;
}

void launch_dma(const GemmP& p, int nz, int al, int bl, int tm, int tn, hipStream_t st) {
  if (al == 0 && bl == 0) return launch_tiles<0, 0>(p, nz, tm, tn, st);
  if (al == 0 && bl == 1) return launch_tiles<0, 1>(p, nz, tm, tn, st);
  if (al == 1 && bl == 0) return launch_tiles<1, 0>(p, nz, tm, tn, st);
  return launch_tiles<1, 1>(p, nz, tm, tn, st);
}

}  // namespace rpgemm
