// bf16_path.h -- device helpers shared by the kernels of the bf16 data path (attention_bf16.hip, emm_bf16.hip): LDS-DMA of bf16
// tiles, transpose-read MFMA operands, the LDS swizzles that serve both read patterns, bf16 row stores of transposed accumulators.
#pragma once
#include "common.h"

namespace bf16path {

constexpr int NTOK = 576;
typedef unsigned short bf16_t;
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4* lds_s16x4_ptr;

RP_DEV void glds16b(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
               : "=&s"(keep) : "v"(voff), "s"(lds_byte_addr), "s"(sbase) : "memory");
}
RP_DEV const void* uniform_vptr(const void* p) {
  const unsigned long long v = (unsigned long long)p;
  const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
  return (const void*)(((unsigned long long)hi << 32) | lo);
}
RP_DEV unsigned lds_addr_of(const void* p) { return (unsigned)(size_t)(rp_lds_ptr_t)(p); }

// two transpose reads = one bf16x8 MFMA operand: slots 0-3 from the 4-row block at a0, slots 4-7 from the one at a1
RP_DEV bf16x8 tr_operand(const bf16_t* a0, const bf16_t* a1) {
  typedef short s16x8 __attribute__((ext_vector_type(8)));
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a0);
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_ptr)a1);
  s16x8 v;
  v[0] = lo[0]; v[1] = lo[1]; v[2] = lo[2]; v[3] = lo[3]; v[4] = hi[0]; v[5] = hi[1]; v[6] = hi[2]; v[7] = hi[3];
  return __builtin_bit_cast(bf16x8, v);
}
RP_DEV bf16x8 ld_bf16x8_lds(const bf16_t* p) { return *reinterpret_cast<const bf16x8*>(p); }

RP_DEV bf16x8 ones8() {
  typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
  u32x4_t w;
  w[0] = w[1] = w[2] = w[3] = 0x3f803f80u;
  return __builtin_bit_cast(bf16x8, w);
}

// LDS image of a K-type stage: row r = 8 chunks of 16 B, chunk c stored at slot c ^ ((r >> 1) & 7)  (ds_read_b128 by lane = row)
// LDS image of a V-type stage: chunk c stored at slot c ^ (((r >> 1) & 1) << 2): the four rows of a transpose-read block then lie in
// four different 64-byte quarters of the 256-byte bank row
RP_DEV int swz_k(int r) { return (r >> 1) & 7; }
RP_DEV int swz_v(int r) { return ((r >> 1) & 1) << 2; }

RP_DEV int swz_d(int r) { return (((r >> 1) & 1) << 2) | (((r >> 2) & 1) << 1) | ((r >> 3) & 1); }

// per-lane element offsets of the transpose reads of one 32-row tile in a swz_d image: [db][half] for the 8-row groups (rows
// 16 c2 + 8 half + 4 hi + (t16 >> 2), columns 16 g + 4 (t16 & 3) + 32 db); add 1024 c2 for the second 16 rows
RP_DEV void tr_offsets_d(int lane, int (&off)[2][2]) {
  const int hi = lane >> 5, t16 = lane & 15, g = (lane >> 4) & 1;
#pragma unroll
  for (int db = 0; db < 2; ++db)
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int row = 8 * half + 4 * hi + (t16 >> 2);
      off[db][half] = row * 64 + (((2 * g + ((t16 & 3) >> 1) + 4 * db) ^ swz_d(row)) << 3) + 4 * (t16 & 1);
    }
}

// A wave's transposed 32 x 64 accumulator tile (lane = owner row l31, register r of block b = column 32 b + acc_row(r, hi)), times mul,
// -> bf16 rows of `dst` (row stride ld elements) through 4 KB of this wave's LDS: whole 128-byte rows per store instruction instead of
// 8-byte pieces.  Chunk slot ^ (row & 7) spreads the writes over the banks.  Same wave writes and reads: no barrier.
RP_DEV void store_ownerT_bf16(bf16_t* Os, bf16_t* dst, int ld, int lane, const f32x16& o0, const f32x16& o1, float mul) {
  const int l31 = lane & 31, hi = lane >> 5;
#pragma unroll
  for (int gq = 0; gq < 4; ++gq) {
    const uint2 w0 = make_uint2(pk_bf16(o0[4 * gq] * mul, o0[4 * gq + 1] * mul), pk_bf16(o0[4 * gq + 2] * mul, o0[4 * gq + 3] * mul));
    const uint2 w1 = make_uint2(pk_bf16(o1[4 * gq] * mul, o1[4 * gq + 1] * mul), pk_bf16(o1[4 * gq + 2] * mul, o1[4 * gq + 3] * mul));
    *reinterpret_cast<uint2*>(Os + l31 * 64 + ((gq ^ (l31 & 7)) << 3) + 4 * hi) = w0;
    *reinterpret_cast<uint2*>(Os + l31 * 64 + (((gq + 4) ^ (l31 & 7)) << 3) + 4 * hi) = w1;
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 8 * i + (lane >> 3), ch = lane & 7;
    const uint4 w = *reinterpret_cast<const uint4*>(Os + row * 64 + ((ch ^ (row & 7)) << 3));
    *reinterpret_cast<uint4*>(dst + (long long)row * ld + 8 * ch) = w;
  }
}

// column sums of the same tile over its 32 owner rows (x mul): part[0..63], fixed order (see attention.hip: colsum_ownerT)
RP_DEV void colsum_ownerT_bf(float* part, int l31, int hi, const f32x16& o0, const f32x16& o1, float mul) {
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


}  // namespace bf16path
