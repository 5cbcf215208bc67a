// conv3x3_c128_f32.hip -- the 3x3 / stride 1 / pad 1 convolutions with 128 INPUT channels on 28 x 28 maps in EXACT fp32: resnet.layer2's
// three 128 -> 128 convolutions (src/model.py:132; forward and -- with the filter's roles swapped in-kernel -- input gradient) and the
// forward of extractor_final_conv.conv1, 128 -> 192 with bias (src/modules/extractor.py:9,51):
//
//     Y[n, y, x, co] = bias[co] + sum over (r, s, ci) of X[n, y + r - 1, x + s - 1, ci] * W[co][r][s][ci]        (X zero outside the image)
//
// MIOpen's implicit-GEMM solvers run these at 0.70-0.73 of the fp32 MFMA peak (267 / 263 / 385 us at 128 images, profiles/r2_conv_probe.txt).
// Same design as conv3x3_f32.hip (resnet.layer1, 64 channels), re-cut for 128 input channels:
//   * the FILTER LIVES IN REGISTERS: a wave owns 16 output channels and keeps their 9 x 128 filter values as the A operands of
//     v_mfma_f32_16x16x4_f32 (lane (co = l & 15, kq = l >> 4) holds W[co][tap][4 kk + kq]: 288 VGPRs, one wave per SIMD); a workgroup = 64
//     output channels, so the 128 / 192 output channels are 2 / 3 CHANNEL GROUPS of workgroups that walk the same tiles -- placed on one
//     XCD (workgroup b runs on XCD b % 8), where the second and third reader of a tile hit the L2;
//   * a tile = FOUR image rows = 112 pixels = seven 16-pixel blocks (28 = 7 x 4 rows per image: no ragged tile); lane (px = l & 15, kq)
//     reads X[pixel + tap][4 kk + kq] from an LDS ring of six 30-position row slots (positions 0 / 29 = the zero padding) with a pixel
//     stride of 130 floats (= 2 mod 32: the 16 pixels x 2 channels of a half-wave hit 32 different banks) and a slot stride of 24 mod 32
//     (= 2 x 28: a block that straddles two rows still reads 32 different banks); ONE conflict-free ds_read_b32 behind each MFMA, tap
//     column and k-step in the immediate offset;
//   * a tile needs input rows y - 1 .. y + 4 and fetches only the four new ones through registers while the previous tile computes; they
//     replace rows the previous tile was still reading, so they are written between two barriers (57 KB per 2016 MFMAs per wave: ~1 %);
//     rows outside the image read a seventh, permanently zero slot (an address select per tile, no branch in the loop).
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

constexpr int CI = 128, IW = 28, IH = 28;
constexpr int RT = 4;                          // image rows per tile
constexpr int PS = CI + 2;                     // floats per pixel position in LDS: 2 (mod 32)
constexpr int ROWF = (IW + 2) * PS + 28;       // floats of one row slot: 3928 = 24 (mod 32) = 2 x IW, see above
constexpr int NSLOT = 6;                       // ring slots (rows y - 1 .. y + 4 of a tile); slot 6 = a row of zeros
constexpr int TPI = IH / RT;                   // 7 tiles per image
constexpr int NBLK = 7;                        // 16-pixel blocks per tile
constexpr int KS = CI / 4;                     // 32 k-steps per tap
constexpr int F4ROW = IW * CI / 4;             // 896 float4 per image row
static_assert(ROWF % 32 == 24 && PS % 32 == 2 && RT * IW == 16 * NBLK && RT * F4ROW == 14 * 256, "layout");

struct CvG {
  const float* x;       // [N,28,28,128]
  const float* w;       // forward: [CO][3][3][128]; input gradient: the forward weight [128 (= this kernel's K)][3][3][CO = 128]
  const float* bias;    // [CO] or null
  float* y;             // [N,28,28,CO]
  double* stats;        // null, or [tile chunks][2][CO]: per-chunk sums of y and y^2 per output channel (BatchNorm statistics of the output for
                        // rp_bn_stats_from_partials; a workgroup writes the 64 channels of its group)
  int ntiles;           // N * 7
  int CO;               // 128 or 192 (multiple of 64)
  int dgrad;            // 1: the filter W'[ci][r][s][co] = W[co][2 - r][2 - s][ci] is read out of the forward weight (CO == 128 only)
};

template <int OFF> RP_DEV float rd32g(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N, class F> RP_DEV void sforg(F&& f) {
  if constexpr (N > 0) {
    sforg<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

__global__ __launch_bounds__(256, 1) void conv3x3_c128_f32_kernel(CvG p) {
  __shared__ __attribute__((aligned(16))) float Xr[NSLOT + 1][ROWF];      // 109 984 B
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
  // workgroup b = 8 q + x runs on XCD x: channel group q % ncg of tile chunk (q / ncg) * 8 + x -- the groups of a chunk share an L2
  const int ncg = p.CO >> 6, q = blockIdx.x >> 3;
  const int cg = q % ncg, chunk = (q / ncg) * 8 + (blockIdx.x & 7), nchunk = (gridDim.x >> 3) / ncg * 8;
  if (chunk >= nchunk) return;
  const int t0 = (int)((long long)p.ntiles * chunk / nchunk), t1 = (int)((long long)p.ntiles * (chunk + 1) / nchunk);
  if (t0 >= t1) {
    if (p.stats && tid < 128) p.stats[((long long)chunk * 2 + (tid >> 6)) * p.CO + 64 * cg + (tid & 63)] = 0.0;
    return;
  }
  double sd1[4] = {0.0, 0.0, 0.0, 0.0}, sd2[4] = {0.0, 0.0, 0.0, 0.0};      // this lane's sums of y, y^2 over its pixels, channels 16 wave + 4 kq + e
  const unsigned xs0 = lds_byte_addr(&Xr[0][0]);
  const int lastrow = p.ntiles * RT - 1;
  // staging: NR consecutive rows from flattened (image, row) index g = NR * 896 float4, 3.5 NR per thread: float4 f -> row f / 896, pixel
  // (f % 896) / 32, channels 4 (f % 32)
  float4 pre[14], pre2[7];
  auto gload4 = [&](int g) {
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int f = tid + 256 * i;
      const int row = min(g + f / F4ROW, lastrow);                 // (clamped re-fetch at the very end: harmless)
      pre[i] = ld4(p.x + (long long)row * (IW * CI) + (f % F4ROW) * 4);
    }
  };
  auto sstore4 = [&](int g) {
#pragma unroll
    for (int i = 0; i < 14; ++i) {
      const int f = tid + 256 * i, r = f / F4ROW, rem = f % F4ROW;
      float2* d = reinterpret_cast<float2*>(&Xr[(g + r) % NSLOT][((rem >> 5) + 1) * PS + (rem & 31) * 4]);      // (8-byte aligned: PS, ROWF even)
      d[0] = make_float2(pre[i].x, pre[i].y);
      d[1] = make_float2(pre[i].z, pre[i].w);
    }
  };
  // prologue: rows a0 .. a0 + 5 (a0 = the first tile's row y - 1, or 0 at the very start: row 5 is then simply early)
  const int a0 = max(RT * t0 - 1, 0);
  gload4(a0);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int f = tid + 256 * i;
    const int row = min(a0 + 4 + f / F4ROW, lastrow);
    pre2[i] = ld4(p.x + (long long)row * (IW * CI) + (f % F4ROW) * 4);
  }
  // the filter of this wave's 16 output channels: A operand of k-step (tap, kk) = W[64 cg + 16 wave + l15][tap][4 kk + kq]
  float wreg[9][KS];
  if (p.dgrad) {
    // the forward weight is [K = 128][3][3][CO]: A[m][k] of tap = w[k][8 - tap][m] -- the 16 lanes of a kq group read 64 contiguous bytes
    const float* wp = p.w + (64 * cg + 16 * wave + l15) + (long long)kq * (9 * p.CO);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < KS; ++kk) wreg[tap][kk] = wp[(8 - tap) * p.CO + (long long)(4 * kk) * (9 * p.CO)];
  } else {
    // forward: lane (m, kq) needs W[m][tap][4 kk + kq], 4-byte pieces 16 bytes apart -- read as such they cost 16 separate 64-byte
    // requests per wave instruction (33 us of the launch).  Instead the four kq lanes of a channel read the row as 16-byte pieces (lane kq:
    // floats 16 j + 4 kq .. + 3) and transpose the 4 x 4 blocks among themselves: v_permlane32_swap exchanges the off-diagonal 2 x 2
    // blocks (lanes l, l + 32), v_permlane16_swap transposes inside them (lanes l, l + 16).
    const float* wp = p.w + (long long)(64 * cg + 16 * wave + l15) * (9 * CI) + 4 * kq;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < KS / 4; ++j) {
        const float4 v = ld4(wp + tap * CI + 16 * j);
        unsigned r0 = __builtin_bit_cast(unsigned, v.x), r1 = __builtin_bit_cast(unsigned, v.y), r2 = __builtin_bit_cast(unsigned, v.z),
                 r3 = __builtin_bit_cast(unsigned, v.w);
        auto a = __builtin_amdgcn_permlane32_swap(r0, r2, false, false);
        r0 = a[0]; r2 = a[1];
        auto b = __builtin_amdgcn_permlane32_swap(r1, r3, false, false);
        r1 = b[0]; r3 = b[1];
        auto c = __builtin_amdgcn_permlane16_swap(r0, r1, false, false);
        r0 = c[0]; r1 = c[1];
        auto d = __builtin_amdgcn_permlane16_swap(r2, r3, false, false);
        r2 = d[0]; r3 = d[1];
        wreg[tap][4 * j + 0] = __builtin_bit_cast(float, r0);
        wreg[tap][4 * j + 1] = __builtin_bit_cast(float, r1);
        wreg[tap][4 * j + 2] = __builtin_bit_cast(float, r2);
        wreg[tap][4 * j + 3] = __builtin_bit_cast(float, r3);
      }
  }
  float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
  if (p.bias) bv = ld4(p.bias + 64 * cg + 16 * wave + 4 * kq);
  // zero padding of the ring slots (staging only ever writes positions 1 .. 28) and the zero row
  for (int i = tid; i < NSLOT * 2 * PS; i += 256) Xr[i / (2 * PS)][((i / PS) & 1) * (IW + 1) * PS + (i % PS)] = 0.f;
  for (int i = tid; i < ROWF; i += 256) Xr[NSLOT][i] = 0.f;
  sstore4(a0);
#pragma unroll
  for (int i = 0; i < 7; ++i) {
    const int f = tid + 256 * i, r = f / F4ROW, rem = f % F4ROW;
    float2* d = reinterpret_cast<float2*>(&Xr[(a0 + 4 + r) % NSLOT][((rem >> 5) + 1) * PS + (rem & 31) * 4]);
    d[0] = make_float2(pre2[i].x, pre2[i].y);
    d[1] = make_float2(pre2[i].z, pre2[i].w);
  }
  __syncthreads();

  // per-lane pixel of each 16-pixel block: flattened index 16 j + l15 of the four rows -> (row 0 .. 3, column)
  int orow[NBLK], ocol[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) {
    const int pxi = 16 * j + l15;
    orow[j] = pxi / IW;
    ocol[j] = pxi - IW * orow[j];
  }

  for (int t = t0; t < t1; ++t) {
    const int y = RT * (t % TPI), g0 = RT * t;
    if (t + 1 < t1) gload4(g0 + RT + 1);                 // next tile's four new rows g0 + 5 .. g0 + 8
    // operand addresses: block j, tap row r -> slot of input row g0 + orow + r - 1 (zero slot outside the image) + column * PS + kq
    unsigned xa[NBLK][3];
    const int m0 = (g0 + NSLOT - 1) % NSLOT;             // slot of row g0 - 1
#pragma unroll
    for (int j = 0; j < NBLK; ++j)
#pragma unroll
      for (int r = 0; r < 3; ++r) {
        const int yin = y + orow[j] + r - 1;
        int slot = m0 + orow[j] + r;
        slot = slot >= NSLOT ? slot - NSLOT : slot;
        slot = (yin < 0 || yin >= IH) ? NSLOT : slot;
        xa[j][r] = xs0 + (unsigned)(slot * ROWF + ocol[j] * PS + kq) * 4u;
      }
    f32x4 acc[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) acc[j] = f32x4{bv.x, bv.y, bv.z, bv.w};
    // Two operand register sets, picked by the k-step's parity at compile time; the seven reads of k-step k + 1 go out ONE BEHIND EACH MFMA
    // of k-step k, and each MFMA waits only for ITS operand: the LDS queue retires in order and exactly six younger reads are in flight in
    // front of it -- lgkmcnt(6) (conv3x3_f32.hip).
    float b0[NBLK], b1[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) b0[j] = rd32g<0>(xa[j][0]);
    sforg<9 * KS>([&](auto kc) {
      constexpr int k = kc, tap = k / KS, kk = k % KS;
      float (&bc)[NBLK] = (k & 1) ? b1 : b0;
      float (&bn)[NBLK] = (k & 1) ? b0 : b1;
      sforg<NBLK>([&](auto jc) {
        constexpr int j = jc;
        if constexpr (k + 1 < 9 * KS) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bc[j]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bc[j]));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][kk], bc[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k + 1 < 9 * KS) {
          constexpr int k1 = k + 1, tap1 = k1 / KS, kk1 = k1 % KS, r1 = tap1 / 3, s1 = tap1 % 3;
          constexpr int off = (s1 * PS + 4 * kk1) * 4;
          bn[j] = rd32g<off>(xa[j][r1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
    // D[m = co 4 kq + e][n = pixel l15]: four consecutive output channels of one pixel per lane
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
      float* o = p.y + (((long long)g0 + orow[j]) * IW + ocol[j]) * p.CO + 64 * cg + 16 * wave + 4 * kq;
      st4(o, make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]));
    }
    if (p.stats) {                                            // (wave-uniform) fp32 over the tile's seven pixels, double across tiles
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float a = 0.f, q = 0.f;
#pragma unroll
        for (int j = 0; j < NBLK; ++j) { a += acc[j][e]; q = fmaf(acc[j][e], acc[j][e], q); }
        sd1[e] += (double)a;
        sd2[e] += (double)q;
      }
    }
    if (t + 1 < t1) {
      __syncthreads();                                        // rows g0 - 1 .. g0 + 2 are dead for everybody: their slots take g0 + 5 .. g0 + 8
      sstore4(g0 + RT + 1);
    }
    __syncthreads();
  }
  if (p.stats) {
    // the 16 pixel lanes of a (wave, kq) group are summed in a fixed order through LDS (the ring is dead: everybody is past the last barrier)
    double* red = reinterpret_cast<double*>(&Xr[0][0]);       // [2][64 channels][16 lanes]
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = 16 * wave + 4 * kq + e;
      red[ch * 16 + l15] = sd1[e];
      red[1024 + ch * 16 + l15] = sd2[e];
    }
    __syncthreads();
    if (tid < 128) {
      const double* r = red + (tid >> 6) * 1024 + (tid & 63) * 16;
      double a = r[0];
#pragma unroll
      for (int l = 1; l < 16; ++l) a += r[l];
      p.stats[((long long)chunk * 2 + (tid >> 6)) * p.CO + 64 * cg + (tid & 63)] = a;
    }
  }
}

}  // namespace

/* workgroups of a launch: CO / 64 channel groups x tile chunks, whole groups of 8 (one per XCD), at most one workgroup per CU; the number of
 * tile chunks (= rows of the statistics partials) is blocks / (CO / 64) */
extern "C" int rp_conv3x3_c128_f32_blocks(int N, int CO) {
  if (N <= 0 || CO <= 0 || CO % 64) return 0;
  const int ncg = CO / 64, tiles = N * TPI;
  int per8 = 32 / ncg;                                     // chunks per XCD slot: 8 * per8 * ncg <= 256 workgroups
  while (per8 > 1 && 8 * (per8 - 1) >= tiles) --per8;      // no empty chunks
  return 8 * per8 * ncg;
}

/* y [N,28,28,CO] = bias + conv3x3(x [N,28,28,128], w [CO][3][3][128]), stride 1, pad 1, exact fp32 (NHWC memory; w = the memory of a
 * channels-last [CO,128,3,3] weight; CO = 128 or 192; bias [CO] or null).  input_gradient != 0 (CO == 128): x is dY and the result is dX of
 * the 128 -> 128 convolution whose FORWARD weight is w -- the filter w'[ci][r][s][co] = w[co][2 - r][2 - s][ci] is read out of it.  stats: NULL,
 * or [rp_conv3x3_c128_f32_blocks(N, CO) / (CO / 64)][2][CO] doubles = per-chunk sums of y and y^2 per channel (rp_bn_stats_from_partials). */
extern "C" int rp_conv3x3_c128_f32(const float* x, const float* w, const float* bias, float* y, double* stats, int N, int H, int W, int CO,
                                   int input_gradient, void* stream) {
  if (!x || !w || !y || N <= 0) return RP_EBADSHAPE;
  if (H != IH || W != IW || (CO != 128 && CO != 192) || (input_gradient && (CO != 128 || bias))) return RP_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)bias | (uintptr_t)stats) & 15) return RP_EALIGN;
  CvG p{x, w, bias, y, stats, N * TPI, CO, input_gradient ? 1 : 0};
  hipLaunchKernelGGL(conv3x3_c128_f32_kernel, dim3(rp_conv3x3_c128_f32_blocks(N, CO)), dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
