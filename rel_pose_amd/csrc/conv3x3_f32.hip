// conv3x3_f32.hip -- the 3x3 / stride 1 / pad 1, 64 -> 64 convolutions of resnet.layer1 in EXACT fp32 (the headline configuration;
// src/model.py:131's BasicBlock convolutions, which the reference runs through cuDNN), forward and -- with the filter rotated and its
// channel roles swapped by the host -- input gradient:
//
//     Y[n, y, x, co] = sum over (r, s, ci) of X[n, y + r - 1, x + s - 1, ci] * W[co][r][s][ci]          (X zero outside the image)
//
// MIOpen's implicit-GEMM solvers run this shape at 0.71 (forward, 267 us at 128 images) and 0.61 (backward-data, 307 us) of the fp32
// MFMA peak.  Here:
//   * the FILTER LIVES IN REGISTERS: wave w owns 16 output channels and keeps their 9 x 64 filter values as the A operands of
//     v_mfma_f32_16x16x4_f32 (lane (co = l & 15, kq = l >> 4) holds W[co][tap][4 kk + kq]: 144 VGPRs), one wave per SIMD, persistent
//     workgroup per CU walking pairs of image rows;
//   * the INPUT is the B operand: a pair of output rows is 112 pixels = seven 16-pixel blocks; lane (px = l & 15, kq) reads
//     X[pixel + tap][4 kk + kq] from an LDS ring of six 58-position row slots (position 0 / 57 = the zero padding) whose pixel stride is
//     66 floats -- the 16 pixels x 2 channels of a half-wave hit 32 different banks -- with ONE conflict-free ds_read_b32 per MFMA, the tap column and the
//     k-step in the immediate offset (next to fp32 MFMAs LDS instructions are free, VALU instructions are not: profiles/r5_shadow_lab.txt).
//     16x16x4 because it holds the matrix pipe's rate at any occupancy and its 4-deep k-step is what makes the padded layout conflict-free;
//   * a tile (two rows) needs input rows y - 1 .. y + 2 and fetches only the two new ones, one tile ahead, through registers (the padded
//     layout rules LDS-DMA out; 7 loads + 14 ds_write_b64 per thread per 1008 MFMAs); rows outside the image read a seventh, permanently
//     zero slot (an address select per tile, no branch in the loop);
//   * operands are read one k-step ahead by asm ds_read_b32, one read behind each MFMA, with a counted wait per MFMA written out by hand
//     (hipcc sinks C++-level reads next to their MFMAs behind lgkmcnt(0)); the accumulators (co = rows, pixels = columns) leave as one
//     16-byte store per lane and block.
#include <type_traits>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

constexpr int C = 64, IW = 56, IH = 56;
constexpr int PS = 66;                         // floats per pixel position in LDS: 2 (mod 32) -- the LDS has 32 banks and serves 32 lanes per
                                               // cycle: lanes 0-31 = 16 pixels x k in {0, 1} land on banks 2 px + k, all different (68 = 4 mod 32
                                               // made pixels p and p + 8 collide: LDS conflict 0.49 in the counters)
constexpr int ROWF = (IW + 2) * PS + 28;       // floats of one row slot (58 positions + pad to 16 mod 32: the block that straddles the two rows
                                               // of a tile reads two slots)
constexpr int NSLOT = 6;                       // ring slots; slot 6 = a row of zeros
constexpr int TPI = IH / 2;                    // 28 tiles (row pairs) per image
constexpr int NBLK = 7;                        // 16-pixel blocks per tile

struct CvF {
  const float* x;       // [N,56,56,64]
  const float* w;       // [64 co][3][3][64 ci]
  float* y;             // [N,56,56,64]
  int ntiles;           // N * 28
  const float* res;     // null, or [N,56,56,64]: added to the result (the input gradient of a BasicBlock's first convolution + the gradient that
                        // arrives over the identity path: autograd's separate add pass, 309 MB, disappears)
  RpBnMask bn;          // bn.x != null: the result is masked by the ReLU of batch_norm(bn.x) and `stats` receives sums of g and g * xhat
  double* stats;        // null, or [gridDim][2][64]: per-workgroup sums of y and y^2 per output channel (the BatchNorm statistics of the
                        // OUTPUT, csrc/batchnorm.hip: rp_bn_stats_from_partials) -- the statistics pass over y is then not needed
  int dgrad;            // 0: filter W[co][r][s][ci] as it lies; 1: the input gradient's filter W'[ci][r][s][co] = W[co][2 - r][2 - s][ci], read
                        // from the SAME forward weight (strided, 144 loads per lane once per workgroup: no rotated copy, no extra launch)
};

template <int OFF> RP_DEV float rd32c(unsigned addr) {
  float v;
  asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return v;
}
template <int N, class F> RP_DEV void sforc(F&& f) {
  if constexpr (N > 0) {
    sforc<N - 1>(f);
    f(std::integral_constant<int, N - 1>{});
  }
}

// RES / BN: the epilogue options as TEMPLATE parameters -- as run-time branches of one kernel they cost the plain launches 15 us each (294 -> 406
// registers and a different schedule around the tile loop: profiles/r6_ab.txt), more than the passes they replace bring
template <bool RES, bool BN>
__global__ __launch_bounds__(256, 1) void conv3x3_c64_f32_kernel(CvF p) {
  __shared__ __attribute__((aligned(16))) float Xr[NSLOT + 1][ROWF];      // 107 968 B
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), l15 = lane & 15, kq = lane >> 4;
  const int G = gridDim.x, b = blockIdx.x;
  const int t0 = (int)((long long)p.ntiles * b / G), t1 = (int)((long long)p.ntiles * (b + 1) / G);
  if (t0 >= t1) {
    if (p.stats && tid < 2 * C) p.stats[(long long)b * 2 * C + tid] = 0.0;
    return;
  }
  double sd1[4] = {0.0, 0.0, 0.0, 0.0}, sd2[4] = {0.0, 0.0, 0.0, 0.0};      // this lane's sums of y, y^2 over its pixels, channels 16 wave + 4 kq + e
  const unsigned xs0 = lds_byte_addr(&Xr[0][0]);
  // staging: the two rows g, g + 1 (flattened (image, row) index) = 1792 float4, 7 per thread: float4 f -> row f / 896, pixel (f % 896) / 16
  float4 pre[7], pre2[7];
  auto gload = [&](float4 (&r)[7], int g) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int f = tid + 256 * i;
      const int row = min(g + f / 896, p.ntiles * 2 - 1);      // (clamped re-fetch at the very end: harmless)
      r[i] = ld4(p.x + ((long long)row * IW) * C + (f % 896) * 4);
    }
  };
  auto sstore = [&](const float4 (&v)[7], int g) {
#pragma unroll
    for (int i = 0; i < 7; ++i) {
      const int f = tid + 256 * i, r = f / 896, rem = f % 896;
      float2* d = reinterpret_cast<float2*>(&Xr[(g + r) % NSLOT][((rem >> 4) + 1) * PS + (rem & 15) * 4]);      // (8-byte aligned: PS is even)
      d[0] = make_float2(v[i].x, v[i].y);
      d[1] = make_float2(v[i].z, v[i].w);
    }
  };
  // prologue, every request out before anything is waited for (the launch's fixed cost is what separates 128 images from the kernel's
  // steady state): rows 2 t0 - 1 .. 2 t0 + 2 (0 .. 3 for the very first tile: row 3 is then simply early), the filter, the zeros
  const int a0 = max(2 * t0 - 1, 0);
  gload(pre, a0);
  gload(pre2, a0 + 2);
  // the filter of this wave's 16 output channels: A operand of k-step (tap, kk) = W[16 wave + l15][tap][4 kk + kq]
  float wreg[9][16];
  if (p.dgrad) {
    // W'[ci][r][s][co] = W[co][2 - r][2 - s][ci] out of the forward weight: A[m][k] of tap = w[k][8 - tap][m] -- the 16 lanes of a kq group read
    // 64 contiguous bytes
    const float* wp = p.w + (16 * wave + l15) + (long long)kq * (9 * C);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int kk = 0; kk < 16; ++kk) wreg[tap][kk] = wp[(8 - tap) * C + (long long)(4 * kk) * (9 * C)];
  } else {
    // forward: lane (m, kq) needs W[m][tap][4 kk + kq], 4-byte pieces 16 bytes apart = sixteen 64-byte requests per wave instruction (the
    // forward launches ran 12-15 us behind the input gradients).  Instead the four kq lanes of a channel read the row as 16-byte pieces (lane
    // kq: floats 16 j + 4 kq .. + 3) and transpose the 4 x 4 blocks among themselves (conv3x3_c128_f32.hip): v_permlane32_swap exchanges the
    // off-diagonal 2 x 2 blocks (lanes l, l + 32), v_permlane16_swap transposes inside them (lanes l, l + 16).
    const float* wp = p.w + (long long)(16 * wave + l15) * (9 * C) + 4 * kq;
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float4 v = ld4(wp + tap * C + 16 * j);
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
  // zero padding of the ring slots (staging only ever writes positions 1 .. 56) and the zero row
  for (int i = tid; i < NSLOT * 2 * PS; i += 256) Xr[i / (2 * PS)][((i / PS) & 1) * (IW + 1) * PS + (i % PS)] = 0.f;
  for (int i = tid; i < ROWF; i += 256) Xr[NSLOT][i] = 0.f;
  sstore(pre, a0);
  sstore(pre2, a0 + 2);
  __syncthreads();

  // BatchNorm-mask epilogue: this lane's four channels' mean, rstd * gamma (the forward's product), beta, rstd
  float bmu[4] = {0.f, 0.f, 0.f, 0.f}, brg[4] = {0.f, 0.f, 0.f, 0.f}, bbe[4] = {0.f, 0.f, 0.f, 0.f}, brs[4] = {0.f, 0.f, 0.f, 0.f};
  if constexpr (BN) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int ch = 16 * wave + 4 * kq + e;
      bmu[e] = p.bn.mean[ch]; brs[e] = p.bn.rstd[ch]; brg[e] = p.bn.rstd[ch] * p.bn.gamma[ch]; bbe[e] = p.bn.beta[ch];
    }
  }
  // per-lane pixel of each 16-pixel block: flattened index 16 j + l15 of the two rows -> (row 0 / 1, column)
  int orow[NBLK], ocol[NBLK];
#pragma unroll
  for (int j = 0; j < NBLK; ++j) {
    const int pxi = 16 * j + l15;
    orow[j] = pxi >= IW ? 1 : 0;
    ocol[j] = pxi - IW * orow[j];
  }

  for (int t = t0; t < t1; ++t) {
    const int y = 2 * (t % TPI), g0 = 2 * t;
    if (t + 1 < t1) gload(pre, g0 + 3);                  // next tile's two new rows
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
    for (int j = 0; j < NBLK; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    float4 rv[NBLK], bx[NBLK];                               // the residual / the BatchNorm input at this tile's outputs, requested a tile's worth of MFMAs early
    if constexpr (RES) {
#pragma unroll
      for (int j = 0; j < NBLK; ++j) rv[j] = ld4(p.res + (((long long)g0 + orow[j]) * IW + ocol[j]) * C + 16 * wave + 4 * kq);
    }
    if constexpr (BN) {
#pragma unroll
      for (int j = 0; j < NBLK; ++j) bx[j] = ld4(p.bn.x + (((long long)g0 + orow[j]) * IW + ocol[j]) * C + 16 * wave + 4 * kq);
    }
    // Two operand register sets, picked by the k-step's parity at compile time.  The seven reads of k-step k + 1 go out ONE BEHIND EACH
    // MFMA of k-step k (a block of seven reads per k-step left the matrix pipe drained while they issued: 19 % of the tile), and each MFMA
    // waits only for ITS operand: the LDS queue retires in order and exactly six younger reads are in flight in front of it -- lgkmcnt(6).
    float b0[NBLK], b1[NBLK];
#pragma unroll
    for (int j = 0; j < NBLK; ++j) b0[j] = rd32c<0>(xa[j][0]);
    sforc<144>([&](auto kc) {
      constexpr int k = kc, tap = k / 16, kk = k % 16;
      float (&bc)[NBLK] = (k & 1) ? b1 : b0;
      float (&bn)[NBLK] = (k & 1) ? b0 : b1;
      sforc<NBLK>([&](auto jc) {
        constexpr int j = jc;
        if constexpr (k + 1 < 144) asm volatile("s_waitcnt lgkmcnt(6)" : "+v"(bc[j]));
        else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bc[j]));
        acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[tap][kk], bc[j], acc[j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (k + 1 < 144) {
          constexpr int k1 = k + 1, tap1 = k1 / 16, kk1 = k1 % 16, r1 = tap1 / 3, s1 = tap1 % 3;
          constexpr int off = (s1 * PS + 4 * kk1) * 4;
          bn[j] = rd32c<off>(xa[j][r1]);
          __builtin_amdgcn_sched_barrier(0);
        }
      });
    });
    // D[m = co 4 kq + e][n = pixel l15]: four consecutive output channels of one pixel per lane.  The epilogue's two sums per channel: y and
    // y^2 (forward: BatchNorm statistics of the output), or -- bn -- g and g * xhat of the masked gradient g (the BatchNorm's backward sums);
    // fp32 over the tile's seven pixels, double across tiles
    float ta[4] = {0.f, 0.f, 0.f, 0.f}, tq[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < NBLK; ++j) {
      const long long off = (((long long)g0 + orow[j]) * IW + ocol[j]) * C + 16 * wave + 4 * kq;
      if constexpr (RES) { acc[j][0] += rv[j].x; acc[j][1] += rv[j].y; acc[j][2] += rv[j].z; acc[j][3] += rv[j].w; }
      if constexpr (BN) {
        const float4 xv = bx[j];
        const float xs[4] = {xv.x, xv.y, xv.z, xv.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float d = xs[e] - bmu[e];
          const float yv = __builtin_fmaf(d, brg[e], bbe[e]);          // = bn_affine (csrc/batchnorm.hip), bit for bit
          acc[j][e] = yv > 0.f ? acc[j][e] : 0.f;
          ta[e] += acc[j][e];
          tq[e] = fmaf(acc[j][e], d * brs[e], tq[e]);
        }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { ta[e] += acc[j][e]; tq[e] = fmaf(acc[j][e], acc[j][e], tq[e]); }
      }
      st4(p.y + off, make_float4(acc[j][0], acc[j][1], acc[j][2], acc[j][3]));
    }
    if (p.stats) {                                            // (wave-uniform)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        sd1[e] += (double)ta[e];
        sd2[e] += (double)tq[e];
      }
    }
    if (t + 1 < t1) sstore(pre, g0 + 3);                      // slots of rows g0 + 3, g0 + 4: not among this tile's g0 - 1 .. g0 + 2
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
    if (tid < 2 * C) {
      const double* r = red + (tid >> 6) * 1024 + (tid & 63) * 16;
      double a = r[0];
#pragma unroll
      for (int l = 1; l < 16; ++l) a += r[l];
      p.stats[(long long)b * 2 * C + tid] = a;                // [which][channel]
    }
  }
}

}  // namespace

extern "C" int rp_conv3x3_c64_f32_blocks(int N) {
  const int tiles = N * TPI;
  return tiles < 256 ? tiles : 256;
}

/* y [N,56,56,64] = conv3x3(x [N,56,56,64], w [64][3][3][64]), stride 1, pad 1, exact fp32 (NHWC memory; w = the memory of a channels-last
 * [64,64,3,3] weight).  input_gradient != 0: x is dY and the result is dX of the same convolution -- the filter w'[ci][r][s][co] =
 * w[co][2 - r][2 - s][ci] is read out of the forward weight w.  stats: NULL, or [rp_conv3x3_c64_f32_blocks(N)][2][64] doubles that receive the
 * per-workgroup sums of y and y^2 per channel (BatchNorm statistics of the output, finished by rp_bn_stats_from_partials with a zero pivot).
 * res: NULL, or a tensor of y's shape that is ADDED to the result in the epilogue (y = conv + res; the statistics then describe that sum).
 * bn: NULL, or the BatchNorm-mask epilogue (include/relpose_hip.h: RpBnMask; needs stats). */
extern "C" int rp_conv3x3_c64_f32(const float* x, const float* w, float* y, double* stats, const float* res, const RpBnMask* bn, int N, int H, int W,
                                  int input_gradient, void* stream) {
  if (!x || !w || !y || N <= 0) return RP_EBADSHAPE;
  if (H != IH || W != IW) return RP_EUNSUPPORTED;
  if (((uintptr_t)x | (uintptr_t)w | (uintptr_t)y | (uintptr_t)stats | (uintptr_t)res) & 15) return RP_EALIGN;
  if (bn && (!bn->x || !bn->mean || !bn->rstd || !bn->gamma || !bn->beta || !stats || ((uintptr_t)bn->x & 15))) return RP_EBADSHAPE;
  CvF p{x, w, y, N * TPI, res, bn ? *bn : RpBnMask{nullptr, nullptr, nullptr, nullptr, nullptr}, stats, input_gradient ? 1 : 0};
  const dim3 grid(rp_conv3x3_c64_f32_blocks(N));
  if (bn && res) return RP_EUNSUPPORTED;
  if (bn) hipLaunchKernelGGL((conv3x3_c64_f32_kernel<false, true>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else if (res) hipLaunchKernelGGL((conv3x3_c64_f32_kernel<true, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((conv3x3_c64_f32_kernel<false, false>), grid, dim3(256), 0, (hipStream_t)stream, p);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
