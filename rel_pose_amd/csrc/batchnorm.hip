// batchnorm.hip -- channels-last BatchNorm2d fused with the residual add and ReLU that follow it in the CNN front-end
// (torchvision BasicBlock: relu(bn1(.)), relu(bn2(.) + identity); reference src/modules/extractor.py:51-65:
// relu(norm1(.)), relu(norm2(.)), relu(norm3(downsample) + y)), forward and backward.
//
// Why (profiles/r1_full_step_summary.txt): around MIOpen's convolutions the front-end spent 4.7 ms per step in separate
// BatchNorm (2 passes forward, 2 backward), ReLU (forward clamp, backward mask) and residual-add kernels, all HBM-bound
// over the same [N*H*W, C] activations.  Here a BatchNorm+add+ReLU is 2 passes forward (column statistics; normalise + add +
// clamp) and 2 backward (masked column sums; dx), i.e. the ReLU and add passes disappear.
//
// Layout: x is [R, C] with R = N*H*W rows and C (multiple of 4, <= 256) contiguous channels -- a channels-last NCHW tensor.
// Statistics: per-thread fp32 partial sums over <= 64 rows, everything above that in double (block partials and the
// final combine), so E[x^2] - mean^2 has no cancellation problem at R = 1.6 M rows.
#include <stdlib.h>
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

constexpr int BN_MAXC = 256;

// Element type of the ACTIVATIONS and their gradients (x, y, residual, dy, dx, the pooled tensors): float, or bf16 storage for the
// bf16 configuration (BASELINE.json configs[4]: MIOpen's bf16 convolutions read and write bf16 channels-last tensors; keeping the
// BatchNorm / ReLU / pool passes between them in bf16 halves their HBM traffic and removes the fp32 <-> bf16 cast kernels around every
// convolution -- 6.3 ms of the 34 ms step at 128 pairs, profiles/r2_bf16_128pairs_full_step_summary.txt).  All arithmetic, the
// statistics and the per-channel parameters stay fp32 / double; bf16 values are widened exactly on load and rounded to nearest-even
// on store.
typedef unsigned short bf16s;      // storage only
template <typename T> RP_DEV float4 ldv(const T* p);
template <> RP_DEV float4 ldv<float>(const float* p) { return ld4(p); }
template <> RP_DEV float4 ldv<bf16s>(const bf16s* p) {
  const uint2 w = *reinterpret_cast<const uint2*>(p);
  return make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                     __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
}
template <typename T> RP_DEV void stv(T* p, float4 v);
template <> RP_DEV void stv<float>(float* p, float4 v) { st4(p, v); }
template <> RP_DEV void stv<bf16s>(bf16s* p, float4 v) { *reinterpret_cast<uint2*>(p) = make_uint2(pk_bf16(v.x, v.y), pk_bf16(v.z, v.w)); }
// G float4 groups (4 G channels) per thread and access: 1 for fp32 (16 bytes); 2 for bf16 storage when C % 8 == 0 -- with 8-byte accesses the bf16
// passes were INSTRUCTION-bound (the same bytes took 1.5x the fp32 kernels' time: profiles/r6_ab.txt), with 16-byte accesses they are HBM-bound
template <typename T, int G> RP_DEV void ldg(const T* p, float4 (&v)[G]) {
  if constexpr (G == 1) {
    v[0] = ldv<T>(p);
  } else {
    static_assert(sizeof(T) == 2 && G == 2, "two groups per access: bf16 storage only");
    const uint4 w = *reinterpret_cast<const uint4*>(p);
    v[0] = make_float4(__builtin_bit_cast(float, w.x << 16), __builtin_bit_cast(float, w.x & 0xffff0000u),
                       __builtin_bit_cast(float, w.y << 16), __builtin_bit_cast(float, w.y & 0xffff0000u));
    v[1] = make_float4(__builtin_bit_cast(float, w.z << 16), __builtin_bit_cast(float, w.z & 0xffff0000u),
                       __builtin_bit_cast(float, w.w << 16), __builtin_bit_cast(float, w.w & 0xffff0000u));
  }
}
template <typename T, int G> RP_DEV void stg(T* p, const float4 (&v)[G]) {
  if constexpr (G == 1) {
    stv<T>(p, v[0]);
  } else {
    *reinterpret_cast<uint4*>(p) = make_uint4(pk_bf16(v[0].x, v[0].y), pk_bf16(v[0].z, v[0].w), pk_bf16(v[1].x, v[1].y), pk_bf16(v[1].z, v[1].w));
  }
}
static bool bn_wide(int bf16, int C) {
  static const bool off = getenv("RP_BN_NARROW") != nullptr;       // A/B aid: 8-byte bf16 accesses as before
  return bf16 && (C % 8) == 0 && !off;
}
template <typename T> RP_DEV float ld1(const T* p);
template <> RP_DEV float ld1<float>(const float* p) { return *p; }
template <> RP_DEV float ld1<bf16s>(const bf16s* p) { return __builtin_bit_cast(float, (unsigned)(*p) << 16); }

// the normalisation, written with explicit fmas: the backward kernels re-evaluate it to rebuild the ReLU mask from x alone
// (no residual), and must get bit-identical values
RP_DEV float4 bn_affine(const float4 xv, const float4 mu, const float4 rs, const float4 ga, const float4 be) {
  float4 v;
  v.x = __builtin_fmaf(xv.x - mu.x, rs.x * ga.x, be.x); v.y = __builtin_fmaf(xv.y - mu.y, rs.y * ga.y, be.y);
  v.z = __builtin_fmaf(xv.z - mu.z, rs.z * ga.z, be.z); v.w = __builtin_fmaf(xv.w - mu.w, rs.w * ga.w, be.w);
  return v;
}

inline int bn_rows_per_block(long long R) {
  long long rpb = (R + 1023) / 1024;       // ~1024 blocks (4 per CU)
  if (rpb < 128) rpb = 128;
  return (int)rpb;
}

// The stem is conv -> BatchNorm -> ReLU -> 3x3/2 max-pool (reference src/model.py:127-130).  Fused ("POOL" instantiations): the
// incoming gradient of the BatchNorm OUTPUT at input pixel (n, ih, iw) is not read from memory but gathered from the pooled
// gradient dp [N,OH,OW,C] through the stored window positions (every pixel looks at the <= 4 windows that contain it: exactly
// maxpool_bwd_kernel's sum, same order), so the [N,H,W,C] pool-backward tensor is never written nor re-read twice.
struct PoolSrc {
  const void* dp;
  const unsigned char* idx;
  int H, W, OH, OW;
};

template <typename T>
RP_DEV float4 pool_gather(const PoolSrc& ps, unsigned row, int C, int c) {     // row = (n * H + ih) * W + iw (< 2^31), c = first of 4 channels
  const unsigned t = row / (unsigned)ps.W;                                       // 32-bit index arithmetic: these run per element
  const int iw = (int)(row - t * (unsigned)ps.W);
  const unsigned nn = t / (unsigned)ps.H;
  const int ih = (int)(t - nn * (unsigned)ps.H), n = (int)nn;
  float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
  const int oh0 = ih >> 1, ow0 = iw >> 1;
#pragma unroll
  for (int a = 0; a < 2; ++a) {
    const int oh = oh0 + a;
    const int kh = ih - (2 * oh - 1);
    if (oh >= ps.OH || kh < 0 || kh > 2) continue;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int ow = ow0 + b;
      const int kw = iw - (2 * ow - 1);
      if (ow >= ps.OW || kw < 0 || kw > 2) continue;
      const unsigned o = (((unsigned)n * ps.OH + oh) * ps.OW + ow) * C + c;       // pooled tensor < 2^32 elements (checked by the host)
      const uchar4 k4 = *reinterpret_cast<const uchar4*>(ps.idx + o);
      const float4 d = ldv<T>(static_cast<const T*>(ps.dp) + o);
      const int k = kh * 3 + kw;
      if (k4.x == k) g.x += d.x;
      if (k4.y == k) g.y += d.y;
      if (k4.z == k) g.z += d.z;
      if (k4.w == k) g.w += d.w;
    }
  }
  return g;
}

// stage 1: column sums of (a, b) over this block's rows.  MODE 0: a = x, b = x*x.
// MODE 1: g = dy * (relu ? y > 0 : 1); a = g, b = g * xhat, xhat = (x - mean) * rstd; optionally stores g.
template <int MODE, bool POOL, typename T>
__global__ __launch_bounds__(256) void bn_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                        const T* __restrict__ y, const float* __restrict__ mean,
                                                        const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, T* __restrict__ gout,
                                                        double* __restrict__ partial, long long R, int C, int rpb, int relu,
                                                        PoolSrc ps = PoolSrc{}) {
  __shared__ double red[2][256][4];
  const int c4n = C >> 2, nrl = 256 / c4n;
  const int tid = threadIdx.x, cg = tid % c4n, rl = tid / c4n;
  const long long r0 = (long long)blockIdx.x * rpb, r1 = min(R, r0 + rpb);
  double a0[4] = {0, 0, 0, 0}, b0[4] = {0, 0, 0, 0};
  if (rl < nrl) {
    float4 mu = make_float4(0.f, 0.f, 0.f, 0.f), rs = mu, ga = mu, be = mu;
    if (MODE == 1) { mu = ld4(mean + 4 * cg); rs = ld4(rstd + 4 * cg); }
    // MODE 0 sums (x - pivot) and (x - pivot)^2 with pivot = row 0 of the tensor: E[d^2] - E[d]^2 then has no cancellation
    // even when |mean| >> std (d is O(std) whenever the data are concentrated anywhere)
    const float4 pv = MODE == 0 ? ldv<T>(x + 4 * cg) : make_float4(0.f, 0.f, 0.f, 0.f);
    const bool remask = MODE == 1 && relu && y == nullptr;       // ReLU mask rebuilt from x (no residual was added)
    if (remask) { ga = ld4(gamma + 4 * cg); be = ld4(beta + 4 * cg); }
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    int cnt = 0;
    auto acc1 = [&](const float4 xv, float4 g, const float4 yv) {
      if (MODE == 0) {
        const float dx_ = xv.x - pv.x, dy_ = xv.y - pv.y, dz_ = xv.z - pv.z, dw_ = xv.w - pv.w;
        sa.x += dx_; sa.y += dy_; sa.z += dz_; sa.w += dw_;
        sb.x += dx_ * dx_; sb.y += dy_ * dy_; sb.z += dz_ * dz_; sb.w += dw_ * dw_;
      } else {
        sa.x += g.x; sa.y += g.y; sa.z += g.z; sa.w += g.w;
        sb.x += g.x * (xv.x - mu.x) * rs.x; sb.y += g.y * (xv.y - mu.y) * rs.y;
        sb.z += g.z * (xv.z - mu.z) * rs.z; sb.w += g.w * (xv.w - mu.w) * rs.w;
      }
    };
    auto mask = [&](float4 g, const float4 yv) {
      g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      return g;
    };
    constexpr int U = 4;                     // rows in flight per thread (independent 16-byte loads)
    long long r = r0 + rl;
    for (; r + (U - 1) * (long long)nrl < r1; r += (long long)U * nrl) {
      float4 xv[U], g[U], yv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        const long long off = (r + (long long)u * nrl) * C + 4 * cg;
        xv[u] = ldv<T>(x + off);
        if (MODE == 1) {
          g[u] = POOL ? pool_gather<T>(ps, (unsigned)(r + (long long)u * nrl), C, 4 * cg) : ldv<T>(dy + off);
          if (relu && !remask) yv[u] = ldv<T>(y + off);
        }
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        if (MODE == 1) {
          if (remask) yv[u] = bn_affine(xv[u], mu, rs, ga, be);
          if (relu) g[u] = mask(g[u], yv[u]);
          if (gout) stv<T>(gout + (r + (long long)u * nrl) * C + 4 * cg, g[u]);
        }
        acc1(xv[u], g[u], yv[u]);
      }
      cnt += U;
      if (cnt >= 64) {       // flush the fp32 partials into double
        a0[0] += sa.x; a0[1] += sa.y; a0[2] += sa.z; a0[3] += sa.w;
        b0[0] += sb.x; b0[1] += sb.y; b0[2] += sb.z; b0[3] += sb.w;
        sa = make_float4(0.f, 0.f, 0.f, 0.f); sb = sa; cnt = 0;
      }
    }
    for (; r < r1; r += nrl) {
      const long long off = r * C + 4 * cg;
      const float4 xv = ldv<T>(x + off);
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), yv = g;
      if (MODE == 1) {
        g = POOL ? pool_gather<T>(ps, (unsigned)r, C, 4 * cg) : ldv<T>(dy + off);
        if (relu) { yv = remask ? bn_affine(xv, mu, rs, ga, be) : ldv<T>(y + off); g = mask(g, yv); }
        if (gout) stv<T>(gout + off, g);
      }
      acc1(xv, g, yv);
    }
    a0[0] += sa.x; a0[1] += sa.y; a0[2] += sa.z; a0[3] += sa.w;
    b0[0] += sb.x; b0[1] += sb.y; b0[2] += sb.z; b0[3] += sb.w;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][tid][e] = a0[e]; red[1][tid][e] = b0[e]; }
  __syncthreads();
  if (tid < c4n) {          // row lane 0 of each column group sums the others (fixed order)
    for (int l = 1; l < nrl; ++l)
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] += red[0][tid + l * c4n][e]; b0[e] += red[1][tid + l * c4n][e]; }
    double* p = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[4 * cg + e] = a0[e]; p[C + 4 * cg + e] = b0[e]; }
  }
}

// bn_reduce_kernel<MODE, false, T> with G float4 groups (4 G channels) per thread: one 16-byte access per row and thread for bf16 storage (G = 2).
// Same sums; the per-thread row sets differ from the G = 1 form (256 / (C / 4G) row lanes), so the results agree to rounding, not bitwise.
template <int MODE, typename T, int G>
__global__ __launch_bounds__(256) void bn_reduce_g_kernel(const T* __restrict__ x, const T* __restrict__ dy, const T* __restrict__ y,
                                                          const float* __restrict__ mean, const float* __restrict__ rstd,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta, T* __restrict__ gout,
                                                          double* __restrict__ partial, long long R, int C, int rpb, int relu) {
  __shared__ double red[2][256][4 * G];
  constexpr int W = 4 * G;
  const int cgn = C / W, nrl = 256 / cgn;
  const int tid = threadIdx.x, cg = tid % cgn, rl = tid / cgn;
  const long long r0 = (long long)blockIdx.x * rpb, r1 = min(R, r0 + rpb);
  double a0[W], b0[W];
#pragma unroll
  for (int e = 0; e < W; ++e) { a0[e] = 0.0; b0[e] = 0.0; }
  if (rl < nrl) {
    float4 mu[G], rs[G], ga[G], be[G], pv[G], sa[G], sb[G];
    const bool remask = MODE == 1 && relu && y == nullptr;       // ReLU mask rebuilt from x (no residual was added)
    if (MODE == 0) ldg<T, G>(x + W * cg, pv);                    // pivot = row 0 of the tensor (see bn_reduce_kernel)
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
      mu[g] = rs[g] = ga[g] = be[g] = sa[g] = sb[g] = z;
      if (MODE == 0) continue;
      pv[g] = z;
      mu[g] = ld4(mean + W * cg + 4 * g);
      rs[g] = ld4(rstd + W * cg + 4 * g);
      if (remask) { ga[g] = ld4(gamma + W * cg + 4 * g); be[g] = ld4(beta + W * cg + 4 * g); }
    }
    int cnt = 0;
    auto row = [&](long long r) {
      const long long off = r * C + W * cg;
      float4 xv[G], gv[G], yv[G];
      ldg<T, G>(x + off, xv);
      if (MODE == 1) {
        ldg<T, G>(dy + off, gv);
        if (relu && !remask) ldg<T, G>(y + off, yv);
      }
#pragma unroll
      for (int g = 0; g < G; ++g) {
        if (MODE == 0) {
          const float dx_ = xv[g].x - pv[g].x, dy_ = xv[g].y - pv[g].y, dz_ = xv[g].z - pv[g].z, dw_ = xv[g].w - pv[g].w;
          sa[g].x += dx_; sa[g].y += dy_; sa[g].z += dz_; sa[g].w += dw_;
          sb[g].x += dx_ * dx_; sb[g].y += dy_ * dy_; sb[g].z += dz_ * dz_; sb[g].w += dw_ * dw_;
        } else {
          if (remask) yv[g] = bn_affine(xv[g], mu[g], rs[g], ga[g], be[g]);
          if (relu) {
            gv[g].x = yv[g].x > 0.f ? gv[g].x : 0.f; gv[g].y = yv[g].y > 0.f ? gv[g].y : 0.f;
            gv[g].z = yv[g].z > 0.f ? gv[g].z : 0.f; gv[g].w = yv[g].w > 0.f ? gv[g].w : 0.f;
          }
          sa[g].x += gv[g].x; sa[g].y += gv[g].y; sa[g].z += gv[g].z; sa[g].w += gv[g].w;
          sb[g].x += gv[g].x * (xv[g].x - mu[g].x) * rs[g].x; sb[g].y += gv[g].y * (xv[g].y - mu[g].y) * rs[g].y;
          sb[g].z += gv[g].z * (xv[g].z - mu[g].z) * rs[g].z; sb[g].w += gv[g].w * (xv[g].w - mu[g].w) * rs[g].w;
        }
      }
      if (MODE == 1 && gout) stg<T, G>(gout + off, gv);
    };
    auto flush = [&]() {
#pragma unroll
      for (int g = 0; g < G; ++g) {
        a0[4 * g] += sa[g].x; a0[4 * g + 1] += sa[g].y; a0[4 * g + 2] += sa[g].z; a0[4 * g + 3] += sa[g].w;
        b0[4 * g] += sb[g].x; b0[4 * g + 1] += sb[g].y; b0[4 * g + 2] += sb[g].z; b0[4 * g + 3] += sb[g].w;
        sa[g] = make_float4(0.f, 0.f, 0.f, 0.f); sb[g] = sa[g];
      }
    };
    constexpr int U = 4;                     // rows in flight per thread
    long long r = r0 + rl;
    for (; r + (U - 1) * (long long)nrl < r1; r += (long long)U * nrl) {
#pragma unroll
      for (int u = 0; u < U; ++u) row(r + (long long)u * nrl);
      cnt += U;
      if (cnt >= 64) { flush(); cnt = 0; }       // fp32 partials over <= 64 rows, double above that
    }
    for (; r < r1; r += nrl) row(r);
    flush();
  }
#pragma unroll
  for (int e = 0; e < W; ++e) { red[0][tid][e] = a0[e]; red[1][tid][e] = b0[e]; }
  __syncthreads();
  if (tid < cgn) {          // row lane 0 of each column group sums the others (fixed order)
    for (int l = 1; l < nrl; ++l)
#pragma unroll
      for (int e = 0; e < W; ++e) { a0[e] += red[0][tid + l * cgn][e]; b0[e] += red[1][tid + l * cgn][e]; }
    double* p = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < W; ++e) { p[W * cg + e] = a0[e]; p[C + W * cg + e] = b0[e]; }
  }
}

// stage 2: a block takes 16 channels x 64 lanes (1024 threads); lane l sums block partials l, l + 64, ... (<= 16 of them: all loads
// independent and in flight together -- the round-2 form walked 64 partials per lane behind each other and took 11 us, 26 times per
// step), then the 64 lane sums of a channel are combined through LDS in a fixed order (8 x 8), all in double.
// MODE 0: mean, rstd (+ running statistics update).  MODE 1: dbeta, dgamma and the two means of the dx pass.
template <int MODE, typename TP = float>
__global__ __launch_bounds__(1024) void bn_finalize_kernel(const double* __restrict__ partial, int nblk, int C, long long R,
                                                           float* __restrict__ o0, float* __restrict__ o1,
                                                           float* __restrict__ run_mean, float* __restrict__ run_var,
                                                           float momentum, float eps, float* __restrict__ c12,
                                                           const TP* __restrict__ pivot) {
  __shared__ double red[2][64][16];
  const int cl = threadIdx.x & 15, lane = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  double a = 0.0, b = 0.0;
  if (c < C) {
    double av[16], bv[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int k = lane + 64 * j;
      const bool ok = k < nblk;
      const long long off = (long long)(ok ? k : 0) * 2 * C + c;
      av[j] = ok ? partial[off] : 0.0;
      bv[j] = ok ? partial[off + C] : 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; ++j) { a += av[j]; b += bv[j]; }
    for (int k = lane + 1024; k < nblk; k += 64) {          // (more than 1024 partials never happens with bn_rows_per_block)
      a += partial[(long long)k * 2 * C + c];
      b += partial[(long long)k * 2 * C + C + c];
    }
  }
  red[0][lane][cl] = a;
  red[1][lane][cl] = b;
  __syncthreads();
  if (lane < 8) {
    a = red[0][lane][cl]; b = red[1][lane][cl];
    for (int l = lane + 8; l < 64; l += 8) { a += red[0][l][cl]; b += red[1][l][cl]; }
  }
  __syncthreads();
  if (lane < 8) { red[0][lane][cl] = a; red[1][lane][cl] = b; }
  __syncthreads();
  if (lane != 0 || c >= C) return;
  for (int l = 1; l < 8; ++l) { a += red[0][l][cl]; b += red[1][l][cl]; }
  if (MODE == 0) {
    const double md = a / (double)R;                 // mean of (x - pivot)
    double var = b / (double)R - md * md;
    if (var < 0.0) var = 0.0;
    const double m = md + (double)ld1<TP>(pivot + c);
    o0[c] = (float)m;
    o1[c] = (float)(1.0 / sqrt(var + (double)eps));
    if (run_mean) {
      const double unb = R > 1 ? var * (double)R / (double)(R - 1) : var;
      run_mean[c] = (float)((1.0 - momentum) * run_mean[c] + momentum * m);
      run_var[c] = (float)((1.0 - momentum) * run_var[c] + momentum * unb);
    }
  } else {
    o0[c] = (float)a;            // dbeta
    o1[c] = (float)b;            // dgamma
    c12[c] = (float)(a / (double)R);
    c12[C + c] = (float)(b / (double)R);
  }
}

// y = relu?((x - mean) * rstd * gamma + beta (+ residual))
template <typename T, int G>
__global__ __launch_bounds__(256) void bn_apply_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                           const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const T* __restrict__ res,
                                                           T* __restrict__ y, long long ng, int cgn, int relu) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ng; i += (long long)gridDim.x * 256) {
    const int c = 4 * G * (int)(i % cgn);
    float4 xv[G], r[G], v[G];
    ldg<T, G>(x + 4 * G * i, xv);
    if (res) ldg<T, G>(res + 4 * G * i, r);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const float4 mu = ld4(mean + c + 4 * g), rs = ld4(rstd + c + 4 * g), ga = ld4(gamma + c + 4 * g), be = ld4(beta + c + 4 * g);
      v[g] = bn_affine(xv[g], mu, rs, ga, be);
      if (res) { v[g].x += r[g].x; v[g].y += r[g].y; v[g].z += r[g].z; v[g].w += r[g].w; }
      if (relu) { v[g].x = fmaxf(v[g].x, 0.f); v[g].y = fmaxf(v[g].y, 0.f); v[g].z = fmaxf(v[g].z, 0.f); v[g].w = fmaxf(v[g].w, 0.f); }
    }
    stg<T, G>(y + 4 * G * i, v);
  }
}

// dx = gamma * rstd * (g - c1 - xhat * c2)   (training);   dx = gamma * rstd * g   (eval: c12 == nullptr)
// g is read from `g` when given (it was stored by the reduce pass for the residual branch), else recomputed from dy, y
template <bool POOL, typename T>
__global__ __launch_bounds__(256) void bn_apply_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                           const T* __restrict__ gin, const T* __restrict__ x,
                                                           const float* __restrict__ mean, const float* __restrict__ rstd,
                                                           const float* __restrict__ gamma, const float* __restrict__ beta,
                                                           const float* __restrict__ c12, T* __restrict__ dx, long long n4,
                                                           int c4n, int relu, PoolSrc ps = PoolSrc{}) {
  const int C = 4 * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const int c = 4 * (int)(i % c4n);
    const float4 rs = ld4(rstd + c), ga = ld4(gamma + c), mu = ld4(mean + c);
    const float4 xv = ldv<T>(x + 4 * i);
    float4 g;
    if (gin) {
      g = ldv<T>(gin + 4 * i);
    } else {
      g = POOL ? pool_gather<T>(ps, (unsigned)i / (unsigned)c4n, C, c) : ldv<T>(dy + 4 * i);
      if (relu) {
        const float4 yv = y ? ldv<T>(y + 4 * i) : bn_affine(xv, mu, rs, ga, ld4(beta + c));
        g.x = yv.x > 0.f ? g.x : 0.f; g.y = yv.y > 0.f ? g.y : 0.f; g.z = yv.z > 0.f ? g.z : 0.f; g.w = yv.w > 0.f ? g.w : 0.f;
      }
    }
    float4 v;
    if (c12) {
      const float4 c1 = ld4(c12 + c), c2 = ld4(c12 + C + c);
      v.x = (ga.x * rs.x) * (g.x - c1.x - (xv.x - mu.x) * rs.x * c2.x);
      v.y = (ga.y * rs.y) * (g.y - c1.y - (xv.y - mu.y) * rs.y * c2.y);
      v.z = (ga.z * rs.z) * (g.z - c1.z - (xv.z - mu.z) * rs.z * c2.z);
      v.w = (ga.w * rs.w) * (g.w - c1.w - (xv.w - mu.w) * rs.w * c2.w);
    } else {
      v.x = ga.x * rs.x * g.x; v.y = ga.y * rs.y * g.y; v.z = ga.z * rs.z * g.z; v.w = ga.w * rs.w * g.w;
    }
    stv<T>(dx + 4 * i, v);
  }
}

// bn_apply_bwd_kernel<false, T> with G groups per thread (same arithmetic per element)
template <typename T, int G>
__global__ __launch_bounds__(256) void bn_apply_bwd_g_kernel(const T* __restrict__ dy, const T* __restrict__ y, const T* __restrict__ gin,
                                                             const T* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, const float* __restrict__ c12,
                                                             T* __restrict__ dx, long long ng, int cgn, int relu) {
  const int C = 4 * G * cgn;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < ng; i += (long long)gridDim.x * 256) {
    const int c0 = 4 * G * (int)(i % cgn);
    float4 xv[G], gv[G], yv[G], v[G];
    ldg<T, G>(x + 4 * G * i, xv);
    if (gin) {
      ldg<T, G>(gin + 4 * G * i, gv);
    } else {
      ldg<T, G>(dy + 4 * G * i, gv);
      if (relu && y) ldg<T, G>(y + 4 * G * i, yv);
    }
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int c = c0 + 4 * g;
      const float4 rs = ld4(rstd + c), ga = ld4(gamma + c), mu = ld4(mean + c);
      if (!gin && relu) {
        if (!y) yv[g] = bn_affine(xv[g], mu, rs, ga, ld4(beta + c));
        gv[g].x = yv[g].x > 0.f ? gv[g].x : 0.f; gv[g].y = yv[g].y > 0.f ? gv[g].y : 0.f;
        gv[g].z = yv[g].z > 0.f ? gv[g].z : 0.f; gv[g].w = yv[g].w > 0.f ? gv[g].w : 0.f;
      }
      if (c12) {
        const float4 c1 = ld4(c12 + c), c2 = ld4(c12 + C + c);
        v[g].x = (ga.x * rs.x) * (gv[g].x - c1.x - (xv[g].x - mu.x) * rs.x * c2.x);
        v[g].y = (ga.y * rs.y) * (gv[g].y - c1.y - (xv[g].y - mu.y) * rs.y * c2.y);
        v[g].z = (ga.z * rs.z) * (gv[g].z - c1.z - (xv[g].z - mu.z) * rs.z * c2.z);
        v[g].w = (ga.w * rs.w) * (gv[g].w - c1.w - (xv[g].w - mu.w) * rs.w * c2.w);
      } else {
        v[g].x = ga.x * rs.x * gv[g].x; v[g].y = ga.y * rs.y * gv[g].y; v[g].z = ga.z * rs.z * gv[g].z; v[g].w = ga.w * rs.w * gv[g].w;
      }
    }
    stg<T, G>(dx + 4 * G * i, v);
  }
}

int bn_check(long long R, int C) {
  if (R <= 0 || C <= 0 || (C & 3) || C > BN_MAXC) return RP_EBADSHAPE;
  return RP_OK;
}
int apply_grid(long long n4) {
  long long b = (n4 + 255) / 256;
  return (int)(b > 8192 ? 8192 : b);
}

}  // namespace

extern "C" int rp_bn_partial_blocks(long long R) {
  const int rpb = bn_rows_per_block(R);
  return (int)((R + rpb - 1) / rpb);
}

template <typename T>
static int bn_stats_t(const T* x, long long R, int C, double* partial, float* mean, float* rstd, float* running_mean,
                      float* running_var, float momentum, float eps, hipStream_t st) {
  const int rpb = bn_rows_per_block(R), nblk = (int)((R + rpb - 1) / rpb);
  bool wide = false;
  if constexpr (sizeof(T) == 2) wide = bn_wide(1, C);
  if constexpr (sizeof(T) == 2) {
    if (wide) hipLaunchKernelGGL((bn_reduce_g_kernel<0, T, 2>), dim3(nblk), dim3(256), 0, st, x, (const T*)nullptr, (const T*)nullptr, nullptr, nullptr,
                                 nullptr, nullptr, (T*)nullptr, partial, R, C, rpb, 0);
  }
  if (!wide) hipLaunchKernelGGL((bn_reduce_kernel<0, false, T>), dim3(nblk), dim3(256), 0, st, x, (const T*)nullptr, (const T*)nullptr, nullptr, nullptr,
                                nullptr, nullptr, (T*)nullptr, partial, R, C, rpb, 0, PoolSrc{});
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL((bn_finalize_kernel<0, T>), dim3((C + 15) / 16), dim3(1024), 0, st, (const double*)partial, nblk, C, R, mean,
                     rstd, running_mean, running_var, momentum, eps, nullptr, x);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_bn_stats(const void* x, long long R, int C, double* partial, float* mean, float* rstd, float* running_mean,
                           float* running_var, float momentum, float eps, int bf16, void* stream) {
  if (int e = bn_check(R, C)) return e;
  if (bf16) return bn_stats_t((const bf16s*)x, R, C, partial, mean, rstd, running_mean, running_var, momentum, eps, (hipStream_t)stream);
  return bn_stats_t((const float*)x, R, C, partial, mean, rstd, running_mean, running_var, momentum, eps, (hipStream_t)stream);
}

// statistics from per-block partial sums produced elsewhere (rp_conv_stem_fwd's epilogue): partial [nblk][2][C] doubles =
// sums of (x - pivot[c]) and (x - pivot[c])^2 over disjoint row sets covering all R rows
extern "C" int rp_bn_stats_from_partials(const double* partial, int nblk, long long R, int C, const float* pivot, float* mean,
                                         float* rstd, float* running_mean, float* running_var, float momentum, float eps,
                                         void* stream) {
  if (int e = bn_check(R, C)) return e;
  if (nblk <= 0 || !partial || !pivot) return RP_EBADSHAPE;
  hipLaunchKernelGGL((bn_finalize_kernel<0, float>), dim3((C + 15) / 16), dim3(1024), 0, (hipStream_t)stream, partial, nblk, C, R, mean, rstd,
                     running_mean, running_var, momentum, eps, nullptr, pivot);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_bn_apply_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                               const void* residual, void* y, long long R, int C, int relu, int bf16, void* stream) {
  if (int e = bn_check(R, C)) return e;
  const long long n4 = R * C / 4;
  if (bn_wide(bf16, C)) hipLaunchKernelGGL((bn_apply_fwd_kernel<bf16s, 2>), dim3(apply_grid(n4 / 2)), dim3(256), 0, (hipStream_t)stream, (const bf16s*)x,
                                           mean, rstd, gamma, beta, (const bf16s*)residual, (bf16s*)y, n4 / 2, C / 8, relu);
  else if (bf16) hipLaunchKernelGGL((bn_apply_fwd_kernel<bf16s, 1>), dim3(apply_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const bf16s*)x, mean,
                                    rstd, gamma, beta, (const bf16s*)residual, (bf16s*)y, n4, C / 4, relu);
  else hipLaunchKernelGGL((bn_apply_fwd_kernel<float, 1>), dim3(apply_grid(n4)), dim3(256), 0, (hipStream_t)stream, (const float*)x, mean, rstd, gamma,
                          beta, (const float*)residual, (float*)y, n4, C / 4, relu);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

template <typename T>
static int bn_bwd_t(const T* dy, const T* y, const T* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    T* dx, T* dres, float* dgamma, float* dbeta, double* partial, float* c12, long long R, int C, int relu, int training,
                    hipStream_t st) {
  const int rpb = bn_rows_per_block(R), nblk = (int)((R + rpb - 1) / rpb);
  bool wide = false;
  if constexpr (sizeof(T) == 2) wide = bn_wide(1, C);
  if constexpr (sizeof(T) == 2) {
    if (wide) hipLaunchKernelGGL((bn_reduce_g_kernel<1, T, 2>), dim3(nblk), dim3(256), 0, st, x, dy, y, mean, rstd, gamma, beta, dres, partial, R, C,
                                 rpb, relu);
  }
  if (!wide) hipLaunchKernelGGL((bn_reduce_kernel<1, false, T>), dim3(nblk), dim3(256), 0, st, x, dy, y, mean, rstd, gamma, beta, dres, partial, R, C, rpb,
                                relu, PoolSrc{});
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL((bn_finalize_kernel<1, float>), dim3((C + 15) / 16), dim3(1024), 0, st, (const double*)partial, nblk, C, R, dbeta,
                     dgamma, nullptr, nullptr, 0.f, 0.f, c12, nullptr);
  RP_CHECK_LAUNCH();
  const long long n4 = R * C / 4;
  if constexpr (sizeof(T) == 2) {
    if (bn_wide(1, C)) {
      hipLaunchKernelGGL((bn_apply_bwd_g_kernel<T, 2>), dim3(apply_grid(n4 / 2)), dim3(256), 0, st, dy, y, (const T*)dres, x, mean, rstd, gamma, beta,
                         training ? (const float*)c12 : nullptr, dx, n4 / 2, C / 8, relu);
      RP_CHECK_LAUNCH();
      return RP_OK;
    }
  }
  hipLaunchKernelGGL((bn_apply_bwd_kernel<false, T>), dim3(apply_grid(n4)), dim3(256), 0, st, dy, y, (const T*)dres, x, mean, rstd, gamma,
                     beta, training ? (const float*)c12 : nullptr, dx, n4, C / 4, relu, PoolSrc{});
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_bn_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd,
                         const float* gamma, const float* beta, void* dx, void* dres, float* dgamma, float* dbeta,
                         double* partial, float* c12, long long R, int C, int relu, int training, int bf16, void* stream) {
  if (int e = bn_check(R, C)) return e;
  if (relu && !y && !beta) return RP_EBADSHAPE;
  if (bf16) return bn_bwd_t((const bf16s*)dy, (const bf16s*)y, (const bf16s*)x, mean, rstd, gamma, beta, (bf16s*)dx, (bf16s*)dres, dgamma, dbeta,
                            partial, c12, R, C, relu, training, (hipStream_t)stream);
  return bn_bwd_t((const float*)dy, (const float*)y, (const float*)x, mean, rstd, gamma, beta, (float*)dx, (float*)dres, dgamma, dbeta, partial,
                  c12, R, C, relu, training, (hipStream_t)stream);
}

extern "C" int rp_bn_bwd_from_partials(const float* g, const float* x, const float* mean, const float* rstd, const float* gamma,
                                       const double* partial, int nblk, float* dx, float* dgamma, float* dbeta, float* c12, long long R, int C,
                                       void* stream) {
  if (int e = bn_check(R, C)) return e;
  if (!g || !x || !partial || nblk <= 0 || !dx || !dgamma || !dbeta || !c12) return RP_EBADSHAPE;
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((bn_finalize_kernel<1, float>), dim3((C + 15) / 16), dim3(1024), 0, st, partial, nblk, C, R, dbeta, dgamma, nullptr, nullptr,
                     0.f, 0.f, c12, nullptr);
  RP_CHECK_LAUNCH();
  const long long n4 = R * C / 4;
  hipLaunchKernelGGL((bn_apply_bwd_kernel<false, float>), dim3(apply_grid(n4)), dim3(256), 0, st, g, (const float*)nullptr, g, x, mean, rstd, gamma,
                     (const float*)nullptr, (const float*)c12, dx, n4, C / 4, 0, PoolSrc{});
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// ---- 3x3 / stride 2 / pad 1 max-pool of the stem (torchvision resnet.maxpool), channels-last --------------------------
// PyTorch's NHWC max-pool kernels take 139 us forward and 336 us backward on the [128,64,112,112] stem activation; both are
// plain HBM streams (103 MB in, 26 MB out and back).  Forward stores the window position (0..8) of the FIRST maximum in
// scan order (strict >, like torch's kernel) so that ties -- frequent after a ReLU -- route the gradient identically; backward
// is a gather (every input pixel looks at the <= 4 windows that contain it), no atomics.
namespace {

template <typename T>
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int N, int H, int W, int C, int OH,
                                                          int OW) {
  const int c4n = C >> 2;
  const long long total = (long long)N * OH * OW * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long long p = i / c4n;
    const int ow = (int)(p % OW);
    p /= OW;
    const int oh = (int)(p % OH), n = (int)(p / OH);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    int kx = 0, ky = 0, kz = 0, kw_ = 0;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 2 * oh - 1 + kh;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = 2 * ow - 1 + kw;
        if (iw < 0 || iw >= W) continue;
        const float4 v = ldv<T>(x + (((long long)n * H + ih) * W + iw) * C + 4 * c4);
        const int k = kh * 3 + kw;
        if (v.x > m.x) { m.x = v.x; kx = k; }
        if (v.y > m.y) { m.y = v.y; ky = k; }
        if (v.z > m.z) { m.z = v.z; kz = k; }
        if (v.w > m.w) { m.w = v.w; kw_ = k; }
      }
    }
    stv<T>(y + 4 * i, m);
    *reinterpret_cast<uchar4*>(idx + 4 * i) = make_uchar4((unsigned char)kx, (unsigned char)ky, (unsigned char)kz, (unsigned char)kw_);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const T* __restrict__ dy, const unsigned char* __restrict__ idx,
                                                          T* __restrict__ dx, int N, int H, int W, int C, int OH, int OW) {
  const int c4n = C >> 2;
  const long long total = (long long)N * H * W * c4n;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c4 = (int)(i % c4n);
    long long p = i / c4n;
    const int iw = (int)(p % W);
    p /= W;
    const int ih = (int)(p % H), n = (int)(p / H);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
    // windows (oh, ow) with 2*oh - 1 <= ih <= 2*oh + 1
    const int oh0 = ih >> 1, ow0 = iw >> 1;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int oh = oh0 + a;
      const int kh = ih - (2 * oh - 1);
      if (oh >= OH || kh < 0 || kh > 2) continue;
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int ow = ow0 + b;
        const int kw = iw - (2 * ow - 1);
        if (ow >= OW || kw < 0 || kw > 2) continue;
        const long long o = ((((long long)n * OH + oh) * OW + ow) * c4n + c4) * 4;
        const uchar4 k4 = *reinterpret_cast<const uchar4*>(idx + o);
        const float4 d = ldv<T>(dy + o);
        const int k = kh * 3 + kw;
        if (k4.x == k) g.x += d.x;
        if (k4.y == k) g.y += d.y;
        if (k4.z == k) g.z += d.z;
        if (k4.w == k) g.w += d.w;
      }
    }
    stv<T>(dx + 4 * i, g);
  }
}

}  // namespace

extern "C" int rp_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int bf16, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return RP_EBADSHAPE;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 4);
  if (bf16) hipLaunchKernelGGL(maxpool_fwd_kernel<bf16s>, dim3(apply_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16s*)x, (bf16s*)y, idx,
                               N, H, W, C, OH, OW);
  else hipLaunchKernelGGL(maxpool_fwd_kernel<float>, dim3(apply_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, (float*)y, idx, N, H,
                          W, C, OH, OW);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

extern "C" int rp_maxpool3x3s2_bwd(const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C, int bf16, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || C <= 0 || (C & 3)) return RP_EBADSHAPE;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * H * W * (C / 4);
  if (bf16) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16s>, dim3(apply_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16s*)dy, idx, (bf16s*)dx,
                               N, H, W, C, OH, OW);
  else hipLaunchKernelGGL(maxpool_bwd_kernel<float>, dim3(apply_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)dy, idx, (float*)dx, N,
                          H, W, C, OH, OW);
  RP_CHECK_LAUNCH();
  return RP_OK;
}


// ---- stem: BatchNorm + ReLU + 3x3/2 max-pool as one pass (forward) and without the pool-backward tensor (backward) -------------
namespace {

// pooled[n,oh,ow,c] = max over the window of relu(bn(x)); idx = window position of the FIRST maximum (strict >), i.e. exactly
// maxpool_fwd_kernel applied to bn_apply_fwd_kernel's output, which is never written
template <typename T, int G>
__global__ __launch_bounds__(256) void bn_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, T* __restrict__ y,
                                                          unsigned char* __restrict__ idx, int N, int H, int W, int C, int OH, int OW) {
  constexpr int CW = 4 * G;                                       // channels per thread (G = 2: one 16-byte access of bf16 storage)
  const int cgn = C / CW;
  const long long total = (long long)N * OH * OW * cgn;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = CW * (int)(i % cgn);
    long long p = i / cgn;
    const int ow = (int)(p % OW);
    p /= OW;
    const int oh = (int)(p % OH), n = (int)(p / OH);
    float4 mu[G], rs[G], ga[G], be[G], m[G];
    int kk[G][4];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      mu[g] = ld4(mean + c + 4 * g); rs[g] = ld4(rstd + c + 4 * g); ga[g] = ld4(gamma + c + 4 * g); be[g] = ld4(beta + c + 4 * g);
      m[g] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
      kk[g][0] = kk[g][1] = kk[g][2] = kk[g][3] = 0;
    }
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = 2 * oh - 1 + kh;
      if (ih < 0 || ih >= H) continue;
#pragma unroll
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = 2 * ow - 1 + kw;
        if (iw < 0 || iw >= W) continue;
        float4 xv[G];
        ldg<T, G>(x + (((long long)n * H + ih) * W + iw) * C + c, xv);
        const int k = kh * 3 + kw;
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float4 v = bn_affine(xv[g], mu[g], rs[g], ga[g], be[g]);
          v.x = fmaxf(v.x, 0.f); v.y = fmaxf(v.y, 0.f); v.z = fmaxf(v.z, 0.f); v.w = fmaxf(v.w, 0.f);
          if (v.x > m[g].x) { m[g].x = v.x; kk[g][0] = k; }
          if (v.y > m[g].y) { m[g].y = v.y; kk[g][1] = k; }
          if (v.z > m[g].z) { m[g].z = v.z; kk[g][2] = k; }
          if (v.w > m[g].w) { m[g].w = v.w; kk[g][3] = k; }
        }
      }
    }
    stg<T, G>(y + CW * i, m);
#pragma unroll
    for (int g = 0; g < G; ++g)
      *reinterpret_cast<uchar4*>(idx + CW * i + 4 * g) = make_uchar4((unsigned char)kk[g][0], (unsigned char)kk[g][1], (unsigned char)kk[g][2],
                                                                     (unsigned char)kk[g][3]);
  }
}

// Backward stage 1 of the fused stem, window-major: every pooled element routes its gradient to ONE input pixel (its stored window
// position), so the column sums  a = sum g,  b = sum g * xhat  over the [N,H,W,C] pixels are sums over the [N,OH,OW,C] windows of
// dp * [relu active at the arg-max pixel] (* xhat there): a quarter of the iterations of the pixel-major pass and no window search.
// (Same sums as bn_reduce_kernel<1>, different order: agrees to fp32 rounding, not bitwise.)
template <typename T>
__global__ __launch_bounds__(256) void bn_pool_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dp,
                                                             const unsigned char* __restrict__ idx, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                             const float* __restrict__ beta, double* __restrict__ partial, long long RO,
                                                             int C, int rpb, int H, int W, int OH, int OW) {
  __shared__ double red[2][256][4];
  const int c4n = C >> 2, nrl = 256 / c4n;
  const int tid = threadIdx.x, cg = tid % c4n, rl = tid / c4n;
  const long long r0 = (long long)blockIdx.x * rpb, r1 = min(RO, r0 + rpb);
  double a0[4] = {0, 0, 0, 0}, b0[4] = {0, 0, 0, 0};
  if (rl < nrl) {
    const float4 mu = ld4(mean + 4 * cg), rs = ld4(rstd + 4 * cg), ga = ld4(gamma + 4 * cg), be = ld4(beta + 4 * cg);
    float4 sa = make_float4(0.f, 0.f, 0.f, 0.f), sb = sa;
    int cnt = 0;
    for (long long r = r0 + rl; r < r1; r += nrl) {
      const unsigned ur = (unsigned)r;
      const unsigned t = ur / (unsigned)OW;
      const int ow = (int)(ur - t * (unsigned)OW);
      const unsigned nn = t / (unsigned)OH;
      const int oh = (int)(t - nn * (unsigned)OH);
      const unsigned o = ur * (unsigned)C + 4 * cg;
      const float4 d = ldv<T>(dp + o);
      const uchar4 k4 = *reinterpret_cast<const uchar4*>(idx + o);
      const T* xb = x + (((long long)nn * H + (2 * oh - 1)) * W + (2 * ow - 1)) * C + 4 * cg;         // window origin (may be off-image;
      auto at = [&](int k, int e) { return ld1<T>(xb + ((k / 3) * W + (k % 3)) * C + e); };             //  the arg-max never is)
      const float xv[4] = {at(k4.x, 0), at(k4.y, 1), at(k4.z, 2), at(k4.w, 3)};
      const float dv[4] = {d.x, d.y, d.z, d.w};
      const float muv[4] = {mu.x, mu.y, mu.z, mu.w}, rsv[4] = {rs.x, rs.y, rs.z, rs.w}, gav[4] = {ga.x, ga.y, ga.z, ga.w},
                  bev[4] = {be.x, be.y, be.z, be.w};
      float* sav = &sa.x;
      float* sbv = &sb.x;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float yv = __builtin_fmaf(xv[e] - muv[e], rsv[e] * gav[e], bev[e]);      // bn_affine: the forward's value, bit for bit
        const float g = yv > 0.f ? dv[e] : 0.f;
        sav[e] += g;
        sbv[e] += g * (xv[e] - muv[e]) * rsv[e];
      }
      if (++cnt >= 64) {
        a0[0] += sa.x; a0[1] += sa.y; a0[2] += sa.z; a0[3] += sa.w;
        b0[0] += sb.x; b0[1] += sb.y; b0[2] += sb.z; b0[3] += sb.w;
        sa = make_float4(0.f, 0.f, 0.f, 0.f); sb = sa; cnt = 0;
      }
    }
    a0[0] += sa.x; a0[1] += sa.y; a0[2] += sa.z; a0[3] += sa.w;
    b0[0] += sb.x; b0[1] += sb.y; b0[2] += sb.z; b0[3] += sb.w;
  }
#pragma unroll
  for (int e = 0; e < 4; ++e) { red[0][tid][e] = a0[e]; red[1][tid][e] = b0[e]; }
  __syncthreads();
  if (tid < c4n) {
    for (int l = 1; l < nrl; ++l)
#pragma unroll
      for (int e = 0; e < 4; ++e) { a0[e] += red[0][tid + l * c4n][e]; b0[e] += red[1][tid + l * c4n][e]; }
    double* p = partial + (long long)blockIdx.x * 2 * C;
#pragma unroll
    for (int e = 0; e < 4; ++e) { p[4 * cg + e] = a0[e]; p[C + 4 * cg + e] = b0[e]; }
  }
}

}  // namespace

extern "C" int rp_bn_relu_pool_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                                   void* y, unsigned char* idx, int N, int H, int W, int C, int bf16, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0) return RP_EBADSHAPE;
  if (int e = bn_check((long long)N * H * W, C)) return e;
  const int OH = (H - 1) / 2 + 1, OW = (W - 1) / 2 + 1;
  const long long total = (long long)N * OH * OW * (C / 4);
  if (bn_wide(bf16, C)) hipLaunchKernelGGL((bn_pool_fwd_kernel<bf16s, 2>), dim3(apply_grid(total / 2)), dim3(256), 0, (hipStream_t)stream, (const bf16s*)x,
                                           mean, rstd, gamma, beta, (bf16s*)y, idx, N, H, W, C, OH, OW);
  else if (bf16) hipLaunchKernelGGL((bn_pool_fwd_kernel<bf16s, 1>), dim3(apply_grid(total)), dim3(256), 0, (hipStream_t)stream, (const bf16s*)x, mean, rstd,
                                    gamma, beta, (bf16s*)y, idx, N, H, W, C, OH, OW);
  else hipLaunchKernelGGL((bn_pool_fwd_kernel<float, 1>), dim3(apply_grid(total)), dim3(256), 0, (hipStream_t)stream, (const float*)x, mean, rstd, gamma,
                          beta, (float*)y, idx, N, H, W, C, OH, OW);
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// dx of the fused stem (BatchNorm + ReLU + 3x3/2 max-pool), PATCH-major: a thread takes a 2 x 2 input patch (rows 2a, 2a + 1, columns 2b,
// 2b + 1) x 4 channels.  The four pixels of a patch look at the same four pooling windows (a .. a + 1, b .. b + 1), so a window's position
// bytes and gradient are read ONCE per patch instead of once per pixel that touches it: bn_apply_bwd_kernel<true> issued 80 bytes of L2
// reads per 16 bytes of dx (246 us at 128 images, 3.9 TB/s of HBM traffic but L2-bound), this form 20.  Same sums in the same order
// (windows (a, b), (a, b + 1), (a + 1, b), (a + 1, b + 1)): bit-identical to the pixel-major kernel.
template <typename T, int G>
__global__ __launch_bounds__(256) void bn_pool_apply_bwd_kernel(const T* __restrict__ dp, const unsigned char* __restrict__ idx,
                                                                const T* __restrict__ x, const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, const float* __restrict__ c12,
                                                                T* __restrict__ dx, int N, int H, int W, int C, int OH, int OW) {
  constexpr int CW = 4 * G;                                       // channels per thread (G = 2: one 16-byte access of bf16 storage)
  const int cgn = C / CW, PH = (H + 1) >> 1, PW = (W + 1) >> 1;
  const long long total = (long long)N * PH * PW * cgn;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
    const int c = CW * (int)(i % cgn);
    long long q = i / cgn;
    const int b = (int)(q % PW);
    q /= PW;
    const int a = (int)(q % PH), n = (int)(q / PH);
    float4 rs[G], ga[G], mu[G], be[G], c1[G], c2[G];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      rs[g] = ld4(rstd + c + 4 * g); ga[g] = ld4(gamma + c + 4 * g); mu[g] = ld4(mean + c + 4 * g); be[g] = ld4(beta + c + 4 * g);
      c1[g] = c2[g] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c12) { c1[g] = ld4(c12 + c + 4 * g); c2[g] = ld4(c12 + C + c + 4 * g); }
    }
    // the four windows of the patch: position bytes and gradients (zero where the window does not exist)
    uchar4 k4[2][2][G];
    float4 d4[2][2][G];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int v = 0; v < 2; ++v) {
        const int oh = a + u, ow = b + v;
#pragma unroll
        for (int g = 0; g < G; ++g) { k4[u][v][g] = make_uchar4(255, 255, 255, 255); d4[u][v][g] = make_float4(0.f, 0.f, 0.f, 0.f); }
        if (oh < OH && ow < OW) {
          const unsigned o = (((unsigned)n * OH + oh) * OW + ow) * C + c;
          if constexpr (G == 1) {
            k4[u][v][0] = *reinterpret_cast<const uchar4*>(idx + o);
          } else {
            const uint2 kk = *reinterpret_cast<const uint2*>(idx + o);
            k4[u][v][0] = __builtin_bit_cast(uchar4, kk.x);
            k4[u][v][1] = __builtin_bit_cast(uchar4, kk.y);
          }
          ldg<T, G>(dp + o, d4[u][v]);
        }
      }
#pragma unroll
    for (int pr = 0; pr < 2; ++pr)
#pragma unroll
      for (int pc = 0; pc < 2; ++pc) {
        const int ih = 2 * a + pr, iw = 2 * b + pc;
        if (ih >= H || iw >= W) continue;
        const long long off = (((long long)n * H + ih) * W + iw) * C + c;
        float4 xv[G], o[G];
        ldg<T, G>(x + off, xv);
#pragma unroll
        for (int g = 0; g < G; ++g) {
          float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
          // windows in maxpool_bwd_kernel's order; window (a + u, b + v) holds pixel (ih, iw) at position (ih - 2 (a + u) + 1, iw - 2 (b + v) + 1)
#pragma unroll
          for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v) {
              const int kh = pr - 2 * u + 1, kw = pc - 2 * v + 1;
              if (kh < 0 || kh > 2 || kw < 0 || kw > 2) continue;       // (compile-time: pr, pc, u, v are unrolled)
              const int k = kh * 3 + kw;
              if (k4[u][v][g].x == k) gr.x += d4[u][v][g].x;
              if (k4[u][v][g].y == k) gr.y += d4[u][v][g].y;
              if (k4[u][v][g].z == k) gr.z += d4[u][v][g].z;
              if (k4[u][v][g].w == k) gr.w += d4[u][v][g].w;
            }
          const float4 yv = bn_affine(xv[g], mu[g], rs[g], ga[g], be[g]);
          gr.x = yv.x > 0.f ? gr.x : 0.f; gr.y = yv.y > 0.f ? gr.y : 0.f; gr.z = yv.z > 0.f ? gr.z : 0.f; gr.w = yv.w > 0.f ? gr.w : 0.f;
          if (c12) {
            o[g].x = (ga[g].x * rs[g].x) * (gr.x - c1[g].x - (xv[g].x - mu[g].x) * rs[g].x * c2[g].x);
            o[g].y = (ga[g].y * rs[g].y) * (gr.y - c1[g].y - (xv[g].y - mu[g].y) * rs[g].y * c2[g].y);
            o[g].z = (ga[g].z * rs[g].z) * (gr.z - c1[g].z - (xv[g].z - mu[g].z) * rs[g].z * c2[g].z);
            o[g].w = (ga[g].w * rs[g].w) * (gr.w - c1[g].w - (xv[g].w - mu[g].w) * rs[g].w * c2[g].w);
          } else {
            o[g].x = ga[g].x * rs[g].x * gr.x; o[g].y = ga[g].y * rs[g].y * gr.y; o[g].z = ga[g].z * rs[g].z * gr.z; o[g].w = ga[g].w * rs[g].w * gr.w;
          }
        }
        stg<T, G>(dx + off, o);
      }
  }
}

template <typename T>
static int bn_relu_pool_bwd_t(const T* dp, const unsigned char* idx, const T* x, const float* mean, const float* rstd, const float* gamma,
                              const float* beta, T* dx, float* dgamma, float* dbeta, double* partial, float* c12, int N, int H, int W,
                              int C, int training, hipStream_t st) {
  const long long R = (long long)N * H * W;
  const PoolSrc ps{dp, idx, H, W, (H - 1) / 2 + 1, (W - 1) / 2 + 1};
  int nblk;
  static const bool pixel_major = getenv("RP_BN_POOL_PIXEL_MAJOR") != nullptr;     // A/B aid: the bit-identical pixel-major stage 1
  if (pixel_major) {
    const int rpb = bn_rows_per_block(R);
    nblk = (int)((R + rpb - 1) / rpb);
    hipLaunchKernelGGL((bn_reduce_kernel<1, true, T>), dim3(nblk), dim3(256), 0, st, x, (const T*)nullptr, (const T*)nullptr, mean, rstd, gamma, beta,
                       (T*)nullptr, partial, R, C, rpb, 1, ps);
  } else {
    const long long RO = (long long)N * ps.OH * ps.OW;
    const int rpb = bn_rows_per_block(RO);
    nblk = (int)((RO + rpb - 1) / rpb);                                            // <= the buffer sized for R rows
    hipLaunchKernelGGL(bn_pool_reduce_kernel<T>, dim3(nblk), dim3(256), 0, st, x, dp, idx, mean, rstd, gamma, beta, partial, RO, C, rpb,
                       H, W, ps.OH, ps.OW);
  }
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL((bn_finalize_kernel<1, float>), dim3((C + 15) / 16), dim3(1024), 0, st, (const double*)partial, nblk, C, R, dbeta,
                     dgamma, nullptr, nullptr, 0.f, 0.f, c12, nullptr);
  RP_CHECK_LAUNCH();
  static const bool pixel_apply = getenv("RP_BN_POOL_PIXEL_APPLY") != nullptr;     // A/B aid: the pixel-major third pass (bit-identical)
  if (pixel_apply) {
    const long long n4 = R * C / 4;
    hipLaunchKernelGGL((bn_apply_bwd_kernel<true, T>), dim3(apply_grid(n4)), dim3(256), 0, st, (const T*)nullptr, (const T*)nullptr, (const T*)nullptr,
                       x, mean, rstd, gamma, beta, training ? (const float*)c12 : nullptr, dx, n4, C / 4, 1, ps);
  } else {
    const long long np = (long long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C / 4);
    bool wide = false;
    if constexpr (sizeof(T) == 2) wide = bn_wide(1, C);
    if constexpr (sizeof(T) == 2) {
      if (wide) hipLaunchKernelGGL((bn_pool_apply_bwd_kernel<T, 2>), dim3(apply_grid(np / 2)), dim3(256), 0, st, dp, idx, x, mean, rstd, gamma, beta,
                                   training ? (const float*)c12 : nullptr, dx, N, H, W, C, ps.OH, ps.OW);
    }
    if (!wide) hipLaunchKernelGGL((bn_pool_apply_bwd_kernel<T, 1>), dim3(apply_grid(np)), dim3(256), 0, st, dp, idx, x, mean, rstd, gamma, beta,
                                  training ? (const float*)c12 : nullptr, dx, N, H, W, C, ps.OH, ps.OW);
  }
  RP_CHECK_LAUNCH();
  return RP_OK;
}

// backward of the same chain: dp = gradient of the pooled output [N,OH,OW,C]; dx = gradient of the BatchNorm INPUT [N,H,W,C]
extern "C" int rp_bn_relu_pool_bwd(const void* dp, const unsigned char* idx, const void* x, const float* mean, const float* rstd,
                                   const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, double* partial,
                                   float* c12, int N, int H, int W, int C, int training, int bf16, void* stream) {
  if (N <= 0 || H <= 0 || W <= 0 || !dp || !idx || !beta) return RP_EBADSHAPE;
  const long long R = (long long)N * H * W;
  if (int e = bn_check(R, C)) return e;
  if (R * C >= (1LL << 32)) return RP_EBADSHAPE;                     // 32-bit element indices in the gather
  if (bf16) return bn_relu_pool_bwd_t((const bf16s*)dp, idx, (const bf16s*)x, mean, rstd, gamma, beta, (bf16s*)dx, dgamma, dbeta, partial, c12, N, H,
                                      W, C, training, (hipStream_t)stream);
  return bn_relu_pool_bwd_t((const float*)dp, idx, (const float*)x, mean, rstd, gamma, beta, (float*)dx, dgamma, dbeta, partial, c12, N, H, W, C,
                            training, (hipStream_t)stream);
}
