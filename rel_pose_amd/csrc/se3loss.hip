// se3loss.hip -- the geodesic pose loss of the training step as ONE kernel (forward value + exact derivatives).
//
// Replaces, for the hot loop only, the ~240 tiny elementwise launches PyTorch makes for
//   dP = Ps[:,jj] * Ps[:,ii].inv();  dG = Gs[:,jj] * Gs[:,ii].inv();  d = (dG * dP.inv()).log();  |tau|, |phi| means
// (reference src/geom/losses.py:3-21 over lietorch; restated on plain SE(3) formulas in rel_pose_amd/se3.py -- this kernel
// evaluates exactly those formulas, branch for branch).  One thread per (pair b, term j); derivatives with respect to the 14
// numbers of Gs[b] by forward-mode dual arithmetic (a Dual carries the value and 14 partials), so the backward is one
// scaled sum and cannot drift from the forward.  Parity with lietorch itself stays unpinned.
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

constexpr int ND = 14;

struct Dual {
  float v;
  float d[ND];
};

RP_DEV Dual cst(float v) {
  Dual r;
  r.v = v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = 0.f;
  return r;
}
RP_DEV Dual var(float v, int k) {
  Dual r = cst(v);
  r.d[k] = 1.f;
  return r;
}
RP_DEV Dual operator+(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v + b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] + b.d[i];
  return r;
}
RP_DEV Dual operator-(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v - b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] - b.d[i];
  return r;
}
RP_DEV Dual operator-(const Dual& a) {
  Dual r;
  r.v = -a.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = -a.d[i];
  return r;
}
RP_DEV Dual operator*(const Dual& a, const Dual& b) {
  Dual r;
  r.v = a.v * b.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = a.d[i] * b.v + a.v * b.d[i];
  return r;
}
RP_DEV Dual operator*(float s, const Dual& a) {
  Dual r;
  r.v = s * a.v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = s * a.d[i];
  return r;
}
RP_DEV Dual operator/(const Dual& a, const Dual& b) {
  Dual r;
  const float ib = 1.f / b.v;
  r.v = a.v * ib;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (a.d[i] - r.v * b.d[i]) * ib;
  return r;
}
RP_DEV Dual chain(float v, float dv, const Dual& a) {     // f(a) with f'(a) = dv
  Dual r;
  r.v = v;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = dv * a.d[i];
  return r;
}
RP_DEV Dual dsin(const Dual& a) { return chain(sinf(a.v), cosf(a.v), a); }
RP_DEV Dual dcos(const Dual& a) { return chain(cosf(a.v), -sinf(a.v), a); }
RP_DEV Dual datan2(const Dual& y, const Dual& x) {
  Dual r;
  const float den = 1.f / (x.v * x.v + y.v * y.v);
  r.v = atan2f(y.v, x.v);
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (x.v * y.d[i] - y.v * x.d[i]) * den;
  return r;
}
// |(x,y,z)| with torch's convention for the gradient at 0 (zero, not NaN)
RP_DEV Dual norm3(const Dual& x, const Dual& y, const Dual& z) {
  const float n = sqrtf(x.v * x.v + y.v * y.v + z.v * z.v);
  Dual r;
  r.v = n;
  const float in = n > 0.f ? 1.f / n : 0.f;
#pragma unroll
  for (int i = 0; i < ND; ++i) r.d[i] = (x.v * x.d[i] + y.v * y.d[i] + z.v * z.d[i]) * in;
  return r;
}

struct V3 { Dual x, y, z; };
struct Q4 { Dual x, y, z, w; };
struct Pose { V3 t; Q4 q; };

RP_DEV V3 cross(const V3& a, const V3& b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
RP_DEV V3 add(const V3& a, const V3& b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
RP_DEV V3 scale(const Dual& s, const V3& a) { return {s * a.x, s * a.y, s * a.z}; }
RP_DEV Q4 qmul(const Q4& a, const Q4& b) {          // rel_pose_amd/se3.py:_qmul
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
RP_DEV Q4 qconj(const Q4& q) { return {-q.x, -q.y, -q.z, q.w}; }
RP_DEV V3 qrot(const Q4& q, const V3& v) {           // v + w t + u x t,  t = 2 u x v
  const V3 u = {q.x, q.y, q.z};
  V3 t = cross(u, v);
  t = {2.f * t.x, 2.f * t.y, 2.f * t.z};
  return add(add(v, scale(q.w, t)), cross(u, t));
}
RP_DEV Pose pmul(const Pose& a, const Pose& b) { return {add(a.t, qrot(a.q, b.t)), qmul(a.q, b.q)}; }
RP_DEV Pose pinv(const Pose& a) {
  const Q4 qi = qconj(a.q);
  const V3 r = qrot(qi, a.t);
  return {{-r.x, -r.y, -r.z}, qi};
}

// se3.py:_so3_log
RP_DEV V3 so3_log(const Q4& q) {
  const Dual n = norm3(q.x, q.y, q.z);
  const float sign = q.w.v < 0.f ? -1.f : 1.f;
  const Dual w = sign * q.w;
  Dual fac;
  if (n.v < 1e-6f) {
    Dual wc = w;
    if (wc.v < 1e-12f) wc = cst(1e-12f);            // clamp_min: gradient 0 below the clamp
    fac = cst(2.f) / wc - (2.f / 3.f) * (n * n) / (wc * wc * wc);
  } else {
    fac = 2.f * datan2(n, w) / n;
  }
  const Dual f = sign * fac;
  return {q.x * f, q.y * f, q.z * f};
}

// se3.py:SE3.log -> (|tau|, |phi|)
RP_DEV void se3_log_norms(const Pose& p, Dual& ntau, Dual& nphi) {
  const V3 phi = so3_log(p.q);
  const Dual th = norm3(phi.x, phi.y, phi.z);
  Dual c;
  if (th.v < 1e-4f) {
    c = cst(1.f / 12.f) + (1.f / 720.f) * (th * th);
  } else {
    const Dual h = 0.5f * th;
    c = (cst(1.f) - th * dcos(h) / (2.f * dsin(h))) / (th * th);
  }
  // Vinv t = t - 0.5 phi x t + c phi x (phi x t)
  const V3 k1 = cross(phi, p.t);
  const V3 k2 = cross(phi, k1);
  const V3 tau = {p.t.x - 0.5f * k1.x + c * k2.x, p.t.y - 0.5f * k1.y + c * k2.y, p.t.z - 0.5f * k1.z + c * k2.z};
  ntau = norm3(tau.x, tau.y, tau.z);
  nphi = th;
}

RP_DEV Pose load_const(const float* p) {
  return {{cst(p[0]), cst(p[1]), cst(p[2])}, {cst(p[3]), cst(p[4]), cst(p[5]), cst(p[6])}};
}
RP_DEV Pose load_var(const float* p, int k0) {
  return {{var(p[0], k0), var(p[1], k0 + 1), var(p[2], k0 + 2)},
          {var(p[3], k0 + 3), var(p[4], k0 + 4), var(p[5], k0 + 5), var(p[6], k0 + 6)}};
}

// terms[b][j] = (|tau|, |phi|) of log(dG_j * dP_j^-1), dG_j = G[1-j] * G[j]^-1, dP_j = P[1-j] * P[j]^-1;
// grads[b][j][m][k] = d terms[b][j][m] / d Gs[b].flat[k], k < 14
__global__ void geodesic_terms_kernel(const float* __restrict__ Ps, const float* __restrict__ Gs, float* __restrict__ terms,
                                      float* __restrict__ grads, int B) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 2 * B) return;
  const int b = i >> 1, j = i & 1;
  const Pose P0 = load_const(Ps + (long long)b * 14), P1 = load_const(Ps + (long long)b * 14 + 7);
  const Pose G0 = load_var(Gs + (long long)b * 14, 0), G1 = load_var(Gs + (long long)b * 14 + 7, 7);
  const Pose dP = j == 0 ? pmul(P1, pinv(P0)) : pmul(P0, pinv(P1));
  const Pose dG = j == 0 ? pmul(G1, pinv(G0)) : pmul(G0, pinv(G1));
  Dual nt, np;
  se3_log_norms(pmul(dG, pinv(dP)), nt, np);
  terms[2 * i] = nt.v;
  terms[2 * i + 1] = np.v;
#pragma unroll
  for (int k = 0; k < ND; ++k) {
    grads[(long long)i * 2 * ND + k] = nt.d[k];
    grads[(long long)i * 2 * ND + ND + k] = np.d[k];
  }
}

// losses[0] = mean |tau|, losses[1] = mean |phi| over the 2B terms (fixed order);  dmean[m][b][k] = sum_j grads / (2B)
__global__ __launch_bounds__(256) void geodesic_reduce_kernel(const float* __restrict__ terms, const float* __restrict__ grads,
                                                              float* __restrict__ losses, float* __restrict__ dmean, int B) {
  __shared__ float red[2][256];
  float a = 0.f, c = 0.f;
  for (int i = threadIdx.x; i < 2 * B; i += 256) { a += terms[2 * i]; c += terms[2 * i + 1]; }
  red[0][threadIdx.x] = a;
  red[1][threadIdx.x] = c;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int l = 1; l < 256; ++l) { a += red[0][l]; c += red[1][l]; }
    losses[0] = a / (2.f * B);
    losses[1] = c / (2.f * B);
  }
  const float s = 1.f / (2.f * B);
  for (int i = threadIdx.x; i < B * ND; i += 256) {
    const int b = i / ND, k = i % ND;
    const float* g0 = grads + (long long)(2 * b) * 2 * ND;
    const float* g1 = g0 + 2 * ND;
    dmean[i] = s * (g0[k] + g1[k]);
    dmean[(long long)B * ND + i] = s * (g0[ND + k] + g1[ND + k]);
  }
}

}  // namespace

extern "C" int rp_geodesic_loss(const float* Ps, const float* Gs, float* losses, float* dmean, float* scratch, int B,
                                void* stream) {
  if (B <= 0) return RP_EBADSHAPE;
  hipStream_t st = (hipStream_t)stream;
  float* terms = scratch;                 // [2B][2]
  float* grads = scratch + 4 * (long long)B;   // [2B][2][14]
  hipLaunchKernelGGL(geodesic_terms_kernel, dim3((2 * B + 63) / 64), dim3(64), 0, st, Ps, Gs, terms, grads, B);
  RP_CHECK_LAUNCH();
  hipLaunchKernelGGL(geodesic_reduce_kernel, dim3(1), dim3(256), 0, st, (const float*)terms, (const float*)grads, losses, dmean, B);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
