// split3_lab.hip -- what does v_dot2c_f32_bf16 do with an inline-constant operand?  Checks the on-chip 3-limb split of dw192_split3.hip
// element by element against a host restatement.   hipcc --offload-arch=gfx950 -O3 tools/lab/split3_lab.hip -o /tmp/split3_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

template <int MODE> __device__ void split_pair(float a, float b, unsigned* out, float* res) {
  unsigned c10 = 0x0000BF80u, c01 = 0xBF800000u;
  if (MODE == 1) { asm volatile("" : "+s"(c10)); asm volatile("" : "+s"(c01)); }
  if (MODE == 2) { asm volatile("" : "+v"(c10)); asm volatile("" : "+v"(c01)); }
  const bf16x2 m10 = __builtin_bit_cast(bf16x2, c10), m01 = __builtin_bit_cast(bf16x2, c01);
  const bf16x2 p0 = {(__bf16)a, (__bf16)b};
  const float ra = __builtin_amdgcn_fdot2_f32_bf16(p0, m10, a, false);
  const float rb = __builtin_amdgcn_fdot2_f32_bf16(p0, m01, b, false);
  const bf16x2 p1 = {(__bf16)ra, (__bf16)rb};
  const float sa = __builtin_amdgcn_fdot2_f32_bf16(p1, m10, ra, false);
  const float sb = __builtin_amdgcn_fdot2_f32_bf16(p1, m01, rb, false);
  const bf16x2 p2 = {(__bf16)sa, (__bf16)sb};
  out[0] = __builtin_bit_cast(unsigned, p0); out[1] = __builtin_bit_cast(unsigned, p1); out[2] = __builtin_bit_cast(unsigned, p2);
  res[0] = ra; res[1] = rb; res[2] = sa; res[3] = sb;
}
template <int MODE> __global__ void k(const float* x, unsigned* out, float* res, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) split_pair<MODE>(x[2 * i], x[2 * i + 1], out + 3 * i, res + 4 * i);
}
static float bf(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }
static unsigned short rne(float f) { unsigned u; memcpy(&u, &f, 4); u += 0x7fffu + ((u >> 16) & 1); return (unsigned short)(u >> 16); }
int main() {
  const int n = 1 << 16;
  std::vector<float> x(2 * n);
  unsigned s = 12345;
  for (auto& v : x) { s = s * 1664525u + 1013904223u; float u = (float)(s >> 8) / 16777216.0f - 0.5f; s = s * 1664525u + 1013904223u; v = u * ldexpf(1.0f, (int)(s >> 27) - 16); }
  float *dx, *dres; unsigned* dout;
  hipMalloc(&dx, 8 * n); hipMalloc(&dout, 12 * n); hipMalloc(&dres, 16 * n);
  hipMemcpy(dx, x.data(), 8 * n, hipMemcpyHostToDevice);
  std::vector<unsigned> out(3 * n); std::vector<float> res(4 * n);
  for (int mode = 0; mode < 3; ++mode) {
    if (mode == 0) k<0><<<n / 256, 256>>>(dx, dout, dres, n);
    if (mode == 1) k<1><<<n / 256, 256>>>(dx, dout, dres, n);
    if (mode == 2) k<2><<<n / 256, 256>>>(dx, dout, dres, n);
    hipMemcpy(out.data(), dout, 12 * n, hipMemcpyDeviceToHost);
    hipMemcpy(res.data(), dres, 16 * n, hipMemcpyDeviceToHost);
    int bad_limb = 0, bad_sum = 0, bad_r = 0; double worst = 0;
    for (int i = 0; i < n; ++i) for (int h = 0; h < 2; ++h) {
      const float v = x[2 * i + h];
      const unsigned short l0 = (out[3 * i] >> (16 * h)) & 0xffff, l1 = (out[3 * i + 1] >> (16 * h)) & 0xffff, l2 = (out[3 * i + 2] >> (16 * h)) & 0xffff;
      const unsigned short e0 = rne(v); const float r1 = v - bf(e0); const unsigned short e1 = rne(r1); const float r2 = r1 - bf(e1); const unsigned short e2 = rne(r2);
      if (l0 != e0 || l1 != e1 || l2 != e2) ++bad_limb;
      if (res[4 * i + h] != r1 || res[4 * i + 2 + h] != r2) ++bad_r;
      const double rec = (double)bf(l0) + (double)bf(l1) + (double)bf(l2);
      if (rec != (double)v) { ++bad_sum; worst = fmax(worst, fabs(rec - v) / fabs(v)); }
    }
    printf("mode %d (%s constants): limbs != host %d, residuals != host %d, l0+l1+l2 != x %d of %d (worst rel %.3e)   e.g. x=%.9g r1=%.9g (host %.9g)\n", mode,
           mode == 0 ? "compiler-chosen" : mode == 1 ? "SGPR" : "VGPR", bad_limb, bad_r, bad_sum, 2 * n, worst, x[0], res[0], x[0] - bf(rne(x[0])));
  }
  return 0;
}
