// augment.hip -- the training-time image augmentation on a resident batch (rp_augment_pairs).
//
// Reference: RGBDAugmentor (src/data_readers/augmentation.py:7-37) -- torchvision ColorJitter(brightness, contrast,
// saturation, hue, in a random order) + RandomGrayscale on the two images of a pair glued side by side (one parameter draw
// per pair), then a nearest-neighbour resize to the training size -- run per sample, through PIL, inside DataLoader workers.
// Measured here (tools/loader_bench.py, profiles/r2_loader_bench.txt): that path does 26 pairs/s per host core and the
// fp32 batches it ships through the DataLoader's IPC top out at ~330 pairs/s, one fifth of what ONE MI355X consumes.
// This kernel moves everything after the PNG decode to the GPU: workers return the decoded uint8 pair, the batch is uploaded
// once (1.8 MB per pair instead of 4.7 MB) and jittered + resized here in two passes over HBM:
//   pass 1 (only if the chain contains contrast): mean luma of each PAIR after the ops that precede contrast in its order
//           (torchvision's adjust_contrast blends with the mean grey level of the image it is given);
//   pass 2: per output pixel -- nearest source pixel (floor(dst * in / out), the index rule of F.interpolate), BGR uint8
//           -> RGB in [0,1], the four ops in the pair's order, optional greyscale, back to BGR 0..255 fp32, written as the
//           model's input layout [B,2,3,Ho,Wo].
// The colour maths follows rel_pose_amd/data_readers/augmentation.py op for op (which is the checker in tests/).
#include "common.h"
#include "../../include/relpose_hip.h"

namespace {

struct Rgb { float r, g, b; };

RP_DEV float clamp01(float x) { return fminf(fmaxf(x, 0.f), 1.f); }
RP_DEV float luma(Rgb c) { return 0.299f * c.r + 0.587f * c.g + 0.114f * c.b; }
RP_DEV Rgb blend(Rgb a, float m, float f) {        // f * a + (1 - f) * m, clamped
  return Rgb{clamp01(f * a.r + (1.f - f) * m), clamp01(f * a.g + (1.f - f) * m), clamp01(f * a.b + (1.f - f) * m)};
}
RP_DEV Rgb hue_shift(Rgb c, float shift) {
  const float maxc = fmaxf(c.r, fmaxf(c.g, c.b)), minc = fminf(c.r, fminf(c.g, c.b));
  const bool eq = maxc == minc;
  const float cr = maxc - minc;
  const float s = cr / (eq ? 1.f : maxc);
  const float crd = eq ? 1.f : cr;
  const float rc = (maxc - c.r) / crd, gc = (maxc - c.g) / crd, bc = (maxc - c.b) / crd;
  float h;
  if (maxc == c.r) h = bc - gc;
  else if (maxc == c.g) h = 2.f + rc - bc;
  else h = 4.f + gc - rc;
  h = fmodf(h / 6.f + 1.f, 1.f);
  h = fmodf(h + shift + 1.f, 1.f);
  const float v = maxc;
  const float i6 = floorf(h * 6.f), f = h * 6.f - i6;
  const int i = ((int)i6) % 6;
  const float p = clamp01(v * (1.f - s)), q = clamp01(v * (1.f - f * s)), t = clamp01(v * (1.f - (1.f - f) * s));
  switch (i) {
    case 0: return Rgb{v, t, p};
    case 1: return Rgb{q, v, p};
    case 2: return Rgb{p, v, t};
    case 3: return Rgb{p, q, v};
    case 4: return Rgb{t, p, v};
    default: return Rgb{v, p, q};
  }
}

// params per pair: [order0..3 (0 brightness, 1 contrast, 2 saturation, 3 hue), b, c, s, h, gray]
constexpr int NPRM = 9;

// apply ops order[first .. last) to one pixel; `mean` is the pair's mean luma for contrast
RP_DEV Rgb apply_ops(Rgb c, const float* prm, int first, int last, float mean) {
  for (int k = first; k < last; ++k) {
    const int op = (int)prm[k];
    if (op == 0) c = blend(c, 0.f, prm[4]);
    else if (op == 1) c = blend(c, mean, prm[5]);
    else if (op == 2) c = blend(c, luma(c), prm[6]);
    else c = hue_shift(c, prm[7]);
  }
  return c;
}

__global__ __launch_bounds__(256) void aug_mean_kernel(const unsigned char* __restrict__ img, const float* __restrict__ prm,
                                                       double* __restrict__ part, long long px_per_pair, int nblk) {
  const int b = blockIdx.y;
  const float* p = prm + b * NPRM;
  int cpos = 4;
  for (int k = 0; k < 4; ++k) if ((int)p[k] == 1) cpos = k;
  double acc = 0.0;
  if (cpos < 4) {
    const unsigned char* src = img + (long long)b * px_per_pair * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < px_per_pair; i += (long long)nblk * 256) {
      Rgb c{src[3 * i + 2] * (1.f / 255.f), src[3 * i + 1] * (1.f / 255.f), src[3 * i] * (1.f / 255.f)};
      c = apply_ops(c, p, 0, cpos, 0.f);
      acc += (double)luma(c);
    }
  }
  __shared__ double red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(long long)b * nblk + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void aug_apply_kernel(const unsigned char* __restrict__ img, const float* __restrict__ prm,
                                                        const double* __restrict__ part, float* __restrict__ out, int H, int W,
                                                        int Ho, int Wo, int nblk) {
  const int b = blockIdx.z, im = blockIdx.y;
  const float* p = prm + b * NPRM;
  __shared__ float mean_s;
  if (threadIdx.x == 0) {
    double m = 0.0;
    for (int i = 0; i < nblk; ++i) m += part[(long long)b * nblk + i];      // fixed order
    mean_s = (float)(m / (2.0 * H * W));
  }
  __syncthreads();
  const float mean = mean_s;
  const long long n_out = (long long)Ho * Wo;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_out) return;
  const int oy = (int)(idx / Wo), ox = (int)(idx % Wo);
  const int sy = min((int)floorf(oy * ((float)H / Ho)), H - 1), sx = min((int)floorf(ox * ((float)W / Wo)), W - 1);
  const unsigned char* s = img + ((((long long)b * 2 + im) * H + sy) * W + sx) * 3;
  Rgb c{s[2] * (1.f / 255.f), s[1] * (1.f / 255.f), s[0] * (1.f / 255.f)};
  c = apply_ops(c, p, 0, 4, mean);
  if (p[8] != 0.f) { const float g = luma(c); c = Rgb{g, g, g}; }
  float* o = out + (((long long)b * 2 + im) * 3) * n_out + idx;
  o[0] = 255.f * c.b;
  o[n_out] = 255.f * c.g;
  o[2 * n_out] = 255.f * c.r;
}

}  // namespace

extern "C" int rp_augment_blocks(void) { return 64; }

extern "C" int rp_augment_pairs(const unsigned char* images, const float* params, float* out, double* workspace, int B, int H,
                                int W, int Ho, int Wo, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || !images || !params || !out || !workspace) return RP_EBADSHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = rp_augment_blocks();
  hipLaunchKernelGGL(aug_mean_kernel, dim3(nblk, B), dim3(256), 0, st, images, params, workspace, (long long)2 * H * W, nblk);
  RP_CHECK_LAUNCH();
  const long long n_out = (long long)Ho * Wo;
  hipLaunchKernelGGL(aug_apply_kernel, dim3((unsigned)((n_out + 255) / 256), 2, B), dim3(256), 0, st, images, params,
                     (const double*)workspace, out, H, W, Ho, Wo, nblk);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
