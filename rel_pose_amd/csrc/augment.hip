// augment.hip -- the training-time image augmentation on a resident batch (rp_augment_pairs).
//
// Reference: RGBDAugmentor (src/data_readers/augmentation.py:7-37) -- torchvision ColorJitter(brightness, contrast,
// saturation, hue, in a random order) + RandomGrayscale on the two images of a pair glued side by side (one parameter draw
// per pair), then a nearest-neighbour resize to the training size -- run per sample, through PIL, inside DataLoader workers.
// Measured here (tools/loader_bench.py, profiles/r2_loader_bench.txt): that path does 26 pairs/s per host core and the
// fp32 batches it ships through the DataLoader's IPC top out at ~330 pairs/s, one fifth of what ONE MI355X consumes.
// This kernel moves everything after the PNG decode to the GPU: workers return the decoded uint8 pair, the batch is uploaded
// once (1.8 MB per pair instead of 4.7 MB) and jittered + resized here in two passes over HBM:
//   pass 1 (only if the chain contains contrast): the integer sum of the "L" values of each PAIR after the ops that precede contrast
//           in its order (ImageEnhance.Contrast blends with the grey level int(mean(L) + 0.5) of the image it is given, and the
//           reference glues the pair into one image first);
//   pass 2: per output pixel -- nearest source pixel (floor(dst * in / out), the index rule of F.interpolate), BGR uint8 -> RGB, the
//           four ops in the pair's order, optional greyscale, ToTensor's / 255 and the reference's `255 *` in fp32, written as the
//           model's input layout [B,2,3,Ho,Wo] (BGR).
// The colour arithmetic IS PIL's 8-bit integer arithmetic (the reference runs torchvision's PIL backend on a ToPILImage image): every
// op rounds to uint8 the way Pillow's C does -- Blend.c (float32 `a + alpha (b - a)`, truncated / clipped), Convert.c rgb2l (fixed
// point), rgb2hsv_row / hsv2rgb (float32 quotients, double folding, truncation / round-half-up) -- restated in
// rel_pose_amd/data_readers/augmentation.py (the checker in tests/, itself exhaustively equal to Pillow and bit-identical to samples
// of the reference's own reader, tests/golden/reference_readers.npz).  Same parameter row => the reference's pixels, bit for bit.
// No fused multiply-adds anywhere in this file: the C code it mirrors rounds every product.
#include "common.h"
#include "../../include/relpose_hip.h"

#pragma clang fp contract(off)

namespace {

struct Rgb { int r, g, b; };        // uint8 values

RP_DEV int clip8(int v) { return min(max(v, 0), 255); }
RP_DEV int luma8(Rgb c) { return (c.r * 19595 + c.g * 38470 + c.b * 7471 + 0x8000) >> 16; }          // convert("L")
RP_DEV int blend8(int deg, int img, float a, bool interp) {                                          // Image.blend(deg, img, a)
  // plain operators, NOT __fmul_rn / __fadd_rn: those header inlines carry the default `contract` flag and fuse into one v_fma after
  // inlining (measured: 2 % of the pixels off by one level); under this file's `fp contract(off)` the two roundings stay separate
  const float prod = a * (float)(img - deg);
  const float t = (float)deg + prod;
  if (interp) return (int)t;
  return t <= 0.f ? 0 : (t >= 255.f ? 255 : (int)t);
}
RP_DEV Rgb blend(Rgb c, Rgb deg, float a) {
  const bool interp = a >= 0.f && a <= 1.f;
  return Rgb{blend8(deg.r, c.r, a, interp), blend8(deg.g, c.g, a, interp), blend8(deg.b, c.b, a, interp)};
}
RP_DEV Rgb hue_shift(Rgb c, int shift) {
  // RGB -> HSV (rgb2hsv_row)
  const int maxc = max(c.r, max(c.g, c.b)), minc = min(c.r, min(c.g, c.b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - c.r), cr), gc = __fdiv_rn((float)(maxc - c.g), cr), bc = __fdiv_rn((float)(maxc - c.b), cr);
    float h;
    if (c.r == maxc) h = bc - gc;
    else if (c.g == maxc) h = (float)(2.0 + (double)rc - (double)bc);
    else h = (float)(4.0 + (double)gc - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = clip8((int)((double)h * 255.0));
    us = clip8((int)((double)s * 255.0));
  }
  uh = (uh + shift) & 255;                                  // torchvision functional_pil.adjust_hue: uint8 wrap-around
  // HSV -> RGB (hsv2rgb)
  if (us == 0) return Rgb{uv, uv, uv};
  const double hf = (double)uh * 6.0 / 255.0;
  const int i = (int)floor(hf);
  const double f = (double)(float)(hf - (double)i), fs = (double)(float)((double)us / 255.0), vf = (double)uv;
  const int p = clip8((int)floor(vf * (1.0 - fs) + 0.5)), q = clip8((int)floor(vf * (1.0 - fs * f) + 0.5)),
            t = clip8((int)floor(vf * (1.0 - fs * (1.0 - f)) + 0.5));
  switch (i % 6) {
    case 0: return Rgb{uv, t, p};
    case 1: return Rgb{q, uv, p};
    case 2: return Rgb{p, uv, t};
    case 3: return Rgb{p, q, uv};
    case 4: return Rgb{t, p, uv};
    default: return Rgb{uv, p, q};
  }
}

// params per pair: [order0..3 (0 brightness, 1 contrast, 2 saturation, 3 hue, < 0: skip), b, c, s, h, gray]
constexpr int NPRM = 9;

// apply ops order[first .. last) to one pixel; `mean` is the pair's grey level for contrast
RP_DEV Rgb apply_ops(Rgb c, const float* prm, int first, int last, int mean) {
  for (int k = first; k < last; ++k) {
    const int op = (int)prm[k];
    if (op == 0) c = blend(c, Rgb{0, 0, 0}, prm[4]);
    else if (op == 1) c = blend(c, Rgb{mean, mean, mean}, prm[5]);
    else if (op == 2) { const int l = luma8(c); c = blend(c, Rgb{l, l, l}, prm[6]); }
    else if (op == 3) c = hue_shift(c, (int)((double)prm[7] * 255.0));       // int(shift * 255): truncation toward zero
  }
  return c;
}

__global__ __launch_bounds__(256) void aug_mean_kernel(const unsigned char* __restrict__ img, const float* __restrict__ prm,
                                                       unsigned long long* __restrict__ part, long long px_per_pair, int nblk) {
  const int b = blockIdx.y;
  const float* p = prm + b * NPRM;
  int cpos = 4;
  for (int k = 0; k < 4; ++k) if ((int)p[k] == 1) cpos = k;
  unsigned long long acc = 0;
  if (cpos < 4) {
    const unsigned char* src = img + (long long)b * px_per_pair * 3;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < px_per_pair; i += (long long)nblk * 256) {
      Rgb c{src[3 * i + 2], src[3 * i + 1], src[3 * i]};
      c = apply_ops(c, p, 0, cpos, 0);
      acc += (unsigned long long)luma8(c);
    }
  }
  __shared__ unsigned long long red[256];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) part[(long long)b * nblk + blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void aug_apply_kernel(const unsigned char* __restrict__ img, const float* __restrict__ prm,
                                                        const unsigned long long* __restrict__ part, float* __restrict__ out, int H, int W,
                                                        int Ho, int Wo, int nblk) {
  const int b = blockIdx.z, im = blockIdx.y;
  const float* p = prm + b * NPRM;
  __shared__ int mean_s;
  if (threadIdx.x == 0) {
    unsigned long long m = 0;
    for (int i = 0; i < nblk; ++i) m += part[(long long)b * nblk + i];      // integers: exact in any order
    mean_s = (int)((double)m / (2.0 * H * W) + 0.5);                        // int(ImageStat.mean + 0.5)
  }
  __syncthreads();
  const int mean = mean_s;
  const long long n_out = (long long)Ho * Wo;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= n_out) return;
  const int oy = (int)(idx / Wo), ox = (int)(idx % Wo);
  const int sy = min((int)floorf(oy * ((float)H / Ho)), H - 1), sx = min((int)floorf(ox * ((float)W / Wo)), W - 1);
  const unsigned char* s = img + ((((long long)b * 2 + im) * H + sy) * W + sx) * 3;
  Rgb c{s[2], s[1], s[0]};
  c = apply_ops(c, p, 0, 4, mean);
  if (p[8] != 0.f) { const int g = luma8(c); c = Rgb{g, g, g}; }
  float* o = out + (((long long)b * 2 + im) * 3) * n_out + idx;
  o[0] = 255.f * __fdiv_rn((float)c.b, 255.f);          // ToTensor: uint8 / 255 in fp32; the reference: 255 * that
  o[n_out] = 255.f * __fdiv_rn((float)c.g, 255.f);
  o[2 * n_out] = 255.f * __fdiv_rn((float)c.r, 255.f);
}

}  // namespace

extern "C" int rp_augment_blocks(void) { return 64; }

extern "C" int rp_augment_pairs(const unsigned char* images, const float* params, float* out, double* workspace, int B, int H,
                                int W, int Ho, int Wo, void* stream) {
  if (B <= 0 || H <= 0 || W <= 0 || Ho <= 0 || Wo <= 0 || !images || !params || !out || !workspace) return RP_EBADSHAPE;
  hipStream_t st = (hipStream_t)stream;
  const int nblk = rp_augment_blocks();
  // (the workspace holds the per-block integer partial sums: 8 bytes each, the `double` of the signature is only its size)
  unsigned long long* part = reinterpret_cast<unsigned long long*>(workspace);
  hipLaunchKernelGGL(aug_mean_kernel, dim3(nblk, B), dim3(256), 0, st, images, params, part, (long long)2 * H * W, nblk);
  RP_CHECK_LAUNCH();
  const long long n_out = (long long)Ho * Wo;
  hipLaunchKernelGGL(aug_apply_kernel, dim3((unsigned)((n_out + 255) / 256), 2, B), dim3(256), 0, st, images, params,
                     (const unsigned long long*)part, out, H, W, Ho, Wo, nblk);
  RP_CHECK_LAUNCH();
  return RP_OK;
}
