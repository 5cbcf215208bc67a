/*
 * relpose_hip.h -- C ABI of librelpose_hip.so (gfx950 / MI355X).
 *
 * The reference (crockwell/rel_pose) has NO FFI / plugin interface: its hot path is a chain of stock
 * ATen calls inside Python modules.  This header is therefore the build-defined
 * boundary that SURVEY.md 8b specifies: one shared object, extern "C", plain pointers and sizes, no
 * torch types.  Each entry point names the reference op chain (file:line under /root/reference) it
 * replaces.  INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - every pointer is DEVICE memory owned by the caller (PyTorch tensors); the library never
 *     allocates, frees or keeps global state; workspaces are passed in.
 *   - `stream` is a hipStream_t (void*); launches are enqueued there and never synchronise.
 *   - return value: 0 = ok, <0 = RP_E* argument error, >0 = hipError_t from the launch.
 *   - all tensors are fp32, row-major; "ld*" = row stride in floats.
 *   - token count per image is 576 (24x24), head dim 64 (reference src/model.py:19-23).
 */
#ifndef RELPOSE_HIP_H
#define RELPOSE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RP_OK 0
#define RP_EBADSHAPE (-1)
#define RP_EALIGN (-2)
#define RP_EWORKSPACE (-3)
#define RP_EUNSUPPORTED (-4)

/* library / ABI version and target arch string ("gfx950").
 * RP_ABI_VERSION is bumped whenever an entry point is added, removed or changes its arguments; RP_ABI_EXPORTS is the number of
 * functions this header declares.  rp_abi_version() / rp_abi_export_count() return the values the library was COMPILED with, so a
 * binding (rel_pose_amd/_lib.py parses both macros and counts the declarations) rejects a stale .so at load time instead of
 * failing later on a missing symbol. */
#define RP_ABI_VERSION 26
#define RP_ABI_EXPORTS 110
int rp_abi_version(void);
int rp_abi_export_count(void);
const char* rp_target_arch(void);

/* ---------------------------------------------------------------------------------------------
 * Dense contraction C = epilogue(op(A) * op(B))   (fp32 MFMA v_mfma_f32_32x32x2_f32)
 * replaces nn.Linear forward and its autograd (dX, dW):
 *   vision_transformer.py:323,331 (qkv, proj), :191-195,233-234 (cross qkv, proj_fundamental),
 *   vit_layers/mlp.py:20-26 (fc1, GELU, fc2), src/model.py:91-98 (pose_regressor).
 * a_layout: 0 = A stored [M,K] (K contiguous), 1 = A stored [K,M] (M contiguous)
 * b_layout: 0 = B stored [N,K] (K contiguous; nn.Linear weight), 1 = B stored [K,N]
 * contiguous extents and all ld* must be multiples of 4 (K only when an operand is K-contiguous).
 * batch > 1: independent problems at A + i*stride_a etc. (split_k must be 1).
 * split_k > 1: partial products go to `workspace` ([split_k][M][N] floats) and are reduced in a
 *   fixed order by a second kernel that also applies the epilogue (deterministic).
 * epilogue, in this order:  v += bias[n];  pre_out[m][n] = v;  act (0 none, 1 exact-erf GELU,
 *   2 ReLU);  dact (0 none, 1: v *= gelu'(aux[m][n]), 2: v = aux[m][n] > 0 ? v : 0);
 *   v += residual[m][n];  C[m][n] = v.
 * ------------------------------------------------------------------------------------------- */
typedef struct RpGemm {
  const float* A;
  const float* B;
  float* C;
  int M, N, K;
  int lda, ldb, ldc;
  int a_layout, b_layout;
  int batch;
  long long stride_a, stride_b, stride_c;
  int split_k;
  float* workspace;
  size_t workspace_bytes;
  const float* bias;
  float* pre_out; /* ld = ldc */
  int act;
  int dact;
  const float* aux; /* ld = ldc */
  const float* residual; /* ld = ldc */
  int trans_c; /* split_k > 1 only, no epilogue operands: store C^T, i.e. C[n*ldc + m] */
  int precision; /* how the fp32 operands are multiplied (inputs, outputs and accumulation are fp32 in every case):
                  *   0  exact fp32 MFMA (v_mfma_f32_32x32x2_f32)
                  *   3  split-bf16, 3 limbs per operand, the 6 limb products >= 2^-16 on v_mfma_f32_32x32x16_bf16:
                  *      fp32-grade result (dropped terms <= 2^-24 relative) at 16/6 the matrix-pipe rate
                  *   1  operands rounded to bf16 (the bf16 configuration, BASELINE.json configs[4]) */
  float* colsum_part; /* optional: [2*ceil(M/(64*TM))][N] floats; row (2*mt + wave row) receives the column sums of the FINAL
                       * values that tile stored over its 32*TM rows -- one small rp_colsum over it yields sum_m C[m][n] (the bias
                       * gradient when C is a pre-activation gradient) without re-reading C.  split_k = batch = 1, N % 4 == 0,
                       * not the [K,M]x[K,N] layout, and the tile must be TN <= 2 (true whenever aux / residual is given). */
  /* LayerNorm backward fused into the epilogue (all NULL = off).  With ln_x set, the product op(A) op(B) is taken as the
   * gradient of a LayerNorm OUTPUT (eps / affine as rp_layernorm_fwd) and C receives the gradient of the LayerNorm INPUT:
   *   xhat = (ln_x - ln_mean) * ln_rstd ;  g = P * ln_gamma ;  C = ln_rstd * (g - mean_n(g) - xhat * mean_n(g * xhat)) + residual
   * ln_part [ceil(M/64)][np*192] (np = 3 with a residual, else 2) receives per-64-row-tile column sums of
   * P * xhat (-> d gamma), P (-> d beta) and residual (-> the bias gradient of the Linear that produced that branch); one
   * rp_colsum over it finishes them.  Requirements: N == 192 == ldc, ln_x / residual contiguous [M,192], no bias / act / aux,
   * split_k = batch = 1 (any operand precision: the epilogue arithmetic is fp32).  Replaces reference autograd of vision_transformer.py:352-353 (LayerNorm backward). */
  const float* ln_x;
  const float* ln_mean;
  const float* ln_rstd;
  const float* ln_gamma;
  float* ln_part;
  /* optional hipEvent_t pair recorded on `stream` immediately before and after the MAIN kernel of this call (not the split-K
   * reduce): bench.py's per-launch timing of one kernel instance.  NULL: nothing recorded. */
  void* ev_start;
  void* ev_stop;
  int io_bf16; /* precision 1 only (the bf16 configuration): bf16 STORAGE of activation-sized tensors, a bit mask -- 1: A holds bf16
                * (lda in elements), 2: C and pre_out are written as bf16, 4: aux holds bf16.  The MFMA operands are bf16 in this
                * precision anyway; reading / writing them as bf16 halves the bytes of these HBM-bound launches.  Bit 2 needs the plain /
                * bias / GELU / GELU' epilogues without split-K, residual or ln_*; B, bias, residual stay fp32. */
  int defer_reduce; /* split_k > 1 with no epilogue operands only: launch the main kernel and leave the partial slabs in `workspace`
                     * (which the caller then keeps alive and private to this call); the caller finishes C later with ONE
                     * rp_splitk_reduce_multi over several such calls -- the four weight-gradient GEMMs of a transformer block end in
                     * one reduce launch instead of four */
} RpGemm;

/* hipEvent helpers for ev_start / ev_stop (so that a ctypes host needs no second library): create / destroy / elapsed ms */
void* rp_event_create(void);
void rp_event_destroy(void* ev);
float rp_event_elapsed_ms(void* start, void* stop);
int rp_gemm(const RpGemm* g, void* stream);
size_t rp_gemm_workspace_bytes(int M, int N, int split_k);
/* finish up to RP_SPLITK_MAX deferred split-K products in one launch: C[m][n] (trans_c: C[n*ldc + m]) = sum_z ws[z][m][n], summed in
 * the same fixed order as rp_gemm's own reduce (bit-identical results) */
#define RP_SPLITK_MAX 8
typedef struct RpSplitkTask {
  const float* ws;
  float* C;
  int M, N, ldc, split_k, trans_c;
} RpSplitkTask;
int rp_splitk_reduce_multi(const RpSplitkTask* tasks, int n, void* stream);
/* up to RP_TRANSPOSE_MAX small transposes dst[c][r] = src[r][c] (row-major, contiguous) in one launch: the W^T copies the
 * row-resident input-gradient kernels want (rp_linear_rows192 on the transposed weight, rp_mlp_fused_bwd) */
#define RP_TRANSPOSE_MAX 24
typedef struct RpTransposeTask {
  const float* src;
  float* dst;
  int rows, cols;
} RpTransposeTask;
int rp_transpose_multi(const RpTransposeTask* tasks, int n, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Channels-last BatchNorm2d fused with the residual add and ReLU that follow it in the CNN front-end
 * (torchvision BasicBlock as driven by src/model.py:127-132; ResidualBlock, src/modules/extractor.py:51-65).
 * x, y, dy, dx, residual: [R, C], R = N*H*W rows, C contiguous channels (C % 4 == 0, C <= 256).
 *   rp_bn_stats      mean[c], rstd[c] = 1/sqrt(biased var + eps) over the R rows; if running_* != NULL also the PyTorch
 *                    running-statistics update (momentum, unbiased variance).  partial: [rp_bn_partial_blocks(R)][2][C] doubles.
 *   rp_bn_apply_fwd  y = relu?((x - mean) * rstd * gamma + beta (+ residual))   (training: batch stats; eval: running stats)
 *   rp_bn_bwd        g = dy * (relu ? y > 0 : 1) (y == NULL: the forward had no residual and its output sign is re-evaluated
 *                    from x, bit-identically -- y then need not be kept);  dbeta = sum g;  dgamma = sum g * xhat;
 *                    training: dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat));  eval: dx = gamma*rstd*g;
 *                    dres != NULL: g is also stored there (gradient of the residual branch).  c12: [2][C] floats scratch.
 * `bf16` (these and the pool entry points below): 0 = the activation tensors (x, y, residual, dy, dx, dres, the pooled tensors) are
 * fp32; 1 = they are bf16 in memory (2 bytes per element, the storage MIOpen's bf16 convolutions of the bf16 configuration read and
 * write -- BASELINE.json configs[4]), widened exactly on load and rounded to nearest-even on store.  Statistics, per-channel
 * parameters and all arithmetic are fp32 / double in both cases.
 * ------------------------------------------------------------------------------------------- */
int rp_bn_partial_blocks(long long R);
int rp_bn_stats(const void* x, long long R, int C, double* partial, float* mean, float* rstd, float* running_mean,
                float* running_var, float momentum, float eps, int bf16, void* stream);
int rp_bn_stats_from_partials(const double* partial, int nblk, long long R, int C, const float* pivot, float* mean, float* rstd,
                              float* running_mean, float* running_var, float momentum, float eps, void* stream);
int rp_bn_apply_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta,
                    const void* residual, void* y, long long R, int C, int relu, int bf16, void* stream);
int rp_bn_bwd(const void* dy, const void* y, const void* x, const float* mean, const float* rstd, const float* gamma,
              const float* beta, void* dx, void* dres, float* dgamma, float* dbeta, double* partial, float* c12, long long R, int C, int relu,
              int training, int bf16, void* stream);
/* rp_bn_bwd's second and third pass for a BatchNorm whose masked incoming gradient g [R,C] fp32 and column-sum partials ([nblk][2][C]
 * doubles: sums of g and of g * xhat over disjoint row sets) were produced elsewhere (RpBnMask epilogue of the convolution kernels):
 * dgamma, dbeta, and dx = gamma * rstd * (g - mean(g) - xhat * mean(g * xhat)).  c12: [2 C] floats of scratch. */
int rp_bn_bwd_from_partials(const float* g, const float* x, const float* mean, const float* rstd, const float* gamma, const double* partial,
                            int nblk, float* dx, float* dgamma, float* dbeta, float* c12, long long R, int C, void* stream);

/* 3x3 / stride 2 / pad 1 max-pool (torchvision resnet.maxpool as driven by src/model.py:130), channels-last:
 * x [N,H,W,C] -> y [N,OH,OW,C], OH = (H-1)/2+1; idx (bytes, same shape as y) = window position 0..8 of the first maximum in scan
 * order (PyTorch's tie rule); backward gathers dy through idx into dx [N,H,W,C] (no atomics). */
int rp_maxpool3x3s2_fwd(const void* x, void* y, unsigned char* idx, int N, int H, int W, int C, int bf16, void* stream);
int rp_maxpool3x3s2_bwd(const void* dy, const unsigned char* idx, void* dx, int N, int H, int W, int C, int bf16, void* stream);

/* The stem convolution (torchvision resnet.conv1: 7x7, stride 2, pad 3, 3 -> 64, no bias; src/model.py:127), forward, hand-written
 * implicit GEMM (csrc/conv_stem.hip).  x_padded [N, H+6, W+6, 3]: the channels-last image inside a 3-pixel zero frame; w [64,7,7,3]
 * (the memory of a channels-last nn.Conv2d weight); y [N, (H-1)/2+1, (W-1)/2+1, 64].  W must be even (the pad taps of the last
 * window stay inside its row).  stats (optional) [rp_conv_stem_blocks(N,H,W)][2][64] doubles: per-workgroup sums of y and y^2 per
 * channel -- the following BatchNorm's batch statistics without a pass over y: rp_bn_stats_from_partials(stats, blocks, N*OH*OW, 64,
 * zeros, ...) finishes them (any partial sums of (x - pivot), (x - pivot)^2 over disjoint row sets are accepted). */
int rp_conv_stem_blocks(int N, int H, int W);
int rp_conv_stem_fwd(const float* x_padded, const float* w, float* y, double* stats, int N, int H, int W, void* stream);

/* The stem convolution of the bf16 configuration (csrc/conv_stem_bf16.hip): same operands as rp_conv_stem_fwd -- the fp32 image inside its
 * zero frame and the fp32 filter are rounded to bf16 on chip (v_mfma_f32_16x16x32_bf16, fp32 accumulate) -- y [N,OH,OW,64] bf16; stats as
 * above, of the stored bf16 values. */
int rp_conv_stem_bf16_blocks(int N, int H, int W);
int rp_conv_stem_fwd_bf16(const float* x_padded, const float* w, void* y, double* stats, int N, int H, int W, void* stream);
/* ... and its weight gradient (csrc/conv_stem_wgrad_bf16.hip): dw [64][7][7][3] fp32 from the same framed fp32 image and dY [N,112,112,64]
 * bf16 (H = W = 224 only): space-to-depth of the image into the workspace (bf16), an output-stationary MFMA stream over the pixels, a
 * fixed-order reduce of the per-workgroup partials (three launches, deterministic). */
size_t rp_conv_stem_wgrad_workspace_bytes(int N);
int rp_conv_stem_wgrad_bf16(const float* x_padded, const void* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H, int W,
                            void* stream);

/* The stem's weight gradient in EXACT fp32 (the headline configuration; autograd of src/model.py:127): x_padded [N,230,230,3] the framed
 * fp32 image (rp_preprocess_padded), dy [N,112,112,64] fp32 NHWC, dw [64][7][7][3] fp32 (= a [64,3,7,7] channels-last parameter
 * gradient); 224 x 224 images only.  workspace: rp_conv_stem_wgrad_f32_workspace_bytes(N) (the space-to-depth image + per-workgroup
 * partials, summed in a fixed order: deterministic).  Output-stationary on v_mfma_f32_32x32x2_f32 (csrc/conv_stem_wgrad_f32.hip);
 * replaces MIOpen's backward-weights call for this shape. */
size_t rp_conv_stem_wgrad_f32_workspace_bytes(int N);
int rp_conv_stem_wgrad_f32(const float* x_padded, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H, int W,
                           void* stream);

/* The 3x3 / stride 1 / pad 1, 64 -> 64 convolutions of resnet.layer1 in the bf16 configuration (src/model.py:131; torchvision
 * BasicBlock.conv1 / conv2), hand-written implicit GEMM with the input halo resident in LDS and the filter in registers
 * (csrc/conv3x3_bf16.hip).  x, y [N,56,56,64] bf16 channels-last; w [64][3][3][64] bf16 (the memory of a channels-last nn.Conv2d
 * weight; for the input gradient pass the rotated, transposed filter w'[ci][r][s][co] = w[co][2-r][2-s][ci] and dY as x).
 * scale / shift (optional, both or neither) [64] fp32: the input is max(0, x * scale + shift) -- the producing layer's BatchNorm-apply
 * + ReLU folded into the operand load; padding stays 0.  stats (optional) [rp_conv3x3_c64_blocks(N)][2][64] doubles: per-workgroup
 * sums of the stored y and y^2 per channel (for rp_bn_stats_from_partials, like rp_conv_stem_fwd).  H = W = 56 only. */
int rp_conv3x3_c64_blocks(int N);
int rp_conv3x3_c64_bf16(const void* x, const void* w, void* y, const float* scale, const float* shift, double* stats, int N, int H, int W,
                        void* stream);
/* ... and their weight gradient (csrc/conv3x3_wgrad_bf16.hip): dw [64][3][3][64] bf16 (channels-last weight memory) = sum over pixels of
 * dY[n,y,x,co] * X[n,y+r-1,x+s-1,ci], output-stationary over the pixel stream; workspace: rp_conv3x3_c64_wgrad_workspace_bytes(N) bytes
 * of per-workgroup fp32 partials, summed in a fixed order by the call's second launch (deterministic). */
int rp_conv3x3_c64_wgrad_blocks(int N);
size_t rp_conv3x3_c64_wgrad_workspace_bytes(int N);
int rp_conv3x3_c64_wgrad_bf16(const void* x, const void* dy, void* dw, void* workspace, size_t workspace_bytes, int N, int H, int W,
                              void* stream);

/* The same weight gradient in EXACT fp32 (the headline configuration; autograd of src/model.py:131's BasicBlock convolutions): x and dy
 * [N,56,56,64] fp32 NHWC, dw [64][3][3][64] fp32 (= a [64,64,3,3] channels-last parameter gradient), workspace
 * rp_conv3x3_c64_wgrad_f32_workspace_bytes(N) bytes of per-workgroup fp32 partials, summed in a fixed order (deterministic, no atomics).
 * Output-stationary on v_mfma_f32_32x32x2_f32 (csrc/conv3x3_wgrad_f32.hip); replaces MIOpen's backward-weights call for this shape. */
int rp_conv3x3_c64_wgrad_f32_blocks(int N);
size_t rp_conv3x3_c64_wgrad_f32_workspace_bytes(int N);
int rp_conv3x3_c64_wgrad_f32(const float* x, const float* dy, float* dw, void* workspace, size_t workspace_bytes, int N, int H, int W,
                             void* stream);
/* The same convolution's FORWARD in exact fp32 (csrc/conv3x3_f32.hip): y [N,56,56,64] = conv3x3(x [N,56,56,64], w [64 co][3][3][64 ci]), stride 1,
 * pad 1, NHWC memory, w = the memory of a channels-last [64,64,3,3] weight (src/model.py:131's BasicBlock convolutions, which the reference
 * runs through cuDNN).  The filter lives in registers (a wave owns 16 output channels: 144 VGPRs of A operands of
 * v_mfma_f32_16x16x4_f32), the input rows in a padded LDS ring read by one conflict-free ds_read_b32 per MFMA; one persistent
 * workgroup per CU (rp_conv3x3_c64_f32_blocks).  input_gradient != 0: x is dY and y is dX of the same convolution -- the filter
 * w'[ci][r][s][co] = w[co][2 - r][2 - s][ci] is read out of the forward weight w (no rotated copy). */
/* BatchNorm-backward epilogue of rp_conv3x3_c64_f32 (input_gradient launches): the convolution's result is the gradient of
 * a = relu(batch_norm(x)) (torchvision BasicBlock: out = relu(bn1(conv1(.))) feeds conv2); with `bn` given the kernel masks it,
 * g = result * (fma(x - mean, rstd * gamma, beta) > 0) -- bit for bit rp_bn_apply_fwd's expression --, stores g instead, and `stats` receives
 * the per-workgroup sums of g and g * (x - mean) * rstd: the column sums rp_bn_bwd's first pass would read dy and x again for.
 * rp_bn_bwd_from_partials then finishes that BatchNorm's backward from g and the partials (second + third pass only). */
typedef struct RpBnMask {
  const float* x;        /* the BatchNorm's INPUT, same shape as the convolution's result */
  const float* mean;
  const float* rstd;
  const float* gamma;
  const float* beta;
} RpBnMask;
int rp_conv3x3_c64_f32_blocks(int N);
int rp_conv3x3_c64_f32(const float* x, const float* w, float* y, double* stats, const float* res, const RpBnMask* bn, int N, int H, int W,
                       int input_gradient, void* stream);
/* The 3x3 / stride 1 / pad 1 convolutions with 128 INPUT channels on 28 x 28 maps in exact fp32 (csrc/conv3x3_c128_f32.hip): resnet.layer2's
 * 128 -> 128 convolutions (reference src/model.py:132, torchvision BasicBlock.conv1/conv2 through cuDNN), forward and input gradient, and the
 * forward of extractor_final_conv.conv1, 128 -> 192 with bias (src/modules/extractor.py:9,51).  y [N,28,28,CO] = bias + conv3x3(x [N,28,28,128],
 * w [CO][3][3][128]) in NHWC memory (w = the memory of a channels-last [CO,128,3,3] weight), CO = 128 or 192, bias [CO] or NULL.  Filter in
 * registers (a wave owns 16 output channels: 288 VGPRs of 16x16x4 A operands), CO / 64 channel groups of workgroups per tile of four image
 * rows, padded 6-slot LDS row ring.  stats (both kernels): NULL, or [blocks / channel groups][2][CO] doubles that receive per-workgroup sums of y
 * and y^2 per output channel -- the BatchNorm batch statistics of the output, finished by rp_bn_stats_from_partials with a zero pivot, so
 * the statistics pass over y is not needed (reference: nn.BatchNorm2d behind every one of these convolutions).  res (rp_conv3x3_c64_f32): NULL,
 * or a tensor of y's shape that is added to the result in the epilogue -- used for the input gradient of a BasicBlock's first convolution,
 * where autograd would add the gradient arriving over the identity path (torchvision BasicBlock: out += identity) in a pass of its own.  (The
 * 128-channel kernel has no registers to request a second epilogue operand a tile early, and exposed loads of it cost what the saved passes
 * over its four-times-smaller maps would bring: measured break-even, profiles/r6_ab.txt -- it keeps the statistics epilogue only.)  input_gradient != 0 (CO == 128, no bias): x is dY and y is dX of the convolution whose FORWARD weight is w. */
int rp_conv3x3_c128_f32_blocks(int N, int CO);
int rp_conv3x3_c128_f32(const float* x, const float* w, const float* bias, float* y, double* stats, int N, int H, int W, int CO,
                        int input_gradient, void* stream);

/* The stem's BatchNorm -> ReLU -> MaxPool2d(3, 2, 1) chain (src/model.py:127-130 on torchvision's resnet.bn1 / relu / maxpool) without
 * the [N,H,W,C] intermediates.  Forward (after rp_bn_stats, or with the running statistics in eval): y [N,OH,OW,C], idx = window
 * position (0..8) of the first maximum, bit-identical to rp_bn_apply_fwd(relu) + rp_maxpool3x3s2_fwd.  Backward: dp = gradient of y,
 * dx = gradient of the BatchNorm input, dgamma / dbeta; equal to rp_maxpool3x3s2_bwd + rp_bn_bwd up to fp32 summation order (the
 * column sums run window-major over dp, the dx pass gathers the pool gradient on the fly; the pool-backward tensor never exists).  partial: rp_bn_partial_blocks(N*H*W) * 2 * C doubles; c12: 2 * C floats. */
int rp_bn_relu_pool_fwd(const void* x, const float* mean, const float* rstd, const float* gamma, const float* beta, void* y,
                        unsigned char* idx, int N, int H, int W, int C, int bf16, void* stream);
int rp_bn_relu_pool_bwd(const void* dp, const unsigned char* idx, const void* x, const float* mean, const float* rstd,
                        const float* gamma, const float* beta, void* dx, float* dgamma, float* dbeta, double* partial, float* c12,
                        int N, int H, int W, int C, int training, int bf16, void* stream);

/* Geodesic pose loss of the training step (reference src/geom/losses.py:3-21; SE(3) arithmetic as restated in
 * rel_pose_amd/se3.py since lietorch is not vendored): Ps, Gs [B,2,7] (t, q xyzw);
 *   losses[0] = mean_{b,j} |tau|, losses[1] = mean_{b,j} |phi| of log(dG_j dP_j^-1), dG_j = G[1-j] G[j]^-1, dP_j likewise;
 *   dmean[m][b][k] = d losses[m] / d Gs[b].flat[k]  (m < 2, k < 14; exact forward-mode derivatives of the same arithmetic).
 * scratch: 60*B floats. */
int rp_geodesic_loss(const float* Ps, const float* Gs, float* losses, float* dmean, float* scratch, int B, void* stream);

/* LayerNorm over the last dim C (multiple of 64, <= 512), eps as given (reference uses 1e-6,
 * vision_transformer.py:396).  Saves mean / rstd per row for the backward. */
int rp_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                     int rows, int C, float eps, void* stream);
/* dx = LN'(dy) (+ add if non-null); partial sums are written to dgamma_part as [nblk][np][C], np = 2 (dgamma row, dbeta row
 * per block) or, when add != NULL, np = 3 with a third row = column sums of `add` (the bias gradient of the Linear whose
 * output gradient `add` is -- read here anyway); dbeta_part is ignored.  nblk = rp_layernorm_bwd_blocks(rows) <= 2048: one
 * rp_colsum over [nblk, np*C] finishes all of them */
int rp_layernorm_bwd_blocks(int rows);
int rp_layernorm_bwd(const float* dy, const float* x, const float* gamma, const float* mean, const float* rstd,
                     const float* add, float* dx, float* dgamma_part, float* dbeta_part, int rows, int C, void* stream);

/* out[c] = sum_r in[r][c]  (two-stage, fixed order).  workspace: rp_colsum_workspace_bytes(rows, cols). */
size_t rp_colsum_workspace_bytes(int rows, int cols);
int rp_colsum(const float* in, int rows, int cols, int ld, float* out, float* workspace, size_t workspace_bytes, void* stream);
/* up to RP_COLSUM_MAX independent column sums in ONE stage-1 and ONE stage-2 launch (same arithmetic and order per task as
 * rp_colsum): the bias / LayerNorm-affine gradients that end the backward of one reference module (autograd of
 * vision_transformer.py:349-354 produces them one reduction at a time) */
#define RP_COLSUM_MAX 8
typedef struct RpColsumTask {
  const float* in;
  int rows, cols, ld;
  float* out;
} RpColsumTask;
size_t rp_colsum_multi_workspace_bytes(const RpColsumTask* tasks, int n);
int rp_colsum_multi(const RpColsumTask* tasks, int n, float* workspace, size_t workspace_bytes, void* stream);

/* Image preprocessing (reference src/model.py:115-118,124-125; bit-exact): BGR->RGB, /255, ImageNet mean/std, nearest
 * resize to 224x224.  images [Z,3,H,W] fp32 0..255 -> out: channels-last memory [Z,224,224,3] of a [Z,3,224,224] tensor. */
int rp_preprocess(const float* images, float* out, int Z, int H, int W, void* stream);
/* same, written inside a `pad`-pixel zero frame: out [Z, 224 + 2 pad, 224 + 2 pad, 3] (pad = 3: the input rp_conv_stem_fwd takes) */
int rp_preprocess_padded(const float* images, float* out, int Z, int H, int W, int pad, void* stream);

/* Token layout + learned position embedding: x[z][n][c] = feat[z][c][n] + pos_embed[n][c]
 * (reference src/model.py:136-141,170-171; index part bit-exact).  feat is the CNN map [Z,C,N]. */
int rp_tokens_fwd(const float* feat, const float* pos_embed, float* x, int Z, int C, int N, void* stream);
/* same for a channels-last CNN map (memory already [Z,N,C]): x = feat + pos_embed; its backward is the identity */
int rp_tokens_fwd_nhwc(const float* feat, const float* pos_embed, float* x, int Z, int C, int N, void* stream);
int rp_tokens_bwd(const float* dx, float* dfeat, int Z, int C, int N, void* stream);

/* Fused softmax attention, N=576 tokens, d=64, heads packed along columns (col = h*64 + e):
 *   O[z][i][h*64+:] = softmax_j(scale * q_i . k_j) v_j        (vision_transformer.py:325-329)
 * q/k/v rows for image z: base + (z ^ xor)*576*ld + i*ld, with xor = q_xor for q, bit 0 of k_xor for k and bit 1
 * of k_xor for v (k_xor = 3: keys AND values from the partner image of the pair = the --noess cross attention,
 * vision_transformer.py:239-262; k_xor = 1 with stats_only: the EMM's S = q k_partner^T).  lse[z][h][i] = log sum_j exp(s_ij) saved
 * for the backward.  stats_only != 0: only lse is produced (v, o ignored) -- used for the row and
 * column normalisers of the dual softmax (vision_transformer.py:205-206).
 * bf16 (every attention / EMM entry point that takes it): 0 = exact fp32 MFMA operands (default, the parity path);
 * 1 = the QK^T / PV / dS contractions run on v_mfma_f32_32x32x16_bf16 -- operands (q, k, v, P, dS, dO, X, W) rounded to bf16
 * when they are formed, fp32 accumulation, fp32 softmax state, fp32 inputs and outputs: the "MFMA bf16 attention GEMMs" of
 * BASELINE.json configs[4]. */
int rp_attn_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int Z, int H, int ldq, int ldk,
                int ldv, int ldo, int q_xor, int k_xor, float scale, int stats_only, int bf16, void* stream);
/* The same operation on the bf16 DATA PATH of BASELINE.json configs[4] (csrc/attention_bf16.hip): q / k / v / o are BF16 in memory
 * (same packing: col = h*64 + e, row strides ld* in ELEMENTS, multiples of 8); lse is fp32 in LOG2 units: lse2[z][h][i] =
 * log2 sum_j exp2(scale log2(e) q_i . k_j) (= natural-log lse / ln 2), which is what rp_attn_bwd_bf16 consumes.  K / V tiles reach LDS by LDS-DMA,
 * the matrix pipe is fed by ds_read_b128 / ds_read_b64_tr_b16 with no conversion instruction in the loop, P is packed once per tile;
 * fp32 accumulation and softmax state.  xor / stats_only as rp_attn_fwd.  Replaces vision_transformer.py:325-329 (and, stats_only,
 * the normalisers of :205-206) when the producing Linear (rp_linear_rows192 with io_bf16 bit 1) writes bf16. */
int rp_attn_fwd_bf16(const void* q, const void* k, const void* v, void* o, float* lse, int Z, int H, int ldq, int ldk, int ldv,
                     int ldo, int q_xor, int k_xor, float scale, int stats_only, void* stream);
/* Autograd of the above on the same data path, RECOMPUTE form (no stored dS): delta[z][h][i] = sum_e dO O (fp32, from bf16 rows), then
 * two deterministic kernels -- dK / dV (a wave owns 32 keys and streams Q / dO tiles) and dQ (a wave owns 32 queries and streams K / V
 * tiles) -- that rebuild P = exp2(s - lse2) from the forward's lse2.  q / k / v / dout / dq / dk / dv are BF16 (strides in elements);
 * kv_xor = 1: keys / values of problem z live at image z ^ 1 (backward of k_xor = 3); *_colpart (all NULL, or dk and dv together, dq
 * optional): [Z*18][ldp] fp32 column sums of dq / dk / dv over each 32-token block (pointers at the first of the H*64 columns),
 * taken from the fp32 accumulators -- the qkv bias gradient's partials (vision_transformer.py:323). */
int rp_attn_bwd_delta_bf16(const void* dout, const void* o, float* delta, int Z, int H, int ld, void* stream);
int rp_attn_bwd_bf16(const void* q, const void* k, const void* v, const void* dout, const float* lse2, const float* delta, void* dq,
                     void* dk, void* dv, int Z, int H, int ldq, int ldk, int ldv, int lddo, int lddq, int lddk, int lddv, float scale,
                     int kv_xor, float* dq_colpart, float* dk_colpart, float* dv_colpart, int ldp, void* stream);
/* The dual softmax's two normalisers (vision_transformer.py:205-206) of the EMM score matrix S_z = scale * q_{z^1} k_z^T (queries of the
 * partner image, keys of image z; q / k point at the first of the H*64 columns, rows (z*576 + i)*ld):
 *   rlse[z][h][i] = log sum_j exp(S_z[i][j]),   clse[z][h][j] = log sum_i exp(S_z[i][j])      ([Z,H,576] floats each)
 * as ONE pass over S (bf16 = 0): the rows online, the columns from per-32-row-block (max, sum) partials kept in `workspace`
 * (rp_emm_stats_workspace_bytes(Z, H) bytes) and combined by a second, tiny launch.  bf16 != 0: two rp_attn_fwd(stats_only) passes
 * (the reductions would cost that mode more than the second pass saves); workspace may then be NULL.  Z must be even. */
/* s_out (bf16 = 0 only; NULL = off): [Z][H][18 query blocks][18 key tiles][1024] -- the score tiles themselves, in log2 units (scale
 * log2(e) q.k), element (query i, key j) of a tile at float ((j >> 2) * 32 + i) * 4 + (j & 3) (four contiguous 16-byte stores per lane from
 * the accumulators).  With 288 GB of HBM the three later passes over S (rp_emm_apply forward and swap, rp_emm_grad_ds; `s_in` there)
 * read these 4 MB per image instead of recomputing q k^T: 96 of their 260 MFMAs per tile disappear. */
size_t rp_emm_stats_workspace_bytes(int Z, int H);
int rp_emm_stats(const float* q, const float* k, float* rlse, float* clse, void* workspace, float* s_out, int Z, int H, int ldq, int ldk,
                 float scale, int bf16, void* stream);
/* delta[z][h][i] = sum_e dO[z][i][h*64+e] * O[z][i][h*64+e] */
int rp_attn_bwd_delta(const float* dout, const float* o, float* delta, int Z, int H, int ld, void* stream);
/* dq, dk, dv of the above (recompute-based, two deterministic passes) */
int rp_attn_bwd(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* delta,
                float* dq, float* dk, float* dv, int Z, int H, int ldq, int ldk, int ldv, int lddo, int lddq, int lddk,
                int lddv, float scale, int bf16, void* stream);
/* same with the keys/values of problem z taken from image z ^ kv_xor (dk, dv are written at the rows of the image the
 * keys/values came from): backward of rp_attn_fwd(..., q_xor=0, k_xor=3, ...) when kv_xor = 1 */
int rp_attn_bwd_cross(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                      const float* delta, float* dq, float* dk, float* dv, int Z, int H, int ldq, int ldk, int ldv, int lddo,
                      int lddq, int lddk, int lddv, float scale, int kv_xor, int bf16, void* stream);
/* the two passes of rp_attn_bwd separately (they are independent; the host overlaps them on two HIP streams) */
int rp_attn_bwd_dkdv(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                     const float* delta, float* dk, float* dv, int Z, int H, int ldq, int ldk, int ldv, int lddo, int lddk,
                     int lddv, float scale, int bf16, void* stream);
/* dK/dV pass that also stores scale * dS ([Z,H,576,576] floats) TILED: ds[z][h][i >> 5][j >> 5][r][lane] with, inside a 32x32 tile,
 * i & 31 = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) and j & 31 = lane & 31 (the producing wave's MFMA accumulator image: 4 KB contiguous per
 * tile, coalesced 256-byte stores).  dQ = ds K is then ONE rp_ds_matmul instead of the dQ pass, which would recompute S and dP
 * (5 executed GEMMs instead of 7).  bf16 != 0 (the bf16 configuration): the tiles are stored as BF16 in the same image (2 KB per
 * tile, ds then holds Z*H*576*576 bf16) -- pass ds_bf16 = 1 to rp_ds_matmul.
 * dk_colpart / dv_colpart (both or neither; NULL = off): [Z*18][ldp] arrays (pointers at the first of the H*64 columns) that receive the
 * column sums of dk / dv over each block of 32 token rows -- summed over the Z*18 rows they are the k / v thirds of the qkv bias
 * gradient (vision_transformer.py:323), which then needs no pass over the [tokens, 576] gradient; rp_ds_matmul's `colpart` is the
 * same for its output (the q third). */
int rp_attn_bwd_dkdv_ds(const float* q, const float* k, const float* v, const float* dout, const float* lse,
                        const float* delta, float* dk, float* dv, float* ds, int Z, int H, int ldq, int ldk, int ldv, int lddo,
                        int lddk, int lddv, float scale, int bf16, float* dk_colpart, float* dv_colpart, int ldp, void* stream);
/* out[z][i][h*64 + d] = sum_j ds[z*H + h][i][j] * b[z ^ b_xor][j][h*64 + d] for the [Z,H,576,576] array a stored-dS pass wrote: the
 * dQ = dS K half of Attention's autograd (vision_transformer.py:325-329) after rp_attn_bwd_dkdv_ds, and with b_xor = 1 the
 * dK = dS^T-major x Q(partner image) half of the EMM's (:198-206) after rp_emm_grad_ds.  b / out point at the first of the H*64
 * columns (row strides ldb / ldo floats, 576 rows per image); one launch for all Z*H problems, dS streamed once from memory.
 * ds_bf16: 0 = fp32 tiles, exact fp32 MFMA; 1 = bf16 tiles (what the producers write when their bf16 flag is set): the product
 * runs on v_mfma_f32_32x32x16_bf16 with b rounded to bf16 on chip, fp32 accumulate and output. */
int rp_ds_matmul(const float* ds, const float* b, float* out, int Z, int H, int ldb, int ldo, int b_xor, int ds_bf16, float* colpart,
                 int ldp, void* stream);
int rp_attn_bwd_dq(const float* q, const float* k, const float* v, const float* dout, const float* lse, const float* delta,
                   float* dq, int Z, int H, int ldq, int ldk, int ldv, int lddo, int lddq, float scale, int bf16, void* stream);
/* STORED-P form of Attention's autograd (vision_transformer.py:325-329), exact fp32: with 288 GB of HBM the training forward can
 * afford to keep the probabilities, and the backward then executes exactly its four algorithmic products (dP = dO V^T, dV = P^T dO,
 * dK = dS^T Q, dQ = dS K) -- no Q K^T recompute, no exponential.
 * rp_attn_fwd_savep = rp_attn_fwd(q_xor = k_xor = 0) that also writes
 *   pst  [Z][H][18 query blocks][18 key tiles][1024]: exp2(s_ij - m_t(i)) of each 32 x 32 tile, s in log2 units, m_t(i) the online
 *        softmax's running maximum of query i after key tile t; inside a tile element (query i, key j) is float
 *        ((j >> 2) * 32 + i) * 4 + (j & 3): the 16-byte runs the forward's lanes hold, 1 KB contiguous per store instruction;
 *   mrun [Z][H][18 key tiles][576]: m_t(i).
 * Neither costs the forward an LDS round trip or a VALU instruction: four 16-byte stores per lane straight from the accumulators.
 * rp_attn_bwd_dkdv_p: the dK/dV pass over them (P = pst * exp2(mrun - lse / ln 2)); writes dk, dv (colpart arguments as
 * rp_attn_bwd_dkdv_ds) and scale * dS as ds [Z][H][18 query blocks][18 key blocks][1024], element (query i, key j) of a tile at float
 * ((i >> 2) * 32 + j) * 4 + (i & 3) (again four contiguous 16-byte stores per lane from the accumulators).
 * rp_ds_matmul_t: rp_ds_matmul for that tile layout (dQ = dS K): both operands by LDS-DMA, exact fp32 MFMA. */
int rp_attn_fwd_savep(const float* q, const float* k, const float* v, float* o, float* lse, float* pst, float* mrun, int Z, int H,
                      int ldq, int ldk, int ldv, int ldo, float scale, void* stream);
int rp_attn_bwd_dkdv_p(const float* q, const float* v, const float* dout, const float* lse, const float* delta, const float* pst,
                       const float* mrun, float* dk, float* dv, float* ds, int Z, int H, int ldq, int ldv, int lddo, int lddk,
                       int lddv, float scale, float* dk_colpart, float* dv_colpart, int ldp, void* stream);
int rp_ds_matmul_t(const float* ds, const float* b, float* out, int Z, int H, int ldb, int ldo, int b_xor, float* colpart, int ldp,
                   void* stream);

/* Quadratic positional features (closed form of get_positional_encodings, vision_transformer.py:90-158):
 * pos[b][n] = (p3^2, p4^2, p3 p4, p3, p4, 1), p3 = lin[n%24]*iy_b, p4 = lin[n/24]*ix_b,
 * ix = 1/((fx/(2cx))*2), iy = 1/((fy/(2cy))*2) from intrinsics[b][0] (fx,fy,cx,cy); intrinsics == NULL -> ix=iy=1. */
/* l1 != 0: get_l1_positional_encodings (vision_transformer.py:37-87): (1, 1, 1, p3, p4, 1) */
int rp_posenc(const float* intrinsics, const float* lin24, float* pos, int B, int l1, void* stream);

/* Essential Matrix Module (CrossAttention ess branch, vision_transformer.py:198-223), per image z, head h:
 *   S = scale * q_{z^1} k_z^T ; A = rowsoftmax(S) o colsoftmax(S) = exp(2S - rlse_i - clse_j)
 *   X_z = [v_z | pos | 0] (576 x 96, 70 live columns);  T = A X_z ;  F = X_z^T T  (70x70 live)
 * rp_emm_build_x: gathers v from qkv (col 384 + h*64) and pos[z/2] into x[z][h][576][96].
 * rp_emm_apply : T (optional store, [Z][H][576][96]) and per-workgroup partial F ([Z][H][6][96][96]).
 *                swap != 0 exchanges the roles of q and k / rlse and clse (gives A^T X: used by the backward).
 * rp_emm_finalize: g[z^1][c][h*70+a] = sum_wg Fpart[z][h][wg][a][c], zero-padded to ldg (=224) columns
 *                (the reshape/transpose of vision_transformer.py:229-230 plus the output flip of :238).
 */
int rp_emm_build_x(const float* qkv, const float* pos, float* x, int Z, int H, int ldqkv, void* stream);
int rp_emm_build_x_bwd(const float* dx, float* dqkv, int Z, int H, int ldqkv, void* stream); /* dqkv[:, 384+h*64+e] = dx[..][e] */
/* ablation flags of the reference that are runnable there:
 *   single != 0 : use_single_softmax (:201-203), A = softmax(S,-1) (clse unused);
 *   x_left != 0 : cross_features (:218-220), F_z = X_left[z^1]^T A_z X_z  (x_left indexed like x);
 *   s_in != 0 (bf16 = 0 only): the score tiles rp_emm_stats stored are read instead of q k^T being recomputed (qkv is then unused). */
int rp_emm_apply(const float* qkv, int ldqkv, const float* x, const float* x_left, const float* rlse, const float* clse,
                 const float* s_in, float* t_out, float* f_part, int Z, int H, float scale, int swap, int single, int bf16,
                 void* stream);
int rp_emm_finalize(const float* f_part, float* g, int Z, int H, int ldg, void* stream);
int rp_emm_finalize_bwd(const float* dg, float* df, int Z, int H, int ldg, void* stream); /* df[z][h][96][96] */
/* rowdot: out[r] = sum_c a[r][c]*b[r][c], C = 96 */
int rp_rowdot96(const float* a, const float* b, float* out, long long rows, void* stream);
/* EMM gradient pass: owner side o (rows i if swap==0, cols j if swap!=0):
 *   dS = 2 A dA - R rho_i - C gamma_j,  dA = W X^T (swap==0: W rows i) ;  d(owner operand) = scale * dS * other
 * writes dqkv q-columns of image z^1 (swap==0) or k-columns of image z (swap!=0). */
int rp_emm_grad(const float* qkv, int ldqkv, const float* x, const float* w, const float* rlse, const float* clse,
                const float* rho, const float* gamma, float* dqkv, int Z, int H, float scale, int swap, int single,
                int bf16, void* stream);
/* the owner = query pass (swap = 0) that also stores scale * dS_ij ([Z,H,576,576] floats) key index major and TILED like
 * rp_attn_bwd_dkdv_ds (rows = keys j, columns = queries i): the key-side gradient dk_z = ds_z q_{z^1} is then one
 * rp_ds_matmul(b_xor = 1) instead of the swap = 1 pass.  bf16 != 0: bf16 tiles, as for rp_attn_bwd_dkdv_ds */
/* s_in (NULL = recompute; bf16 = 0 only): rp_emm_stats' stored score tiles, as for rp_emm_apply */
int rp_emm_grad_ds(const float* qkv, int ldqkv, const float* x, const float* w, const float* rlse, const float* clse,
                   const float* rho, const float* gamma, const float* s_in, float* dqkv, float* ds, int Z, int H, float scale,
                   int single, int bf16, void* stream);

/* Weight gradient of a Linear on the bf16 data path (csrc/dw192_bf16.hip): slabs of C[n][k] = sum_m A[m][n] B[m][k] for A [M,N] BF16
 * (row stride lda elements, N a multiple of 192), B [M,192] contiguous, BF16 or (b_is_f32) fp32 rounded to bf16 on chip, M a multiple
 * of 64 -- the dW = dY^T X of vision_transformer.py:323,330 / vit_layers/mlp.py:22,24, where one operand is always 192 wide.  A
 * streaming kernel (LDS-DMA stages, both operands by ds_read_b64_tr_b16, one [192 x 192] fp32 tile per workgroup over a slab of token
 * rows); it leaves rp_dw192_bf16_splits(M, N) split-K slabs [split][N][192] fp32 in `workspace`, which the caller finishes with
 * rp_splitk_reduce_multi (task {ws, C, M = N, N = 192, ldc, split_k, trans_c}: trans_c writes C^T, e.g. fc2's [192,768] weight from
 * A = h [M,768]) -- deterministic, and batchable with the other weight gradients of a block. */
int rp_dw192_f32_splits(int M, int N);
size_t rp_dw192_f32_workspace_bytes(int M, int N);
/* the same product in EXACT fp32 (csrc/dw192_f32.hip; A and B fp32, M a multiple of 32): output-stationary [192 x 192] tiles, one wave
 * per SIMD on v_mfma_f32_32x32x2_f32, LDS-DMA stages -- the parity path's weight gradients (replaces rp_gemm's split-K form for them) */
int rp_dw192_f32(const float* a, int lda, const float* b, int M, int N, void* workspace, size_t workspace_bytes, void* stream);
/* the same product, same arguments, same slabs (rp_dw192_f32_splits / _workspace_bytes), on the BF16 matrix pipe at fp32 grade
 * (csrc/dw192_split3.hip): each fp32 operand is split on chip into three round-to-nearest bf16 limbs (error-free: 3 x 8 bits = 24) and six
 * of the nine limb products are accumulated in fp32 on v_mfma_f32_32x32x16_bf16; the dropped terms are <= 2^-26 of each product.  OPT-IN
 * (RP_DW_SPLIT3=1): the default weight gradients stay on rp_dw192_f32's exact fp32 MFMAs.  An Inf operand gives NaN. */
int rp_dw192_split3(const float* a, int lda, const float* b, int M, int N, void* workspace, size_t workspace_bytes, void* stream);
int rp_dw192_bf16_splits(int M, int N);
size_t rp_dw192_bf16_workspace_bytes(int M, int N);
int rp_dw192_bf16(const void* a, int lda, const void* b, int b_is_f32, int M, int N, void* workspace, size_t workspace_bytes, void* stream);

/* Input gradient of the qkv Linear with the LayerNorm backward behind it, on the bf16 data path (csrc/dx_lnbwd_bf16.hip):
 *   dx = LayerNorm'(dY W; x, gamma, mean, rstd) (+ add)   for dY [M,576] BF16, wt = W^T [192][576] BF16 (host copy of the fp32 master),
 *   x / add / dx [M,192] fp32 (vision_transformer.py:323,352 and their autograd).  Output-resident: a wave keeps 16 whole 192-wide
 *   output rows in its accumulators, dY is read once as MFMA operands, the gradient of the LayerNorm output never reaches memory.
 *   part [ceil(M / rp_dx_lnbwd_bf16_tile_rows())][np][192] (np = 3 with add, else 2) receives the per-tile column sums of dxn o xhat
 *   (dgamma), dxn (dbeta) and add (the bias gradient of the Linear that produced it); the caller column-sums them. */
int rp_dx_lnbwd_bf16_tile_rows(void);
int rp_dx_lnbwd_bf16(const void* dy, const void* wt, const float* x, const float* gamma, const float* mean, const float* rstd,
                     const float* add, float* dx, float* part, int M, int K, void* stream);

/* rp_emm_finalize for f_part [Z][H][nparts][96][96] (nparts = 6: rp_emm_apply's workgroup partials; 1: rp_emm_f_bf16's whole F) */
int rp_emm_finalize_parts(const float* f_part, float* g, int Z, int H, int ldg, int nparts, void* stream);

/* The Essential Matrix Module on the bf16 data path of BASELINE.json configs[4] (csrc/emm_bf16.hip; same algebra and indexing as the
 * fp32 entry points above, vision_transformer.py:198-223 and its autograd): qkv [Z*576, ldqkv] BF16, X / T / U / W / W' [Z][H][576][96]
 * BF16, the normalisers rlse2 / clse2 [Z][H][576] fp32 in LOG2 units (rp_attn_fwd_bf16(stats_only): rlse2 with q_xor = 1, clse2 with
 * the k pointer as q, the q pointer as k and k_xor = 1), rho / gamma [Z][H][576] and dF [Z][H][96][96] fp32.  Default flags only (dual
 * softmax, F = X^T A X): the ablation variants keep the fp32-storage kernels.
 *   build_x : X = [v | pos | 0]                        apply : T = A X (swap = 0) or U = A^T X (swap = 1)
 *   f       : F = X^T T  -> f [Z][H][96][96] fp32 (then rp_emm_finalize_parts(nparts = 1))
 *   w       : W = X dF, W' = X dF^T, rho = <W, T> (from the fp32 accumulators of W)
 *   dx      : dX = T dF^T + U dF -> the v columns of dqkv (bf16), gamma = <W', U>
 *   grad    : dq (swap = 0: x = X, w = W, writes the q columns of image z^1) / dk (swap = 1: w = W', the k columns of image z);
 *             recompute form: S and dA are rebuilt per tile, no dS is stored */
int rp_emm_build_x_bf16(const void* qkv, const float* pos, void* x, int Z, int H, int ldqkv, void* stream);
int rp_emm_apply_bf16(const void* qkv, int ldqkv, const void* x, const float* rlse2, const float* clse2, void* t_out, int Z, int H,
                      float scale, int swap, void* stream);
int rp_emm_f_bf16(const void* x, const void* t, float* f, int Z, int H, void* stream);
int rp_emm_w_bf16(const void* x, const void* t, const float* df, void* w, void* wp, float* rho, int Z, int H, void* stream);
int rp_emm_dx_bf16(const void* t, const void* u, const void* wp, const float* df, void* dqkv, int ldqkv, float* gamma, int Z, int H,
                   void* stream);
int rp_emm_grad_bf16(const void* qkv, int ldqkv, const void* x, const void* w, const float* rlse2, const float* clse2, const float* rho,
                     const float* gamma, void* dqkv, int Z, int H, float scale, int swap, void* stream);

/* q / max(|q|, 0.01), slot 0 <- Gs  (normalize_preds, src/model.py:145-159) */
int rp_pose_normalize_fwd(const float* pred, const float* gs, float* out, int B, void* stream);
int rp_pose_normalize_bwd(const float* pred, const float* dout, float* dpred, int B, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Row-resident Linear for K = 192 (qkv, attention proj, fc1; vision_transformer.py:323,330,352-353, mlp.py:22-23), with the
 * LayerNorm that precedes it optionally fused in:
 *   y = act(LN?(x) W^T + bias) (+ residual),  x [M,192], W [N,192] (nn.Linear layout), N % 32 == 0, N <= 1024.
 * ln_gamma/ln_beta non-NULL: x is layer-normalised (eps) on the way into the MFMA operand registers; xn_out [M,192], mean_out
 * [M], rstd_out [M] (each optional) receive what the backward needs.  y_pre (optional) receives the pre-activation.
 * act: 0 none, 1 GELU.  No workspace.
 * Input-gradient use (dx = (dy W) o act'(aux) of a Linear whose OUTPUT width is 192: fc2, attention proj): x = dy, w = W^T
 * [N_in,192] contiguous, dact_aux = the saved pre-activation [M,N] (y is multiplied by GELU'(aux); NULL: none), colsum_part
 * [ceil(M / rp_linear_rows192_tile_rows()), N] (optional) receives the column sums of y per row tile -- summed, they are the
 * bias gradient of the layer below.
 * precision: 0 exact fp32 MFMA; 1 = the bf16 configuration (operands rounded to bf16, fp32 accumulate:
 * v_mfma_f32_16x16x32_bf16; LayerNorm, bias, GELU and the residual stay fp32) -- in this precision w points to a BF16 copy of
 * the weight ([N,192] bf16, contiguous; the caller refreshes it when the fp32 master changes), x stays fp32 and is rounded to
 * nearest even on chip.  io_bf16 (precision 1 only; RpGemm.io_bf16's bits): bit 0 = x holds BF16 rows (no LayerNorm: a lane's
 * six 8-element MFMA operands are loaded as they lie, e.g. the bf16 attention output into proj), bit 1 = y and y_pre are written as
 * bf16, bit 2 = dact_aux holds bf16, bit 3 = xn_out is written as bf16 (the rounded rows the MFMA consumed: what the bf16 weight-
 * gradient product reads).  The bits select the bf16 data path of BASELINE.json configs[4].
 * ------------------------------------------------------------------------------------------- */
int rp_linear_rows192_tile_rows(void);
int rp_linear_rows192(const float* x, const float* w, const float* bias, const float* residual, const float* ln_gamma,
                      const float* ln_beta, float eps, float* y, float* y_pre, float* xn_out, float* mean_out, float* rstd_out,
                      const float* dact_aux, float* colsum_part, int M, int N, int K, int act, int precision, int io_bf16,
                      void* stream);

/* ---------------------------------------------------------------------------------------------
 * Fused transformer MLP, inference path (vision_transformer.py:353 + vit_layers/mlp.py:20-26):
 *   y = x + fc2(GELU(fc1(LayerNorm(x; gamma, beta, eps)) + b1)) + b2,   x, y [M, dim], w1 [hidden, dim], w2 [dim, hidden]
 * (nn.Linear layouts).  The normalised rows and the hidden activation stay in registers / LDS.  dim = 192, hidden = 768 only
 * (RP_EBADSHAPE otherwise).  workspace: rp_mlp_fused_workspace_bytes(M) bytes (partial tiles of the stream-K split).
 * Training form: with xn_out [M,dim], mean_out [M], rstd_out [M], h_out [M,hidden] and hpre_out [M,hidden] (all five or none;
 * NULL = inference) the kernel also stores what the backward needs -- the normalised rows and their statistics, fc1's pre-activation
 * and the hidden activation -- on the way: the forward of the MLP is then one launch in training too.
 * precision: 0 exact fp32 MFMA; 1 = the bf16 configuration (v_mfma_f32_16x16x32_bf16, fp32 accumulate; LayerNorm, bias, GELU, residual
 * fp32): w1 and w2 then point to BF16 copies ([hidden,dim] and [dim,hidden]), w2 with the hidden units of every 32-chunk in the order
 * documented at rp_mlp_fused_bwd.  io_bf16 (precision 1, training form only): bit 1 = h_out and hpre_out are written as bf16,
 * bit 3 = xn_out is written as bf16 (the rounded rows the fc1 product consumed; what the bf16 weight-gradient kernel rp_dw192_bf16 reads).
 * bit 4 (training and inference forms) = w2 is stored CHUNK-MAJOR, [hidden / 32][dim][32] (every staged tile 12 KB contiguous) instead of
 * [dim][hidden].
 * ------------------------------------------------------------------------------------------- */
size_t rp_mlp_fused_workspace_bytes(int M);
int rp_mlp_fused_fwd(const float* x, const float* gamma, const float* beta, const float* w1, const float* b1, const float* w2,
                     const float* b2, float* y, void* workspace, int M, int dim, int hidden, float eps, float* xn_out, float* mean_out,
                     float* rstd_out, float* h_out, float* hpre_out, int precision, int io_bf16, void* stream);

/* Backward-data of the same MLP (training): dhp [M,hidden] = (dy W2) o GELU'(hpre) -- the gradient of fc1's pre-activation, which
 * fc1's weight gradient needs -- and dxn [M,dim] = dhp W1, the gradient of the LayerNorm output, as one kernel (dh never exists).
 * w2t = W2^T [hidden,dim] and w1t = W1^T [dim,hidden], contiguous.  colpart [ceil(M / rp_mlp_fused_bwd_tile_rows()), hidden]
 * receives the column sums of dhp per row tile (their sum is the fc1 bias gradient).
 * precision: 0 exact fp32 MFMA.  1 = the bf16 configuration (v_mfma_f32_16x16x32_bf16, fp32 accumulate; GELU', column sums and dxn fp32):
 * w2t and w1t then point to BF16 copies, w1t with the 32 hidden units of every chunk c stored in the order a lane's accumulators form the
 * MFMA operand: position 8 q + e of chunk c holds unit 32 c + 4 q + e (e < 4) or 32 c + 16 + 4 q + e - 4 (e >= 4), q = 0..3.  io_bf16
 * (precision 1 only): bit 1 = dhp is written as bf16, bit 2 = hpre holds bf16, bit 4 = w1t is stored chunk-major, [hidden / 32][dim][32]. */
size_t rp_mlp_fused_bwd_workspace_bytes(int M);
int rp_mlp_fused_bwd_tile_rows(void);
int rp_mlp_fused_bwd(const float* dy, const float* hpre, const float* w2t, const float* w1t, float* dhp, float* dxn, float* colpart,
                     void* workspace, int M, int dim, int hidden, int precision, int io_bf16, void* stream);
/* The same with the backward of the LayerNorm in front of fc1 folded into the epilogue (Block.forward, vision_transformer.py:353:
 * x + mlp(norm2(x)) -- the gradient of x is LayerNorm-backward(dxn) + dy): instead of dxn the kernel writes
 *     dx [M,dim] = rstd (g - mean_c(g) - xhat mean_c(g xhat)) + dy,   g = dxn o ln_gamma,  xhat = (ln_x - ln_mean) ln_rstd
 * (ln_x [M,dim] the LayerNorm's input, ln_mean / ln_rstd [M] its saved statistics; dx must not alias dy), and ln_part
 * [rp_mlp_fused_bwd_ln_part_rows(M)][3 dim] receives per row block the column sums of (dxn o xhat | dxn | dy): summed over the rows they
 * are dgamma, dbeta and the bias gradient of the Linear that produced dy (rp_layernorm_bwd's contract with `add` = dy).  dxn never
 * reaches HBM: one [M,dim] write + read and the re-read of dy saved per Block.  Tiles that no single workgroup finished are
 * completed -- LayerNorm backward included -- by a fix-up launch.  All other arguments as rp_mlp_fused_bwd. */
int rp_mlp_fused_bwd_ln_part_rows(int M);
int rp_mlp_fused_bwd_ln(const float* dy, const float* hpre, const float* w2t, const float* w1t, float* dhp, float* dx, float* colpart,
                        void* workspace, int M, int dim, int hidden, int precision, int io_bf16, const float* ln_x,
                        const float* ln_gamma, const float* ln_mean, const float* ln_rstd, float* ln_part, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Training-time augmentation of a resident batch (RGBDAugmentor, src/data_readers/augmentation.py:7-37):
 * ColorJitter(brightness, contrast, saturation, hue; per-pair order) + RandomGrayscale + nearest resize, one parameter row
 * per pair: params[b] = {order[4] (0 brightness, 1 contrast, 2 saturation, 3 hue), b, c, s, h, gray (0/1)} as 9 floats.
 * images: uint8 [B,2,H,W,3] BGR as decoded (cv2 / PIL convention of the readers); out: fp32 [B,2,3,Ho,Wo] BGR 0..255 = the
 * input layout of ViTEss.forward.  workspace: B * rp_augment_blocks() doubles.
 * ------------------------------------------------------------------------------------------- */
int rp_augment_blocks(void);
int rp_augment_pairs(const unsigned char* images, const float* params, float* out, double* workspace, int B, int H, int W,
                     int Ho, int Wo, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Essential-matrix auxiliary (BASELINE.json north_star: "per-pair 3x3 SVD as a one-warp Jacobi sweep with no MFMA").
 * NOT on ViTEss.forward's output path: the reference regresses R,t and never decomposes a matrix (src/model.py:91-98,
 * 145-159), so parity forbids an SVD there.  Offered for consumers of the predicted pose:
 *   rp_essential_from_pose: E[n][3][3] = [t]x R(q) from poses [n][7] = (t, q xyzw); q is normalised internally.
 *   rp_svd3x3: A[n][3][3] = U diag(S) V^T, S descending and non-negative, U and V orthogonal (U completed by cross
 *     products when A is rank deficient, as an essential matrix is).  One lane per matrix, one-sided Jacobi, registers only.
 * ------------------------------------------------------------------------------------------- */
int rp_essential_from_pose(const float* pose, float* E, int n, void* stream);
int rp_svd3x3(const float* A, float* U, float* S, float* V, int n, void* stream);
/* rp_pose_from_essential: the decode E -> (R, t) that north_star's "SVD ... produce relative R,t" names (no reference counterpart:
 * SURVEY.md row a16): per matrix the SVD above, the four candidates (U W V^T | U W^T V^T, +-u_2) and the cheirality vote over P point
 * correspondences x1[n][P][2] <-> x2[n][P][2] (normalised image coordinates, X2 = R X1 + t).  pose[n][7] = (t unit-norm, q xyzw,
 * w >= 0); count[n] (optional) = points in front of both cameras for the winning candidate.  One lane per matrix, registers only. */
int rp_pose_from_essential(const float* E, const float* x1, const float* x2, int P, float* pose, int* count, int n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* RELPOSE_HIP_H */
