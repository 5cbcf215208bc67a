// rows_probe.hip -- where a wave of linear_rows_kernel spends its cycles: shader-clock stamps around the DMA wait, the barrier,
// the MFMA loop and the epilogue of each chunk, for the first 8 workgroups (the product source compiled with RP_ROWS_PROBE).
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I../../rel_pose_amd/csrc -o rows_probe rows_probe.hip
#define RP_ROWS_PROBE 1
#include "../../rel_pose_amd/csrc/linear_rows.hip"
#include <stdio.h>
#include <vector>

static int G_ = 0;
int main(int argc, char** argv) {
  // usage: rows_probe [N] [ln 0/1] [bf16 config 0/1] [io_bf16 bits] [act+pre 0/1] [M]
  const int N = argc > 1 ? atoi(argv[1]) : 576, ln = argc > 2 ? atoi(argv[2]) : 1;
  const int bf = argc > 3 ? atoi(argv[3]) : 0, io = argc > 4 ? atoi(argv[4]) : 0, actpre = argc > 5 ? atoi(argv[5]) : 0;
  const int M = argc > 6 ? atoi(argv[6]) : 73728;
  float *x, *w, *b, *g, *y; long long* probe;
  hipMalloc(&x, (size_t)M * 192 * 4); hipMalloc(&w, (size_t)N * 192 * 4); hipMalloc(&b, N * 4); hipMalloc(&g, 192 * 4);
  hipMalloc(&y, (size_t)M * N * 4); hipMalloc(&probe, 8 * 64 * NW * 5 * 8);
  float* ypre = nullptr;
  if (actpre) hipMalloc(&ypre, (size_t)M * N * 4);
  std::vector<float> h((size_t)M * 192);
  unsigned s = 1u;
  for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 32768.0f - 1.0f; }
  hipMemcpy(x, h.data(), h.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(w, h.data(), (size_t)N * 192 * 4, hipMemcpyHostToDevice);
  hipMemcpy(b, h.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(g, h.data(), 192 * 4, hipMemcpyHostToDevice);
  RowsP p{x, w, b, nullptr, ln ? g : nullptr, ln ? g : nullptr, y, ypre, nullptr, nullptr, nullptr, nullptr, nullptr, M, N, 1e-6f, actpre,
          N / CH, (M + ROWS - 1) / ROWS, 0, 0, io, probe};
  const long long items = (long long)p.tiles * p.nchunk;
  const int slots = bf ? (ln ? rows_slots<true, true>() : rows_slots<false, true>()) : (ln ? rows_slots<true, false>() : rows_slots<false, false>());
  auto go = [&]() {
    if (bf) {
      if (ln) hipLaunchKernelGGL((linear_rows_kernel<true, true>), dim3(G_), dim3(NT), 0, 0, p);
      else hipLaunchKernelGGL((linear_rows_kernel<false, true>), dim3(G_), dim3(NT), 0, 0, p);
    } else if (ln) hipLaunchKernelGGL((linear_rows_kernel<true, false>), dim3(G_), dim3(NT), 0, 0, p);
    else hipLaunchKernelGGL((linear_rows_kernel<false, false>), dim3(G_), dim3(NT), 0, 0, p);
  };
  const int G = (int)(items < slots ? items : slots);
  G_ = G;
  p.base = (int)(items / G); p.rem = (int)(items % G);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipMemset(probe, 0, 8 * 64 * NW * 5 * 8);
    go();
    hipDeviceSynchronize();
  }
  hipEventRecord(e0, 0);
  for (int rep = 0; rep < 20; ++rep) {
    go();
  }
  hipEventRecord(e1, 0); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("20 back-to-back launches: %.1f us each\n", ms * 1e3 / 20);
  std::vector<long long> hp(8 * 64 * NW * 5);
  hipMemcpy(hp.data(), probe, hp.size() * 8, hipMemcpyDeviceToHost);
  printf("N=%d, grid %d workgroups (%d slots), %d items each; per chunk, cycles: [DMA+store wait] [barrier wait] [flush+DMA issue+MFMA loop] [epilogue] | chunk period\n",
         N, G, slots, p.base);
  for (int blk = 0; blk < 8; ++blk) {
    const long long* t = &hp[((blk * 64 + 63) * NW + 0) * 5];
    const long long* f = &hp[((blk * 64 + 1) * NW + 0) * 5];
    const long long* l = &hp[((blk * 64 + p.base - 1) * NW + 0) * 5];
    printf("block %d wave 0: kernel entry -> exit %lld cycles; entry -> chunk 1 %lld; chunk 1 -> last chunk %lld; last chunk -> exit %lld; wall %.1f us -> shader clock %.3f GHz\n",
           blk, t[1] - t[0], f[0] - t[0], l[0] - f[0], t[1] - l[0], (t[2] - t[3]) * 0.01, (t[1] - t[0]) / ((t[2] - t[3]) * 10.0));
  }
  for (int blk = 0; blk < 2; ++blk)
    for (int wv = 0; wv < NW; wv += 3) {
      printf("block %d wave %d:\n", blk, wv);
      double acc[5] = {0, 0, 0, 0, 0}; int n = 0;
      for (int c = 1; c < p.base && c < 63; ++c) {
        const long long* t = &hp[((blk * 64 + c) * NW + wv) * 5];
        const long long* tp = &hp[((blk * 64 + c - 1) * NW + wv) * 5];
        if (!t[0] || !tp[0]) continue;
        const long long d[5] = {t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] - t[3], t[0] - tp[0]};
        if (c < 12) printf("  chunk %2d: %6lld %6lld %6lld %6lld | %6lld\n", c, d[0], d[1], d[2], d[3], d[4]);
        for (int k = 0; k < 5; ++k) acc[k] += d[k];
        ++n;
      }
      if (n) printf("  mean over %d chunks: %.0f %.0f %.0f %.0f | %.0f   (96 MFMAs = 3072 pipe cycles; x3 waves per SIMD = 9216)\n", n, acc[0] / n, acc[1] / n, acc[2] / n, acc[3] / n, acc[4] / n);
    }
  return 0;
}
