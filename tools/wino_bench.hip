// Stand-alone run of the Winograd convolution kernel (csrc/conv_wino.hip.h) on one layer shape: timing (HIP events over
// back-to-back launches) and a check of sampled outputs against a float64 direct convolution on the host.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_bench.hip -o tools/wino_bench
// usage: wino_bench H W cin cout pool(0|1) [reps = 50] [grid = CUs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include <algorithm>
#include "../superpoint-stereo-visual-odometry_amd/csrc/conv_wino2.hip.h"
#include "conv_wino64.hip.h"
#include "../superpoint-stereo-visual-odometry_amd/csrc/conv_wino4.hip.h"
#if defined(WINO4)   // F(4x4, 3x3): -DWINO4
#ifndef WINO4_TB
#define WINO4_TB 2   // -DWINO4_TB=1: the 4-wave form (8 x 32 outputs per workgroup)
#endif
#define KERNEL(P, R, T, O) conv_wino4_kernel<P, R, T, WINO4_TB>
#define PACK pack_conv_weights_wino4
#define THREADS (256 * WINO4_TB)
#define COT 64
#elif defined(WINO64)   // filters resident in registers (cin = 64): -DWINO64
#define KERNEL(P, R, T, O) conv_wino64_kernel<P, R, T>
#define PACK pack_conv_weights_wino64
#define THREADS 256
#define COT 64
#elif defined(WINO2) && defined(WINO2_NARROW)   // the 8-wave form with 32 output channels per workgroup
#define KERNEL(P, R, T, O) conv_wino2_kernel<P, R, T, O, true>
#define PACK(w, b, co, ci) pack_conv_weights_wino2(w, b, co, ci, 32)
#define THREADS 512
#define COT 32
#elif defined(WINO2)
#define KERNEL(P, R, T, O) conv_wino2_kernel<P, R, T, O>
#define PACK pack_conv_weights_wino2
#define THREADS 512
#define COT 64
#else   // (the round-1 four-wave form is gone: the default is the 8-wave F(2x2) kernel)
#define KERNEL(P, R, T, O) conv_wino2_kernel<P, R, T, O>
#define PACK pack_conv_weights_wino2
#define THREADS 512
#define COT 64
#define WINO2 1
#endif
using namespace spvo;
#if defined(WINO4)   // multiplies the matrix pipe executes per multiply of the direct convolution
#define EXEC 0.25
#else
#define EXEC (4.0 / 9.0)
#endif
#if defined(WINO4)
#define LDSB Wino4TileT<WINO4_TB>::LDS_BYTES
#elif defined(WINO64)
#define LDSB Wino64Tile::LDS_BYTES
#elif defined(WINO2)
#define LDSB WINO2_LDS_BYTES
#else
#define LDSB WinoTile::LDS_BYTES
#endif
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
  if (argc < 6) { printf("usage: wino_bench H W cin cout pool [reps] [grid]\n"); return 1; }
  const int H = atoi(argv[1]), W = atoi(argv[2]), cin = atoi(argv[3]), cout = atoi(argv[4]), pool = atoi(argv[5]);
  const int reps = argc > 6 ? atoi(argv[6]) : 50;
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount, batch = getenv("WINO_BATCH") ? atoi(getenv("WINO_BATCH")) : 2;
  const int ihp = padded_h(H), iwp = padded_w(W), OH = pool ? H / 2 : H, OW = pool ? W / 2 : W, ohp = padded_h(OH), owp = padded_w(OW);
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> ud(-1.f, 1.f);
  std::vector<float> in((size_t)batch * cin * ihp * iwp, 0.f), w((size_t)cout * cin * 9), b(cout);
  for (int n = 0; n < batch; ++n)
    for (int c = 0; c < cin; ++c)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) in[(((size_t)n * cin + c) * ihp + y + PADY) * iwp + x + PADX] = ud(rng);
  const float ws = 1.f / std::sqrt((float)cin * 9);
  for (auto &v : w) v = ud(rng) * ws * 1.7f;
  for (auto &v : b) v = ud(rng) * 0.1f;
  const std::vector<float> pk = PACK(w.data(), b.data(), cout, cin);
  float *d_in, *d_out, *d_w;
  const size_t out_n = (size_t)batch * cout * ohp * owp;
  CK(hipMalloc(&d_in, (in.size() + (size_t)24 * iwp) * 4)); CK(hipMalloc(&d_out, (out_n + 1024) * 4));   // + slack rows: a 16-row tile stages halo rows below the last plane's padding
  CK(hipMemset(d_in, 0, (in.size() + (size_t)24 * iwp) * 4)); CK(hipMalloc(&d_w, pk.size() * 4));
  CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
  CK(hipMemset(d_out, 0, out_n * 4));
  ConvArgs a{};
  a.in = d_in; a.out = d_out + (getenv("WINO_OUT_SHIFT") ? atoi(getenv("WINO_OUT_SHIFT")) : 0); a.wpack = d_w; a.bias = nullptr; a.H = H; a.W = W;
  a.in_hp = ihp; a.in_wp = iwp; a.in_ctot = cin; a.in_coff = 0; a.out_hp = ohp; a.out_wp = owp; a.out_ctot = cout; a.out_coff = 0;
#if defined(WINO4)
  if ((cin & 3) || (pool && ((H | W) & 1))) { printf("WINO4: cin must be a multiple of 4, H and W even when pooling\n"); return 1; }
  a.cout = cout; a.n_chunks = cin / Wino4Tile::CK; a.tiles_x = (W + Wino4Tile::TW - 1) / Wino4Tile::TW; a.tiles_y = (H + Wino4TileT<WINO4_TB>::TH - 1) / Wino4TileT<WINO4_TB>::TH;
#elif defined(WINO64)
  if (cin != 64 || ((H | W) & 1)) { printf("WINO64: cin must be 64, H and W even\n"); return 1; }
  a.cout = cout; a.n_chunks = Wino64Tile::NCH; a.tiles_x = (W + Wino64Tile::TW - 1) / Wino64Tile::TW; a.tiles_y = (H + Wino64Tile::TH - 1) / Wino64Tile::TH;
#else
  a.cout = cout; a.n_chunks = cin / WinoTile::CK; a.tiles_x = (W + WinoTile::TW - 1) / WinoTile::TW; a.tiles_y = (H + WinoTile::TH - 1) / WinoTile::TH;
#endif
  a.co_tiles = (cout + COT - 1) / COT; a.batch = batch;
  const long n_items = (long)a.tiles_x * a.tiles_y * a.co_tiles * batch * a.n_chunks;
  const int grid = argc > 7 ? atoi(argv[7]) : (int)std::min<long>(cus, n_items / a.n_chunks);
  if (getenv("WINO_DYNAMIC") && atoi(getenv("WINO_DYNAMIC"))) {   // tiles handed out by a counter instead of blockIdx.x + k gridDim.x (8-wave forms)
    int *d_sched;
    CK(hipMalloc(&d_sched, 64)); CK(hipMemset(d_sched, 0, 64));
    a.sched = d_sched;
  }
  auto launch = [&]() {
    if (pool) { hipLaunchKernelGGL((KERNEL(true, true, 0, false)), dim3(grid), dim3(THREADS), LDSB, 0, a); }
    else if ((H | W) & 1) { hipLaunchKernelGGL((KERNEL(false, true, 0, true)), dim3(grid), dim3(THREADS), LDSB, 0, a); }
    else { hipLaunchKernelGGL((KERNEL(false, true, 0, false)), dim3(grid), dim3(THREADS), LDSB, 0, a); }
  };
  CK(hipFuncSetAttribute((const void *)KERNEL(true, true, 0, false), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  CK(hipFuncSetAttribute((const void *)KERNEL(false, true, 0, false), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  CK(hipFuncSetAttribute((const void *)KERNEL(false, true, 0, true), hipFuncAttributeMaxDynamicSharedMemorySize, LDSB));
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, fl = 2.0 * batch * H * W * (double)cout * cin * 9;
#ifdef WINO_STAMPS
  {
    unsigned long long *d_st;
    const int nw = grid * (THREADS / 64);
    CK(hipMalloc(&d_st, (size_t)nw * 8 * 8));
    CK(hipMemset(d_st, 0, (size_t)nw * 8 * 8));
    a.stamps = d_st;
    launch(); launch();
    CK(hipDeviceSynchronize());
    std::vector<unsigned long long> st((size_t)nw * 8);
    CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
    a.stamps = nullptr;
    double sum[7] = {};
    int n = 0;
    for (int i = 0; i < nw; ++i) { if (!st[8 * i + 4]) continue; ++n; for (int k = 0; k < 7; ++k) sum[k] += (double)st[8 * i + k]; }
    const double items = sum[4] / n, clk = sum[5] / sum[6] * 100e6;
    {
      double post = 0;
      for (int i = 0; i < nw; ++i) if (st[8 * i + 4]) post += (double)st[8 * i + 7];
      printf("  of the waits at an item's start, in a tile's second item (behind the epilogue's stores): %.0f cycles per wave = %.0f per tile (%.1f tiles per wave)\n",
             post / n, post / n / std::max(1.0, sum[4] / n / a.n_chunks), sum[4] / n / a.n_chunks);
    }
    {   // by wave of the workgroup (waves w and w + 4 share a SIMD; w is the older one)
      const int wpw = THREADS / 64;
      printf("  by wave, cycles per item (barrier wait | matrix stream):");
      for (int w = 0; w < wpw; ++w) {
        double b = 0, m = 0, it2 = 0;
        for (int i = w; i < nw; i += wpw) if (st[8 * i + 4]) { b += (double)st[8 * i + 1]; m += (double)st[8 * i + 2]; it2 += (double)st[8 * i + 4]; }
        printf("  w%d %.0f | %.0f", w, b / it2, m / it2);
      }
      printf("\n");
    }
#ifdef WINO_STAMPS_ROLES
    {
      double bx = 0, mx = 0;
      for (int i = 0; i < nw; ++i) if (st[8 * i + 4]) { mx += (double)st[8 * i + 0]; bx += (double)st[8 * i + 7]; }
      const double it = sum[4], ix = it / 2, bn = sum[1] - bx, mn = sum[2] - mx, in = it - ix;
      printf("  by role, cycles per item: items in which the wave does the transform's first half: matrix stream %.0f, barrier wait in front %.0f; its other items (second half): matrix stream %.0f, barrier wait in front %.0f\n",
             mx / ix, bx / ix, mn / in, bn / in);
      sum[0] = 0;
    }
#endif
    printf("  per wave: %.1f items, %.0f cycles total at %.2f GHz (100 MHz reference); per item: LDS-DMA wait %.0f, barrier %.0f, matrix stream %.0f, epilogue %.0f (per item share), other %.0f cycles\n",
           items, sum[5] / n, clk / 1e9, sum[0] / n / items, sum[1] / n / items, sum[2] / n / items, sum[3] / n / items,
           (sum[5] - sum[0] - sum[1] - sum[2] - sum[3]) / n / items);
  }
#endif
  std::vector<float> out(out_n);
  CK(hipMemcpy(out.data(), d_out, out_n * 4, hipMemcpyDeviceToHost));
  // sampled check against a float64 direct convolution (+ bias, ReLU, 2x2 max-pool)
  auto conv_at = [&](int n, int co, int y, int x) {
    double s = b[co];
    for (int c = 0; c < cin; ++c)
      for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx)
          s += (double)w[((size_t)co * cin + c) * 9 + ky * 3 + kx] * in[(((size_t)n * cin + c) * ihp + y + ky - 1 + PADY) * iwp + x + kx - 1 + PADX];
    return s > 0 ? s : 0.0;
  };
  double maxerr = 0, maxref = 0;
  std::uniform_int_distribution<int> rn(0, batch - 1), rc(0, cout - 1), ry(0, OH - 1), rx(0, OW - 1);
  for (int k = 0; k < 6000; ++k) {
    const int n = rn(rng), co = rc(rng);
    int y = ry(rng), x = rx(rng);
    if (k < 64) { y = (k & 1) ? OH - 1 : 0; x = (k & 2) ? OW - 1 : 0; }   // corners
    double ref;
    if (pool) ref = std::max(std::max(conv_at(n, co, 2 * y, 2 * x), conv_at(n, co, 2 * y, 2 * x + 1)), std::max(conv_at(n, co, 2 * y + 1, 2 * x), conv_at(n, co, 2 * y + 1, 2 * x + 1)));
    else ref = conv_at(n, co, y, x);
    const double got = out[(((size_t)n * cout + co) * ohp + y + PADY) * owp + x + PADX];
    maxerr = std::max(maxerr, std::fabs(got - ref));
    maxref = std::max(maxref, std::fabs(ref));
  }
  // the zero border of the output planes must stay zero (it is the next layer's halo)
  double border = 0;
  for (int n = 0; n < batch; ++n)
    for (int co = 0; co < cout; co += 7)
      for (int y = 0; y < ohp; ++y)
        for (int x = 0; x < owp; ++x)
          if (y < PADY || y >= OH + PADY || x < PADX || x >= OW + PADX) border = std::max(border, (double)std::fabs(out[(((size_t)n * cout + co) * ohp + y) * owp + x]));
  printf("%dx%d %d->%d pool=%d grid=%d items=%ld (%.2f per workgroup): %8.2f us  %6.1f TFLOP/s algorithmic, %5.1f executed = %.3f of peak | max err %.2e (max |ref| %.2f) border %.1e %s\n",
         H, W, cin, cout, pool, grid, n_items, (double)n_items / grid, us, fl / us / 1e6, fl / us / 1e6 * EXEC, fl / us / 1e6 * EXEC / 157.3, maxerr, maxref, border,
         (maxerr <= 2e-5 * std::max(1.0, maxref) && border == 0) ? "OK" : "MISMATCH");
  return 0;
}
