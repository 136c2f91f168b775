// conv_bench.hip -- standalone check + timing of the MFMA convolution variants.
// Build: hipcc --offload-arch=gfx950 -O3 -I superpoint-stereo-visual-odometry_amd/csrc tools/conv_bench.hip -o tools/conv_bench
// Run on the GPU box: ./tools/conv_bench
// Checks every variant against a naive one-thread-per-output kernel on the same
// data, then times it with HIP events and prints achieved TFLOP/s.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include <string>
#include <cstring>
#include <algorithm>
#include "conv_mfma.hip.h"

using namespace spvo;
static int g_oversub = 1;

#define CK_HIP(x)                                                                  \
  do {                                                                             \
    hipError_t e = (x);                                                            \
    if (e != hipSuccess) {                                                         \
      fprintf(stderr, "HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
      exit(1);                                                                     \
    }                                                                              \
  } while (0)

__global__ void naive_conv(const float *in, float *out, const float *w, const float *bias, int B,
                           int cin, int cout, int KS, int H, int W, int hp, int wp, int relu) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int co = blockIdx.z % cout, b = blockIdx.z / cout;
  if (x >= W) return;
  const size_t plane = (size_t)hp * wp;
  float s = bias[co];
  const int hk = KS / 2;
  for (int ky = 0; ky < KS; ++ky)
    for (int kx = 0; kx < KS; ++kx)
      for (int ci = 0; ci < cin; ++ci)
        s = fmaf(w[((size_t)(co * cin + ci) * KS + ky) * KS + kx],
                 in[((size_t)b * cin + ci) * plane + (size_t)(y + PADY + ky - hk) * wp + (x + PADX + kx - hk)], s);
  if (relu) s = fmaxf(s, 0.f);
  out[((size_t)b * cout + co) * plane + (size_t)(y + PADY) * wp + (x + PADX)] = s;
}

__global__ void naive_pool(const float *in, float *out, int C, int OH, int OW, int ihp, int iwp, int ohp, int owp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y, c = blockIdx.z;
  if (x >= OW) return;
  const float *ip = in + (size_t)c * ihp * iwp + (size_t)(2 * y + PADY) * iwp + (2 * x + PADX);
  out[(size_t)c * ohp * owp + (size_t)(y + PADY) * owp + x + PADX] = fmaxf(fmaxf(ip[0], ip[1]), fmaxf(ip[iwp], ip[iwp + 1]));
}

static void pack_weights(const std::vector<float> &w, const std::vector<float> &bias, int cout, int cin, int KS, int CK, std::vector<float> &out, int &co_tiles) {
  co_tiles = (cout + CO_TILE - 1) / CO_TILE;
  out = pack_conv_weights(w.data(), bias.data(), cout, cin, KS, CK);
}

template <int KS, int CK, int WR, int WC, bool POOL, int MINW = 1, int ABL = 0>
static double run_variant(const char *name, int B, int cin, int cout, int H, int W, int reps) {
  using T = ConvTile<KS, CK, WR, WC>;
  const int hp = padded_h(H), wp = padded_w(W);
  const int OH = POOL ? H / 2 : H, OW = POOL ? W / 2 : W;
  const int ohp = padded_h(OH), owp = padded_w(OW);
  const size_t in_n = (size_t)B * cin * hp * wp, out_n = (size_t)B * cout * ohp * owp, full_n = (size_t)B * cout * hp * wp;
  std::vector<float> hin(in_n, 0.f), hw((size_t)cout * cin * KS * KS), hb(((cout + 63) / 64) * 64, 0.f);
  srand(1234);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < cin; ++c)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) hin[(((size_t)b * cin + c) * hp + y + PADY) * wp + x + PADX] = rnd();
  const float ws = sqrtf(2.f / (cin * KS * KS));
  for (auto &v : hw) v = rnd() * ws * 1.7f;
  for (int i = 0; i < cout; ++i) hb[i] = rnd() * 0.1f;
  std::vector<float> hpk;
  int co_tiles;
  pack_weights(hw, hb, cout, cin, KS, CK, hpk, co_tiles);

  float *din, *dout, *dref, *dfull, *dw, *dpk, *db;
  CK_HIP(hipMalloc(&din, in_n * 4));
  CK_HIP(hipMalloc(&dout, out_n * 4));
  CK_HIP(hipMalloc(&dref, out_n * 4));
  CK_HIP(hipMalloc(&dfull, full_n * 4));
  CK_HIP(hipMalloc(&dw, hw.size() * 4));
  CK_HIP(hipMalloc(&dpk, hpk.size() * 4));
  CK_HIP(hipMalloc(&db, hb.size() * 4));
  CK_HIP(hipMemcpy(din, hin.data(), in_n * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(dw, hw.data(), hw.size() * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(dpk, hpk.data(), hpk.size() * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemset(dout, 0, out_n * 4));
  CK_HIP(hipMemset(dref, 0, out_n * 4));
  CK_HIP(hipMemset(dfull, 0, full_n * 4));

  // reference
  naive_conv<<<dim3((W + 63) / 64, H, B * cout), 64>>>(din, POOL ? dfull : dref, dw, db, B, cin, cout, KS, H, W, hp, wp, 1);
  if (POOL) naive_pool<<<dim3((OW + 63) / 64, OH, B * cout), 64>>>(dfull, dref, B * cout, OH, OW, hp, wp, ohp, owp);
  CK_HIP(hipDeviceSynchronize());

  ConvArgs a;
  a.in = din; a.out = dout; a.wpack = dpk; a.bias = db;
  a.H = H; a.W = W; a.in_hp = hp; a.in_wp = wp; a.in_ctot = cin; a.in_coff = 0;
  a.out_hp = ohp; a.out_wp = owp; a.out_ctot = cout; a.out_coff = 0; a.cout = cout;
  a.n_chunks = cin / CK;
  a.tiles_x = (W + T::TW - 1) / T::TW; a.tiles_y = (H + T::TH - 1) / T::TH; a.co_tiles = co_tiles;
  CK_HIP(hipFuncSetAttribute((const void *)conv_mfma_kernel<KS, CK, WR, WC, POOL, true, MINW, ABL>, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  a.batch = B;
  const int n_tiles = a.tiles_x * a.tiles_y * co_tiles * B;
  int per_cu = 1;
  CK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)conv_mfma_kernel<KS, CK, WR, WC, POOL, true, MINW, ABL>, 256, T::LDS_BYTES));
  const int grid = std::min(n_tiles, 256 * std::max(per_cu, 1) * (g_oversub));
  auto kern = conv_mfma_kernel<KS, CK, WR, WC, POOL, true, MINW, ABL>;
  kern<<<grid, 256, T::LDS_BYTES>>>(a);
  CK_HIP(hipDeviceSynchronize());
  std::vector<float> ho(out_n), hr(out_n);
  CK_HIP(hipMemcpy(ho.data(), dout, out_n * 4, hipMemcpyDeviceToHost));
  CK_HIP(hipMemcpy(hr.data(), dref, out_n * 4, hipMemcpyDeviceToHost));
  double maxd = 0, maxr = 0;
  size_t bad = 0;
  for (size_t i = 0; i < out_n; ++i) {
    const double d = fabs((double)ho[i] - hr[i]);
    if (d > maxd) maxd = d;
    if (fabs(hr[i]) > maxr) maxr = fabs(hr[i]);
    if (d > 1e-4 * (1 + fabs(hr[i]))) ++bad;
  }
  hipEvent_t e0, e1;
  CK_HIP(hipEventCreate(&e0));
  CK_HIP(hipEventCreate(&e1));
  for (int i = 0; i < 3; ++i) kern<<<grid, 256, T::LDS_BYTES>>>(a);
  CK_HIP(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) kern<<<grid, 256, T::LDS_BYTES>>>(a);
  CK_HIP(hipEventRecord(e1));
  CK_HIP(hipEventSynchronize(e1));
  float ms;
  CK_HIP(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  const double flops = 2.0 * B * H * W * (double)cout * cin * KS * KS;
  printf("%-28s B%d %3d->%3d %4dx%-4d k%d grid %5d(%d/CU) lds %6d : %8.3f ms  %7.2f TFLOP/s  maxdiff %.3g (ref max %.3g) bad %zu %s\n",
         name, B, cin, cout, H, W, KS, grid, per_cu, T::LDS_BYTES, ms, flops / ms * 1e-9, maxd, maxr, bad, bad ? "FAIL" : "ok");
  hipFree(din); hipFree(dout); hipFree(dref); hipFree(dfull); hipFree(dw); hipFree(dpk); hipFree(db);
  return ms;
}

// In-kernel clock of a production variant under sustained load (MI355X_MICROARCH.md, DVFS give-back
// item 6): >= 2 s of back-to-back launches, then the stamps of the last launch, median over workgroups.
template <int KS, int CK, int WR, int WC, bool POOL, int ABL = 4, int MINW = 1>
static void run_clock(const char *name, int B, int cin, int cout, int H, int W, double seconds) {
  using T = ConvTile<KS, CK, WR, WC>;
  const int hp = padded_h(H), wp = padded_w(W);
  const int OH = POOL ? H / 2 : H, OW = POOL ? W / 2 : W;
  const int ohp = padded_h(OH), owp = padded_w(OW);
  const size_t in_n = (size_t)B * cin * hp * wp, out_n = (size_t)B * cout * ohp * owp;
  std::vector<float> hin(in_n, 0.f), hw((size_t)cout * cin * KS * KS), hb(((cout + 63) / 64) * 64, 0.f), hpk;
  srand(99);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < cin; ++c)
      for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x) hin[(((size_t)b * cin + c) * hp + y + PADY) * wp + x + PADX] = rnd();
  for (auto &v : hw) v = rnd() * 0.1f;
  int co_tiles;
  pack_weights(hw, hb, cout, cin, KS, CK, hpk, co_tiles);
  float *din, *dout, *dpk, *db;
  unsigned long long *dst;
  CK_HIP(hipMalloc(&din, in_n * 4)); CK_HIP(hipMalloc(&dout, out_n * 4)); CK_HIP(hipMalloc(&dpk, hpk.size() * 4)); CK_HIP(hipMalloc(&db, hb.size() * 4));
  CK_HIP(hipMemcpy(din, hin.data(), in_n * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(dpk, hpk.data(), hpk.size() * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemcpy(db, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
  CK_HIP(hipMemset(dout, 0, out_n * 4));
  ConvArgs a;
  a.in = din; a.out = dout; a.wpack = dpk; a.bias = db;
  a.H = H; a.W = W; a.in_hp = hp; a.in_wp = wp; a.in_ctot = cin; a.in_coff = 0;
  a.out_hp = ohp; a.out_wp = owp; a.out_ctot = cout; a.out_coff = 0; a.cout = cout;
  a.n_chunks = cin / CK;
  a.tiles_x = (W + T::TW - 1) / T::TW; a.tiles_y = (H + T::TH - 1) / T::TH; a.co_tiles = co_tiles; a.batch = B;
  auto kern = conv_mfma_kernel<KS, CK, WR, WC, POOL, true, MINW, ABL>;
  CK_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
  int per_cu = 1;
  CK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, (const void *)kern, 256, T::LDS_BYTES));
  const int n_tiles = a.tiles_x * a.tiles_y * co_tiles * B;
  const int grid = std::min(n_tiles, 256 * std::max(per_cu, 1));
  CK_HIP(hipMalloc(&dst, (size_t)grid * 16));
  a.stamps = dst;
  hipEvent_t e0, e1;
  CK_HIP(hipEventCreate(&e0)); CK_HIP(hipEventCreate(&e1));
  kern<<<grid, 256, T::LDS_BYTES>>>(a);
  CK_HIP(hipDeviceSynchronize());
  double total_ms = 0, last = 0;
  int launches = 0;
  while (total_ms < seconds * 1e3) {
    CK_HIP(hipEventRecord(e0));
    for (int i = 0; i < 200; ++i) kern<<<grid, 256, T::LDS_BYTES>>>(a);
    CK_HIP(hipEventRecord(e1));
    CK_HIP(hipEventSynchronize(e1));
    float ms;
    CK_HIP(hipEventElapsedTime(&ms, e0, e1));
    total_ms += ms; launches += 200; last = ms / 200;
  }
  if (ABL != 4) {
    const double fl = 2.0 * B * H * W * (double)cout * cin * KS * KS;
    printf("%-22s grid %5d: %7.3f ms/launch sustained  %7.2f TFLOP/s (ablation %d: results not checked)\n", name, grid, last, fl / last * 1e-9, ABL);
    hipFree(din); hipFree(dout); hipFree(dpk); hipFree(db); hipFree(dst);
    return;
  }
  std::vector<unsigned long long> st((size_t)grid * 2);
  CK_HIP(hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost));
  std::vector<double> mhz, busy_ms;
  for (int i = 0; i < grid; ++i) {
    mhz.push_back((double)st[2 * i] / (double)st[2 * i + 1] * 100.0);
    busy_ms.push_back((double)st[2 * i + 1] * 1e-5);
  }
  std::sort(mhz.begin(), mhz.end());
  std::sort(busy_ms.begin(), busy_ms.end());
  const double flops = 2.0 * B * H * W * (double)cout * cin * KS * KS;
  const double clk = mhz[grid / 2];
  printf("%-22s grid %5d: %7.3f ms/launch (last 200 of %d)  %7.2f TFLOP/s | in-kernel clock median %6.0f MHz (min %6.0f max %6.0f) -> peak at that clock %6.1f TF, frac %.3f | workgroup lifetime median %.3f ms max %.3f ms\n",
         name, grid, last, launches, flops / last * 1e-9, clk, mhz.front(), mhz.back(), 157.3 * clk / 2400.0,
         flops / last * 1e-9 / (157.3 * clk / 2400.0), busy_ms[grid / 2], busy_ms.back());
  hipFree(din); hipFree(dout); hipFree(dpk); hipFree(db); hipFree(dst);
}

int main(int argc, char **argv) {
  const int reps = argc > 1 ? atoi(argv[1]) : 10;
  if (argc > 2) g_oversub = atoi(argv[2]);
  hipDeviceProp_t prop;
  CK_HIP(hipGetDeviceProperties(&prop, 0));
  printf("device: %s, %d CUs, arch %s\n", prop.name, prop.multiProcessorCount, prop.gcnArchName);
  if (argc > 3 && !strcmp(argv[3], "clock")) {
    run_clock<3, 8, 2, 2, true>("conv1b 8x64 pool", 2, 64, 64, 360, 1176, 2.5);
    run_clock<3, 8, 2, 2, false>("ideal 8x64", 2, 64, 64, 256, 1024, 2.5);
    run_clock<3, 8, 1, 2, false>("conv2a 4x64", 2, 64, 64, 180, 588, 2.0);
    run_clock<3, 8, 1, 1, false>("conv4a 4x32", 2, 128, 128, 45, 147, 2.0);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "sweep")) {   // tile / chunk / occupancy variants per VGG layer shape, sustained
#define SW(KS, CK, WR, WC, POOL, MINW, NAME, CI, CO, H, W) run_clock<KS, CK, WR, WC, POOL, 4, MINW>(NAME " k" #KS " ck" #CK " " #WR "x" #WC " w" #MINW, 2, CI, CO, H, W, 0.6)
#define SW_POOL(NAME, CI, CO, H, W) \
    SW(3, 8, 2, 2, true, 1, NAME, CI, CO, H, W); SW(3, 4, 2, 2, true, 2, NAME, CI, CO, H, W); \
    SW(3, 8, 2, 1, true, 2, NAME, CI, CO, H, W); SW(3, 4, 2, 1, true, 2, NAME, CI, CO, H, W); SW(3, 4, 2, 1, true, 4, NAME, CI, CO, H, W)
#define SW_PLAIN(NAME, CI, CO, H, W) \
    SW(3, 8, 2, 2, false, 1, NAME, CI, CO, H, W); SW(3, 4, 2, 2, false, 2, NAME, CI, CO, H, W); \
    SW(3, 8, 2, 1, false, 2, NAME, CI, CO, H, W); SW(3, 4, 2, 1, false, 4, NAME, CI, CO, H, W); \
    SW(3, 8, 1, 2, false, 2, NAME, CI, CO, H, W); SW(3, 4, 1, 2, false, 4, NAME, CI, CO, H, W); \
    SW(3, 8, 1, 1, false, 2, NAME, CI, CO, H, W); SW(3, 8, 1, 1, false, 4, NAME, CI, CO, H, W); SW(3, 4, 1, 1, false, 4, NAME, CI, CO, H, W)
    SW_POOL("conv1b", 64, 64, 360, 1176);
    SW_PLAIN("conv2a", 64, 64, 180, 588);
    SW_POOL("conv2b", 64, 64, 180, 588);
    SW_PLAIN("conv3a", 64, 128, 90, 294);
    SW_POOL("conv3b", 128, 128, 90, 294);
    SW_PLAIN("conv4a", 128, 128, 45, 147);
    SW_PLAIN("convPa", 128, 256, 45, 147);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "occ")) {   // one big workgroup per CU vs two / three smaller-chunk ones
    run_clock<3, 8, 2, 2, false, 4, 1>("ideal 8x64 ck8 1/CU", 2, 64, 64, 256, 1024, 1.5);
    run_clock<3, 4, 2, 2, false, 4, 2>("ideal 8x64 ck4 2/CU", 2, 64, 64, 256, 1024, 1.5);
    run_clock<3, 8, 2, 1, false, 4, 1>("ideal 8x32 ck8", 2, 64, 64, 256, 1024, 1.5);
    run_clock<3, 4, 2, 1, false, 4, 2>("ideal 8x32 ck4 w2", 2, 64, 64, 256, 1024, 1.5);
    run_clock<3, 8, 2, 2, true, 4, 1>("conv1b 8x64 ck8 1/CU", 2, 64, 64, 360, 1176, 1.5);
    run_clock<3, 4, 2, 2, true, 4, 2>("conv1b 8x64 ck4 2/CU", 2, 64, 64, 360, 1176, 1.5);
    run_clock<3, 8, 2, 1, true, 4, 1>("conv1b 8x32 ck8", 2, 64, 64, 360, 1176, 1.5);
    run_clock<3, 4, 2, 1, true, 4, 2>("conv1b 8x32 ck4 w2", 2, 64, 64, 360, 1176, 1.5);
    run_clock<3, 8, 1, 2, false, 4, 1>("conv2a 4x64 ck8", 2, 64, 64, 180, 588, 1.5);
    run_clock<3, 4, 1, 2, false, 4, 2>("conv2a 4x64 ck4 w2", 2, 64, 64, 180, 588, 1.5);
    run_clock<3, 8, 1, 1, false, 4, 1>("conv2a 4x32 ck8", 2, 64, 64, 180, 588, 1.5);
    run_clock<3, 8, 1, 1, false, 4, 1>("conv4a 4x32 ck8", 2, 128, 128, 45, 147, 1.5);
    run_clock<3, 4, 1, 1, false, 4, 1>("conv4a 4x32 ck4", 2, 128, 128, 45, 147, 1.5);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "abl")) {   // what each part of the kernel costs, at sustained clocks
    run_clock<3, 8, 2, 2, false, 4>("ideal 8x64 full", 2, 64, 64, 256, 1024, 1.5);
    run_clock<3, 8, 2, 2, false, 1>("ideal 8x64 no glds", 2, 64, 64, 256, 1024, 1.5);
    run_clock<3, 8, 2, 2, false, 2>("ideal 8x64 no glds/ds", 2, 64, 64, 256, 1024, 1.5);
    run_clock<3, 8, 2, 2, true, 4>("conv1b full", 2, 64, 64, 360, 1176, 1.5);
    run_clock<3, 8, 1, 2, false, 4>("conv2a 4x64 full", 2, 64, 64, 180, 588, 1.5);
    run_clock<3, 8, 1, 1, false, 4>("conv4a 4x32 full", 2, 128, 128, 45, 147, 1.5);
    run_clock<1, 16, 1, 1, false, 4>("convDb 4x32 full", 2, 256, 256, 45, 147, 1.5);
    return 0;
  }
  if (argc > 3 && !strcmp(argv[3], "one")) {   // a single production kernel, for PMC passes
    run_variant<3, 8, 2, 2, true>("conv1b 8x64 pool ck8", 2, 64, 64, 360, 1176, reps);
    return 0;
  }
  // small ragged shapes first (correctness incl. edges)
  run_variant<3, 8, 2, 2, false>("small k3 8x64", 1, 16, 64, 21, 75, 2);
  run_variant<3, 8, 2, 2, true>("small k3 8x64 pool", 2, 16, 70, 22, 74, 2);
  run_variant<3, 8, 1, 1, false>("small k3 4x32", 1, 8, 65, 13, 41, 2);
  run_variant<1, 16, 1, 1, false>("small k1 4x32", 2, 32, 65, 13, 41, 2);
  run_variant<1, 16, 2, 2, false>("small k1 8x64", 1, 16, 48, 21, 75, 2);
  // ablations on an ideal shape (outputs are wrong by construction: FAIL is expected)
  run_variant<3, 8, 2, 2, false, 1, 0>("abl0 8x64 full", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 2, false, 1, 1>("abl1 8x64 no glds", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 2, false, 1, 2>("abl2 8x64 no glds/ds", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 1, true, 1, 0>("abl0 8x32p full", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 1, true, 1, 1>("abl1 8x32p no glds", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 1, true, 1, 2>("abl2 8x32p no glds/ds", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 1, true, 1, 3>("abl3 8x32p stagger", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 1, 1, false, 1, 0>("abl0 4x32 full", 2, 128, 128, 48, 512, reps);
  run_variant<3, 8, 1, 1, false, 1, 1>("abl1 4x32 no glds", 2, 128, 128, 48, 512, reps);
  run_variant<3, 8, 1, 1, false, 1, 2>("abl2 4x32 no glds/ds", 2, 128, 128, 48, 512, reps);
  // scaling with the number of K-chunks (fixed cost vs per-chunk cost), 4 workgroups per CU in sequence
  run_variant<3, 8, 2, 2, false>("scale cin 16", 2, 16, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 2, false>("scale cin 32", 2, 32, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 2, false>("scale cin 64", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 2, false>("scale cin 128", 2, 128, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 2, false>("scale cin 256", 2, 256, 64, 256, 1024, reps);
  // perfectly divisible shapes: in-kernel efficiency without tile waste / grid quantisation
  run_variant<3, 8, 2, 2, false>("ideal 8x64  (4 blk/CU)", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 2, 1, true>("ideal 8x32p (8 blk/CU)", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 4, 2, 1, true>("ideal 8x32p ck4 (8/CU)", 2, 64, 64, 256, 1024, reps);
  run_variant<3, 8, 1, 1, false>("ideal 4x32  (6 blk/CU)", 2, 64, 64, 96, 512, reps);
  run_variant<3, 8, 1, 1, false>("ideal 4x32 128ch (6/CU)", 2, 128, 128, 48, 512, reps);
  run_variant<3, 8, 1, 1, false>("ideal 4x32 128ch (3/CU)", 2, 128, 128, 48, 256, reps);
  // VGG layer shapes at 360x1176, stereo pair (B = 2)
  run_variant<3, 8, 2, 2, true>("conv1b 8x64 pool ck8", 2, 64, 64, 360, 1176, reps);
  run_variant<3, 8, 2, 1, true>("conv1b 8x32 pool ck8", 2, 64, 64, 360, 1176, reps);
  run_variant<3, 4, 2, 1, true>("conv1b 8x32 pool ck4", 2, 64, 64, 360, 1176, reps);
  run_variant<3, 8, 2, 2, false>("conv2a 8x64", 2, 64, 64, 180, 588, reps);
  run_variant<3, 8, 1, 2, false>("conv2a 4x64", 2, 64, 64, 180, 588, reps);
  run_variant<3, 8, 1, 1, false>("conv2a 4x32", 2, 64, 64, 180, 588, reps);
  run_variant<3, 4, 1, 2, false>("conv2a 4x64 ck4", 2, 64, 64, 180, 588, reps);
  run_variant<3, 8, 2, 2, true>("conv2b 8x64 pool", 2, 64, 64, 180, 588, reps);
  run_variant<3, 8, 2, 1, true>("conv2b 8x32 pool", 2, 64, 64, 180, 588, reps);
  run_variant<3, 8, 1, 2, false>("conv3a 4x64", 2, 64, 128, 90, 294, reps);
  run_variant<3, 8, 1, 1, false>("conv3a 4x32", 2, 64, 128, 90, 294, reps);
  run_variant<3, 8, 2, 1, true>("conv3b 8x32 pool", 2, 128, 128, 90, 294, reps);
  run_variant<3, 8, 1, 1, false>("conv4a 4x32", 2, 128, 128, 45, 147, reps);
  run_variant<3, 4, 1, 1, false>("conv4a 4x32 ck4", 2, 128, 128, 45, 147, reps);
  run_variant<3, 8, 1, 1, false>("heads3x3 4x32", 2, 128, 512, 45, 147, reps);
  run_variant<3, 4, 1, 1, false>("heads3x3 4x32 ck4", 2, 128, 512, 45, 147, reps);
  run_variant<1, 16, 1, 1, false>("convDb 4x32", 2, 256, 256, 45, 147, reps);
  run_variant<1, 16, 1, 1, false>("convPb 4x32", 2, 256, 65, 45, 147, reps);
  return 0;
}
