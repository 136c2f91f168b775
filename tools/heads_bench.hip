// Stand-alone timing and self-check of heads_fused_kernel (csrc/heads.hip.h) at one feature-map size; HEADS_ABL (compile time) removes
// parts of the kernel to show where its time goes (results are then wrong and the check is skipped).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DHEADS_ABL=n] tools/heads_bench.hip -o tools/heads_bench
// usage: heads_bench H W [batch = 2] [reps = 50]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../superpoint-stereo-visual-odometry_amd/csrc/heads.hip.h"
using namespace spvo;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 45, W = argc > 2 ? atoi(argv[2]) : 147, batch = argc > 3 ? atoi(argv[3]) : 2, reps = argc > 4 ? atoi(argv[4]) : 50;
  const int hp = padded_h(H), wp = padded_w(W);
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> ud(-1.f, 1.f);
  std::vector<float> in((size_t)batch * 512 * hp * wp, 0.f), wd(65 * 256), bd(65), we(256 * 256), be(256);
  for (auto &v : in) v = ud(rng);
  for (auto &v : wd) v = ud(rng) / 16; for (auto &v : we) v = ud(rng) / 16;
  for (auto &v : bd) v = ud(rng) / 10; for (auto &v : be) v = ud(rng) / 10;
  const std::vector<float> pk = pack_heads_weights(wd.data(), bd.data(), 65, we.data(), be.data());
  float *d_in, *d_w, *d_det, *d_desc, *d_raw;
  const size_t plane = (size_t)hp * wp;
  CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_w, pk.size() * 4));
  CK(hipMalloc(&d_det, (size_t)batch * 65 * plane * 4)); CK(hipMalloc(&d_desc, (size_t)batch * H * W * 256 * 4)); CK(hipMalloc(&d_raw, (size_t)batch * 256 * plane * 4));
  CK(hipMemset(d_det, 0, (size_t)batch * 65 * plane * 4));
  CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
  HeadsArgs a{};
  a.in_det = d_in; a.in_desc = d_in + 256 * plane; a.det_in_per_image = a.desc_in_per_image = (size_t)512 * plane; a.in_hp = hp; a.in_wp = wp; a.wpack = d_w;
  a.det = d_det; a.det_per_image = (size_t)65 * plane; a.desc_raw = d_raw; a.raw_per_image = (size_t)256 * plane; a.desc = d_desc; a.H = H; a.W = W; a.batch = batch;
  CK(hipFuncSetAttribute((const void *)heads_fused_kernel<>, hipFuncAttributeMaxDynamicSharedMemorySize, HEADS_LDS_BYTES));
  hipDeviceProp_t prop;
  CK(hipGetDeviceProperties(&prop, 0));
  const dim3 grid(prop.multiProcessorCount);
  auto launch = [&]() { hipLaunchKernelGGL(heads_fused_kernel<>, grid, dim3(HEADS_THREADS), HEADS_LDS_BYTES, 0, a); };
  launch();
  CK(hipDeviceSynchronize());
  if (!HEADS_ABL) {   // self-check against a double-precision evaluation at sampled pixels
    std::vector<float> det((size_t)batch * 65 * plane), desc((size_t)batch * H * W * 256), raw((size_t)batch * 256 * plane);
    CK(hipMemcpy(det.data(), d_det, det.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(desc.data(), d_desc, desc.size() * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(raw.data(), d_raw, raw.size() * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    const int npx = batch * H * W;
    for (int k = 0; k < 400; ++k) {
      const int f = k < 40 && npx >= 40 ? (k < 20 ? k : npx - 1 - (k - 20)) : (int)(rng() % npx);
      const int img = f / (H * W), y = (f % (H * W)) / W, x = f % W;
      const size_t pix = (size_t)(y + PADY) * wp + x + PADX;
      const float *xi = in.data() + (size_t)img * 512 * plane + pix;
      for (int co = 0; co < 65; ++co) {
        double s = bd[co];
        for (int c = 0; c < 256; ++c) s += (double)wd[co * 256 + c] * xi[(size_t)c * plane];
        worst = std::max(worst, std::fabs(s - det[((size_t)img * 65 + co) * plane + pix]));
      }
      double v[256], ss = 0;
      for (int co = 0; co < 256; ++co) {
        double s = be[co];
        for (int c = 0; c < 256; ++c) s += (double)we[co * 256 + c] * xi[(size_t)(256 + c) * plane];
        v[co] = s; ss += s * s;
        worst = std::max(worst, std::fabs(s - raw[((size_t)img * 256 + co) * plane + pix]));
      }
      for (int co = 0; co < 256; ++co) worst = std::max(worst, std::fabs(v[co] / std::sqrt(ss) - desc[(size_t)f * 256 + co]));
    }
    printf("self-check at 400 pixels (first, last, random): max |error| %.3g %s\n", worst, worst < 2e-5 ? "ok" : "FAILED");
    if (!(worst < 2e-5)) return 2;
  }
  a.desc_raw = nullptr;
  for (int i = 0; i < 5; ++i) launch();
  CK(hipDeviceSynchronize());
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  CK(hipEventRecord(e0));
  for (int i = 0; i < reps; ++i) launch();
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms;
  CK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps, fl = 2.0 * batch * H * W * 321.0 * 256;
  printf("heads %dx%d x %d images, %d workgroups: %.2f us, %.1f TFLOP/s algorithmic = %.3f of the fp32 peak\n", H, W, batch, grid.x, us, fl / us / 1e6, fl / us / 1e6 / 157.3);
  return 0;
}
