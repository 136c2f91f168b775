// Stand-alone timing of heads_fused_kernel (csrc/heads.hip.h) at one feature-map size; HEADS_ABL (compile time) removes parts of
// the kernel to show where its time goes (results are then wrong).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DHEADS_ABL=n] tools/heads_bench.hip -o tools/heads_bench
// usage: heads_bench H W [reps = 50]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>
#include "../superpoint-stereo-visual-odometry_amd/csrc/heads.hip.h"
using namespace spvo;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main(int argc, char **argv) {
  const int H = argc > 1 ? atoi(argv[1]) : 45, W = argc > 2 ? atoi(argv[2]) : 147, reps = argc > 3 ? atoi(argv[3]) : 50;
  const int hp = padded_h(H), wp = padded_w(W), batch = 2;
  std::mt19937 rng(1);
  std::uniform_real_distribution<float> ud(-1.f, 1.f);
  std::vector<float> in((size_t)batch * 512 * hp * wp, 0.f), wd(65 * 256), bd(65), we(256 * 256), be(256);
  for (auto &v : in) v = ud(rng);
  for (auto &v : wd) v = ud(rng) / 16; for (auto &v : we) v = ud(rng) / 16;
  for (auto &v : bd) v = ud(rng) / 10; for (auto &v : be) v = ud(rng) / 10;
  const std::vector<float> pk = pack_heads_weights(wd.data(), bd.data(), 65, we.data(), be.data());
  float *d_in, *d_w, *d_det, *d_desc;
  CK(hipMalloc(&d_in, in.size() * 4)); CK(hipMalloc(&d_w, pk.size() * 4));
  CK(hipMalloc(&d_det, (size_t)batch * 65 * hp * wp * 4)); CK(hipMalloc(&d_desc, (size_t)batch * H * W * 256 * 4));
  CK(hipMemcpy(d_in, in.data(), in.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
  HeadsArgs a{};
  a.in = d_in; a.in_per_image = (size_t)512 * hp * wp; a.in_hp = hp; a.in_wp = wp; a.coff_det = 0; a.coff_desc = 256; a.wpack = d_w;
  a.det = d_det; a.det_per_image = (size_t)65 * hp * wp; a.desc_raw = nullptr; a.raw_per_image = 0; a.desc = d_desc; a.H = H; a.W = W;
  CK(hipFuncSetAttribute((const void *)heads_fused_kernel<>, hipFuncAttributeMaxDynamicSharedMemorySize, HEADS_LDS_BYTES));
  const dim3 grid((W + HEADS_PX - 1) / HEADS_PX, H, batch);
  auto launch = [&]() { hipLaunchKernelGGL(heads_fused_kernel<>, grid, dim3(256), HEADS_LDS_BYTES, 0, a); };
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
  printf("heads %dx%d x %d images, %d workgroups: %.2f us, %.1f TFLOP/s algorithmic = %.3f of the fp32 peak\n", H, W, batch, grid.x * grid.y * grid.z, us, fl / us / 1e6, fl / us / 1e6 / 157.3);
  return 0;
}
