// Stand-alone timing of the matcher's kernels (csrc/match.hip.h) in the pipeline's configuration: two jobs (stereo and
// temporal match) per launch.  hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/match_bench.hip -o tools/match_bench
// usage: match_bench [n = 1000] [jobs = 2] [reps = 200] [spread = 0]   -- times the unfused form (K12a writes dt, K12b reads it back) and the
// fused form (K12a reduces each tile per row in LDS, K12m merges) and checks that both give the same matches
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <random>
#include <vector>
#include "../superpoint-stereo-visual-odometry_amd/csrc/match.hip.h"
using namespace spvo;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000, njobs = argc > 2 ? atoi(argv[2]) : 2, reps = argc > 3 ? atoi(argv[3]) : 200;
  const float spread = argc > 4 ? (float)atof(argv[4]) : 0.f;   // > 0: every descriptor = one common vector + spread x noise (near-duplicates: many rows inside every window)
  const int ldt = match_ldt(n);
  std::mt19937 rng(1);
  std::normal_distribution<float> nd;
  std::vector<float> h((size_t)4 * n * 256);
  for (int r = 0; r < 4 * n; ++r) {
    double s = 0;
    for (int k = 0; k < 256; ++k) { float v = nd(rng); if (spread > 0.f) v = std::sin(0.37f * k) + spread * v; h[(size_t)r * 256 + k] = v; s += (double)v * v; }
    for (int k = 0; k < 256; ++k) h[(size_t)r * 256 + k] /= (float)std::sqrt(s);
  }
  float *d, *sq, *dt, *bd;
  int *bi;
  int2 *out, *cand;
  int4 *meta;
  const int nt = (n + MATCH_TT - 1) / MATCH_TT;
  CK(hipMalloc(&cand, (size_t)2 * n * nt * MATCH_C * sizeof(int2))); CK(hipMalloc(&meta, (size_t)2 * n * nt * sizeof(int4)));
  CK(hipMalloc(&d, h.size() * 4)); CK(hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&sq, 4 * n * 4 + 64)); CK(hipMalloc(&dt, (size_t)2 * n * ldt * 4)); CK(hipMalloc(&bd, 4 * n * 4)); CK(hipMalloc(&bi, 4 * n * 4));
  CK(hipMalloc(&out, 2 * n * sizeof(int2)));
  hipLaunchKernelGGL(row_sqnorm_kernel, dim3(n), dim3(256), 0, 0, d, 4 * n, nullptr, sq);
  MatchJobs jobs;
  for (int k = 0; k < 2; ++k) {
    MatchJob &j = jobs.j[k];
    j.A = d + (size_t)(2 * k) * n * 256; j.B = d + (size_t)(2 * k + 1) * n * 256; j.na = j.nb = n; j.na_ptr = j.nb_ptr = nullptr;
    j.nA = sq + (2 * k) * n; j.nB = sq + (2 * k + 1) * n; j.dt = dt + (size_t)k * n * ldt; j.best_d2 = bd + 2 * k * n; j.best_idx = bi + 2 * k * n;
    j.cand = cand + (size_t)k * n * nt * MATCH_C; j.meta = meta + (size_t)k * n * nt;
    j.A8 = j.B8 = nullptr; j.qA8 = j.qB8 = nullptr; j.train_best = nullptr; j.out = out + k * n;
  }
  CK(hipFuncSetAttribute((const void *)match_gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, MATCH_LDS_BYTES));
  CK(hipFuncSetAttribute((const void *)match_gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, MATCH_LDS_BYTES));
  const int gx = (n + MATCH_TT - 1) / MATCH_TT, gy = (n + MATCH_QT - 1) / MATCH_QT;
  const dim3 gg(8 * ((gx * gy * njobs + 7) / 8)), gr((n + 3) / 4, njobs);
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  std::vector<int2> ho[2];
  for (int fused = 0; fused < 2; ++fused) {
    CK(hipMemset(out, 0xFF, 2 * n * sizeof(int2)));
    for (int phase = 0; phase < 3; ++phase) {
      for (int w = 0; w < 2; ++w) {   // w = 0 warms up
        CK(hipEventRecord(e0));
        for (int i = 0; i < reps; ++i) {
          if (phase != 1) {
            if (fused) hipLaunchKernelGGL((match_gemm_kernel<false, true>), gg, dim3(256), MATCH_LDS_BYTES, 0, jobs, ldt, MATCH_ERR_REL, nt, gx, gy, njobs);
            else hipLaunchKernelGGL((match_gemm_kernel<false, false>), gg, dim3(256), MATCH_LDS_BYTES, 0, jobs, ldt, 0.f, 0, gx, gy, njobs);
          }
          if (phase != 0) {
            if (fused) hipLaunchKernelGGL(match_merge_kernel<>, gr, dim3(256), sizeof(MatchRerankLds<0>), 0, jobs, nt, ldt, MATCH_ERR_REL, 1, 0, 0.8f);
            else if (n <= 1024) hipLaunchKernelGGL(match_rerank_kernel<4>, gr, dim3(256), sizeof(MatchRerankLds<4>), 0, jobs, ldt, MATCH_ERR_REL, 1, 0, 0.8f);
            else hipLaunchKernelGGL(match_rerank_kernel<0>, gr, dim3(256), sizeof(MatchRerankLds<0>), 0, jobs, ldt, MATCH_ERR_REL, 1, 0, 0.8f);
          }
        }
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
      }
      const double us = ms * 1e3 / reps, fl = 2.0 * n * n * 256 * njobs;
      const char *form = fused ? "fused  " : "unfused";
      if (phase == 0) printf("%s n=%d jobs=%d  gemm   %7.2f us  %6.1f TFLOP/s = %.3f of the fp32 MFMA peak (%d workgroups)\n", form, n, njobs, us, fl / us / 1e6, fl / us / 1e6 / 157.3, gx * gy * njobs);
      if (phase == 1) printf("%s n=%d jobs=%d  %s %7.2f us\n", form, n, njobs, fused ? "merge " : "rerank", us);
      if (phase == 2) printf("%s n=%d jobs=%d  both   %7.2f us\n", form, n, njobs, us);
    }
    ho[fused].resize(2 * n);
    CK(hipMemcpy(ho[fused].data(), out, ho[fused].size() * sizeof(int2), hipMemcpyDeviceToHost));
  }
  long cs = 0, diff = 0;
  for (int i = 0; i < njobs * n; ++i) { cs += ho[1][i].x; diff += ho[0][i].x != ho[1][i].x || ho[0][i].y != ho[1][i].y; }
  printf("checksum %ld, rows on which the two forms differ: %ld\n", cs, diff);
  return 0;
}
