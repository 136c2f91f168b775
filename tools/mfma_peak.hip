// mfma_peak.hip -- sustained rate of v_mfma_f32_32x32x2_f32 with no memory traffic: the practical
// ceiling the convolution kernels are judged against on this particular box (clock under load).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
  }
  float s = 0;
  for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters, const char *name) {
  float *d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<NACC><<<blocks, 256>>>(d, iters, 0.5f, 0.25f);
  hipEventRecord(e0);
  k<NACC><<<blocks, 256>>>(d, iters, 0.5f, 0.25f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double fl = (double)blocks * 4 * iters * NACC * 32 * 32 * 2 * 2;
  printf("%-24s blocks %5d iters %6d: %8.3f ms  %7.2f TFLOP/s\n", name, blocks, iters, ms, fl / ms * 1e-9);
  hipFree(d);
}
// back-to-back launches of a short kernel for ~1.5 s: what a kernel of the conv layers' duration can reach
template <int NACC>
void run_sustained(int blocks, int iters, const char *name) {
  float *d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  double total = 0, last = 0;
  while (total < 1500) {
    hipEventRecord(e0);
    for (int i = 0; i < 200; ++i) k<NACC><<<blocks, 256>>>(d, iters, 0.5f, 0.25f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    total += ms; last = ms / 200;
  }
  double fl = (double)blocks * 4 * iters * NACC * 32 * 32 * 2 * 2;
  printf("%-24s blocks %5d iters %6d: %8.3f ms/launch sustained  %7.2f TFLOP/s\n", name, blocks, iters, last, fl / last * 1e-9);
  hipFree(d);
}
int main() {
  run_sustained<8>(256, 50, "8 acc 0.04ms b2b");
  run_sustained<8>(256, 200, "8 acc 0.15ms b2b");
  run_sustained<8>(256, 400, "8 acc 0.3ms b2b");
  run_sustained<8>(256, 800, "8 acc 0.6ms b2b");
  run_sustained<8>(256, 4000, "8 acc 3ms b2b");
  run<8>(256, 400, "8 acc, 1 blk/CU, 0.3ms");
  run<8>(256, 4000, "8 acc, 1 blk/CU, 3ms");
  run<8>(256, 40000, "8 acc, 1 blk/CU, 30ms");
  run<2>(512, 8000, "2 acc, 2 blk/CU");
  run<8>(512, 2000, "8 acc, 2 blk/CU");
  run<1>(256, 16000, "1 acc, 1 blk/CU");
  return 0;
}
