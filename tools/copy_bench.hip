// A plain 16-byte-per-lane streaming copy of a known size: the calibration kernel for the HBM traffic counters
// (MI355X_MICROARCH.md, HBM: on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced streaming read).  Profiled in the
// SAME rocprofv3 --pmc session as the kernels whose traffic is quoted: measured / known bytes validates the correction.
// hipcc --offload-arch=gfx950 -O3 tools/copy_bench.hip -o tools/copy_bench ; usage: copy_bench [MiB = 1024] [reps = 5]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
__global__ __launch_bounds__(256) void spvo_copy_calibration_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}
// the same amount read through the LDS-DMA path (global_load_lds_dwordx4), the way the convolution kernels stage their tiles:
// is FETCH_SIZE halved for these loads too?  (nothing is written back: the data only lands in LDS)
__global__ __launch_bounds__(512) void spvo_lds_dma_calibration_kernel(const float4 *__restrict__ src, float *__restrict__ sink, size_t n) {
  extern __shared__ __attribute__((aligned(16))) float lds[];   // 8 KB: one piece per thread
  const int wave = threadIdx.x >> 6;
  for (size_t i = (size_t)blockIdx.x * 512 + threadIdx.x; i < n; i += (size_t)gridDim.x * 512)
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + i), (__attribute__((address_space(3))) void *)(lds + wave * 256), 16, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0 && lds[0] == 12345.678f) sink[blockIdx.x] = lds[1];
}
int main(int argc, char **argv) {
  const size_t mib = argc > 1 ? atoi(argv[1]) : 1024;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  const size_t n = mib * 1024 * 1024 / 16;
  float4 *a, *b;
  if (hipMalloc(&a, n * 16) != hipSuccess || hipMalloc(&b, n * 16) != hipSuccess) return 1;
  hipMemset(a, 1, n * 16);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(spvo_copy_calibration_kernel, dim3(256 * 16), dim3(256), 0, 0, a, b, n);
  hipEventRecord(e0);
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(spvo_copy_calibration_kernel, dim3(256 * 16), dim3(256), 0, 0, a, b, n);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  for (int i = 0; i < reps + 1; ++i) hipLaunchKernelGGL(spvo_lds_dma_calibration_kernel, dim3(256 * 4), dim3(512), 8192, 0, a, (float *)b, n);
  hipDeviceSynchronize();
  printf("copy %zu MiB: %.1f us per launch, %.2f TB/s (read + write), known bytes per launch: read %zu, written %zu\n", mib, ms * 1e3 / reps, 2.0 * n * 16 / (ms * 1e-3 / reps) / 1e12, n * 16, n * 16);
  return 0;
}
