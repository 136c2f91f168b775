// What is the idle time between two DEPENDENT kernels on one stream of an MI355X, and what does it depend on?
// Every kernel stamps the constant-rate clock (s_memrealtime, 100 MHz) when its first workgroup starts and when its last one
// ends; gap(k) = start(k+1) - end(k).  Arms: bytes written per kernel (dirty L2 lines to write back at the kernel's end), the
// stores' cache policy, dynamic LDS per workgroup, grid size, and hipExtAnyOrderLaunch (no barrier bit: independent kernels).
// hipcc --offload-arch=gfx950 -O3 tools/gap_bench.hip -o tools/gap_bench ; usage: gap_bench
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int AUX>
__global__ __launch_bounds__(256) void stamp_kernel(const float4 *__restrict__ src, float4 *__restrict__ dst, size_t n, unsigned long long *stamps,
                                                    int spin_us) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) atomicMin(&stamps[0], t0);
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float4 *>(src), 0, 0x7FFFFFFF, 0x00020000);
  const __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc(dst, 0, 0x7FFFFFFF, 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rs, (unsigned)(i * 16), 0, 0);
    __builtin_amdgcn_raw_buffer_store_b128(v, rd, (unsigned)(i * 16), 0, AUX);
  }
  if (spin_us > 0)   // keeps the kernel alive for a while without memory traffic
    while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
  if (threadIdx.x == 0 && lds == nullptr) dst[0] = float4{lds[0], 0, 0, 0};
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence_block();
    atomicMax(&stamps[1], __builtin_amdgcn_s_memrealtime());
  }
}

typedef void (*kern_t)(const float4 *, float4 *, size_t, unsigned long long *, int);

// a second stream kept busy with small kernels while the chain runs: noise_grid workgroups of noise_us each, noise_lds bytes of LDS
static int g_noise_grid = 0, g_noise_us = 0, g_noise_lds = 0;
__global__ __launch_bounds__(256) void noise_kernel(int us, float *sink) {
  extern __shared__ float lds[];
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(8);
  if (sink == (float *)1) sink[0] = lds[0];
}

static void run(const char *label, kern_t k, size_t mib_written, int lds_bytes, int grid, int spin_us, bool any_order, int chain = 12) {
  const size_t n = std::max<size_t>(mib_written * 1024 * 1024 / 16, 0);
  static float4 *a = nullptr, *b = nullptr;
  static unsigned long long *st = nullptr;
  if (!a) {
    hipMalloc(&a, 512ull << 20); hipMalloc(&b, 512ull << 20); hipMalloc(&st, 2 * 64 * sizeof(unsigned long long));   // chain <= 64
    hipMemset(a, 1, 512ull << 20);
  }
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  if (lds_bytes > 65536) hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
  std::vector<double> gaps, durs;
  for (int rep = 0; rep < 6; ++rep) {
    std::vector<unsigned long long> h(2 * chain);
    for (int i = 0; i < chain; ++i) { h[2 * i] = ~0ull; h[2 * i + 1] = 0; }
    hipMemcpy(st, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipStream_t ns = nullptr;
    if (g_noise_grid) {
      hipStreamCreateWithFlags(&ns, hipStreamNonBlocking);
      for (int i = 0; i < 400; ++i) hipLaunchKernelGGL(noise_kernel, dim3(g_noise_grid), dim3(256), g_noise_lds, ns, g_noise_us, (float *)nullptr);
    }
    for (int i = 0; i < chain; ++i) {
      float4 *src = (i & 1) ? b : a, *dst = (i & 1) ? a : b;     // ping-pong: kernel i+1 reads what kernel i wrote
      if (any_order)
        hipExtLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, s, nullptr, nullptr, hipExtAnyOrderLaunch, (const float4 *)src, dst, n, st + 2 * i, spin_us);
      else
        hipLaunchKernelGGL(k, dim3(grid), dim3(256), lds_bytes, s, (const float4 *)src, dst, n, st + 2 * i, spin_us);
    }
    hipStreamSynchronize(s);
    if (ns) { hipStreamSynchronize(ns); hipStreamDestroy(ns); }
    hipMemcpy(h.data(), st, h.size() * 8, hipMemcpyDeviceToHost);
    if (rep == 0) continue;
    for (int i = 2; i < chain; ++i) {
      gaps.push_back(((double)h[2 * i] - (double)h[2 * i - 1]) / 100.0);
      durs.push_back(((double)h[2 * i + 1] - (double)h[2 * i]) / 100.0);
    }
  }
  std::sort(gaps.begin(), gaps.end()); std::sort(durs.begin(), durs.end());
  printf("%-64s kernel %8.2f us   gap median %6.2f  min %6.2f  max %6.2f us\n", label, durs[durs.size() / 2], gaps[gaps.size() / 2], gaps.front(), gaps.back());
  fflush(stdout);
  hipStreamDestroy(s);
}

int main() {
  run("empty kernel, 256 workgroups", stamp_kernel<0>, 0, 0, 256, 0, false);
  run("empty kernel, 3330 workgroups", stamp_kernel<0>, 0, 0, 3330, 0, false);
  run("empty kernel, 256 workgroups, 157 KB LDS", stamp_kernel<0>, 0, 157 * 1024, 256, 0, false);
  run("40 us of sleeping, no traffic, 256 workgroups", stamp_kernel<0>, 0, 0, 256, 40, false);
  run("40 us of sleeping, no traffic, 256 workgroups, 157 KB LDS", stamp_kernel<0>, 0, 157 * 1024, 256, 40, false);
  run("copy 8 MiB", stamp_kernel<0>, 8, 0, 1024, 0, false);
  run("copy 32 MiB", stamp_kernel<0>, 32, 0, 2048, 0, false);
  run("copy 64 MiB", stamp_kernel<0>, 64, 0, 4096, 0, false);
  run("copy 256 MiB", stamp_kernel<0>, 256, 0, 4096, 0, false);
  run("copy 64 MiB, stores sc0", stamp_kernel<1>, 64, 0, 4096, 0, false);
  run("copy 64 MiB, stores nt", stamp_kernel<2>, 64, 0, 4096, 0, false);
  run("copy 64 MiB, stores sc1", stamp_kernel<16>, 64, 0, 4096, 0, false);
  run("copy 64 MiB, stores sc0 sc1", stamp_kernel<17>, 64, 0, 4096, 0, false);
  run("copy 64 MiB, stores sc1 nt", stamp_kernel<18>, 64, 0, 4096, 0, false);
  run("copy 64 MiB, 157 KB LDS (one workgroup per CU)", stamp_kernel<0>, 64, 157 * 1024, 256, 0, false);
  run("empty kernel, any-order launch (no barrier bit)", stamp_kernel<0>, 0, 0, 256, 0, true);
  run("copy 64 MiB, any-order launch (no barrier bit)", stamp_kernel<0>, 64, 0, 4096, 0, true);
  // the same chains with a second stream active
  g_noise_grid = 4; g_noise_us = 10; g_noise_lds = 0;
  run("copy 64 MiB, 157 KB LDS | noise: 4 workgroups x 10 us", stamp_kernel<0>, 64, 157 * 1024, 256, 0, false, 24);
  run("40 us sleeping, 157 KB LDS | noise: 4 workgroups x 10 us", stamp_kernel<0>, 0, 157 * 1024, 256, 40, false, 24);
  g_noise_grid = 64; g_noise_us = 10; g_noise_lds = 16 * 1024;
  run("copy 64 MiB, 157 KB LDS | noise: 64 workgroups x 10 us, 16 KB LDS", stamp_kernel<0>, 64, 157 * 1024, 256, 0, false, 24);
  run("40 us sleeping, 157 KB LDS | noise: 64 wg x 10 us, 16 KB LDS", stamp_kernel<0>, 0, 157 * 1024, 256, 40, false, 24);
  g_noise_grid = 256; g_noise_us = 20; g_noise_lds = 16 * 1024;
  run("40 us sleeping, 157 KB LDS | noise: 256 wg x 20 us, 16 KB LDS", stamp_kernel<0>, 0, 157 * 1024, 256, 40, false, 24);
  g_noise_grid = 256; g_noise_us = 20; g_noise_lds = 0;
  run("40 us sleeping, 157 KB LDS | noise: 256 wg x 20 us, no LDS", stamp_kernel<0>, 0, 157 * 1024, 256, 40, false, 24);
  return 0;
}
