// How many workgroups of T threads with L bytes of LDS does a CU really hold at once?  Every workgroup spins ~20 us and records
// (XCC id, CU id, start, end); the host counts the maximum overlap per CU.
// hipcc --offload-arch=gfx950 -O3 tools/census.hip -o tools/census ; usage: census threads lds_bytes grid
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <map>
#include <algorithm>
__global__ void k(unsigned long long *out, int spin) {
  extern __shared__ int sm[];
  unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  unsigned hw;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
  unsigned xcc;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
  sm[threadIdx.x] = threadIdx.x;
  __syncthreads();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin) {}
  if (threadIdx.x == 0) {
    out[blockIdx.x * 4 + 0] = t0;
    out[blockIdx.x * 4 + 1] = __builtin_amdgcn_s_memrealtime();
    out[blockIdx.x * 4 + 2] = hw;
    out[blockIdx.x * 4 + 3] = xcc + sm[1] - 1;
  }
}
int main(int argc, char **argv) {
  const int threads = atoi(argv[1]), lds = atoi(argv[2]), grid = atoi(argv[3]);
  unsigned long long *d;
  hipMalloc(&d, grid * 32);
  hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
  int occ = 0;
  hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void *)k, threads, lds);
  hipLaunchKernelGGL(k, dim3(grid), dim3(threads), lds, 0, d, 2000);   // 2000 ticks of 10 ns = 20 us
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid * 4);
  hipMemcpy(h.data(), d, grid * 32, hipMemcpyDeviceToHost);
  std::map<unsigned long long, std::vector<std::pair<unsigned long long, int>>> ev;
  for (int b = 0; b < grid; ++b) {
    const unsigned hw = (unsigned)h[b * 4 + 2];
    const unsigned long long key = ((h[b * 4 + 3] & 15) << 16) | (((hw >> 13) & 7) << 8) | ((hw >> 12) & 1) << 6 | ((hw >> 8) & 15);   // xcc, se, sh, cu
    ev[key].push_back({h[b * 4], +1});
    ev[key].push_back({h[b * 4 + 1], -1});
  }
  int mx = 0;
  std::map<int, int> hist;
  for (auto &kv : ev) {
    std::sort(kv.second.begin(), kv.second.end());
    int cur = 0, m = 0;
    for (auto &e : kv.second) { cur += e.second; m = std::max(m, cur); }
    hist[m]++;
    mx = std::max(mx, m);
  }
  unsigned long long tmin = ~0ull, tmax = 0;
  for (int b = 0; b < grid; ++b) { tmin = std::min(tmin, h[b * 4]); tmax = std::max(tmax, h[b * 4 + 1]); }
  printf("threads %d lds %d grid %d: occupancy API %d per CU; distinct CUs seen %zu; max concurrent per CU %d; total %.1f us; histogram:", threads, lds, grid, occ, ev.size(), mx, (tmax - tmin) / 100.0);
  for (auto &p : hist) printf(" %dx:%d", p.first, p.second);
  printf("\n");
  return 0;
}
