// What does a captured HIP graph save on the HOST for a chain of short dependent kernels on gfx950 / ROCm 7.2?  The small engines' frame loop
// (BASELINE configs 3 and 5) is bound by ~20 kernel launches per frame (NOTES.md round 6).  Arms: N dependent kernels of ~5 us each enqueued with
// hipLaunchKernelGGL on one stream, against ONE hipGraphLaunch of the same N kernels captured from that stream; host time per iteration
// (enqueue only, the stream is drained every 64 iterations) and wall time per iteration.
// hipcc --offload-arch=gfx950 -O3 tools/graph_bench.hip -o tools/graph_bench ; usage: graph_bench [N = 10] [iterations = 2000]
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)

__global__ void spin_kernel(float *p, int us, int k) {
  const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
  while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)us * 100) __builtin_amdgcn_s_sleep(4);
  if (threadIdx.x == 0) p[blockIdx.x] += (float)k;
}

static double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv) {
  const int N = argc > 1 ? std::atoi(argv[1]) : 10, iters = argc > 2 ? std::atoi(argv[2]) : 2000, us = argc > 3 ? std::atoi(argv[3]) : 5;
  float *d;
  CK(hipMalloc(&d, 4096 * 4));
  CK(hipMemset(d, 0, 4096 * 4));
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  auto enqueue = [&]() { for (int k = 0; k < N; ++k) hipLaunchKernelGGL(spin_kernel, dim3(32), dim3(64), 0, s, d, us, k); };
  hipGraph_t g;
  hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  enqueue();
  CK(hipStreamEndCapture(s, &g));
  CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  for (int arm = 0; arm < 2; ++arm) {
    for (int w = 0; w < 50; ++w) { if (arm) CK(hipGraphLaunch(ge, s)); else enqueue(); }
    CK(hipStreamSynchronize(s));
    double host = 0;
    const double t0 = now_us();
    for (int i = 0; i < iters; ++i) {
      const double a = now_us();
      if (arm) CK(hipGraphLaunch(ge, s)); else enqueue();
      host += now_us() - a;
      if ((i & 63) == 63) CK(hipStreamSynchronize(s));
    }
    CK(hipStreamSynchronize(s));
    const double wall = now_us() - t0;
    std::printf("%-28s N = %2d kernels of %d us: host %.1f us per iteration (%.2f per kernel), wall %.1f us per iteration (device floor %d)\n",
                arm ? "one hipGraphLaunch" : "N x hipLaunchKernelGGL", N, us, host / iters, host / iters / N, wall / iters, N * us);
  }
  return 0;
}
