// mfma_coissue.hip -- does anything issue under an fp32 matrix instruction?  One wave per SIMD runs v_mfma_f32_32x32x2_f32 back
// to back (8 independent accumulators) with N filler instructions of one kind between two consecutive matrix instructions, in
// program order (sched_barrier); reported: shader cycles per matrix instruction (s_memtime) -- 64 = the pipe's own time.
// A second experiment puts the fillers into a SECOND wave on the same SIMD (512-thread workgroups: waves w and w + 4 share a
// SIMD) while the first runs matrix instructions only.  For comparison the same with v_mfma_f32_32x32x16_bf16 (32 cycles).
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_coissue.hip -o tools/mfma_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

enum { F_NONE = 0, F_VALU = 1, F_LDSR = 2, F_LDSW = 3, F_SALU = 4 };

template <int KIND, int N, bool BF16, bool SPLIT>
__global__ __launch_bounds__(512) void kern(float *out, unsigned long long *cyc, int iters, float a0) {
  __shared__ float lds[4096];
  const int tid = threadIdx.x, wave = tid >> 6;
  lds[tid] = a0; lds[tid + 512] = a0;
  __syncthreads();
  f32x16 acc[8];
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
  float a = a0 + tid * 1e-3f, b = a0 * 0.5f + tid * 2e-3f;
  bf16x8 ab, bb;
  for (int i = 0; i < 8; ++i) { ab[i] = (__bf16)(a + i); bb[i] = (__bf16)(b - i); }
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = a + i;
  int sacc = tid;
  const bool do_mfma = !SPLIT || wave < 4, do_fill = !SPLIT || wave >= 4;   // SPLIT: fillers in the SIMD's other wave
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (do_mfma) {
        if (BF16) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, acc[i], 0, 0, 0);
        else acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_fill) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
          if (KIND == F_VALU) f[k & 7] = f[k & 7] * 1.0001f + 0.5f;
          if (KIND == F_LDSR) f[k & 7] += lds[(tid + 64 * k + it) & 1023];
          if (KIND == F_LDSW) lds[(tid + 64 * k) & 1023] = f[k & 7];
          if (KIND == F_SALU) sacc = __builtin_amdgcn_readfirstlane(sacc) * 3 + k;
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 8; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  out[blockIdx.x * 512 + tid] = s + sacc + lds[(tid * 7) & 1023];
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int N, bool BF16, bool SPLIT>
void run(const char *what) {
  const int blocks = 256, iters = 2000, threads = SPLIT ? 512 : 256;
  float *d; unsigned long long *c;
  hipMalloc(&d, blocks * 512 * 4); hipMalloc(&c, blocks * 8 * 8); hipMemset(c, 0, blocks * 8 * 8);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kern<KIND, N, BF16, SPLIT>), dim3(blocks), dim3(threads), 0, 0, d, c, iters, 0.5f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) { sum += (double)h[b * 8 + w]; ++n; }   // the waves that run matrix instructions
  printf("%-5s %-28s N=%2d %s: %7.1f cycles per matrix instruction\n", BF16 ? "bf16" : "fp32", what, N, SPLIT ? "fillers in the SIMD's other wave" : "fillers in the same wave       ", sum / n / (iters * 8.0));
  hipFree(d); hipFree(c);
}

int main() {
  run<F_NONE, 0, false, false>("none");
  run<F_VALU, 2, false, false>("v_fma_f32");   run<F_VALU, 4, false, false>("v_fma_f32");   run<F_VALU, 8, false, false>("v_fma_f32");
  run<F_LDSR, 2, false, false>("ds_read_b32"); run<F_LDSR, 4, false, false>("ds_read_b32");
  run<F_LDSW, 2, false, false>("ds_write_b32"); run<F_LDSW, 4, false, false>("ds_write_b32");
  run<F_SALU, 4, false, false>("v_readfirstlane + s_mul");
  run<F_VALU, 4, false, true>("v_fma_f32");    run<F_VALU, 8, false, true>("v_fma_f32");    run<F_VALU, 16, false, true>("v_fma_f32");
  run<F_LDSR, 4, false, true>("ds_read_b32");  run<F_LDSW, 4, false, true>("ds_write_b32");
  run<F_NONE, 0, true, false>("none");
  run<F_VALU, 2, true, false>("v_fma_f32");    run<F_VALU, 4, true, false>("v_fma_f32");    run<F_VALU, 8, true, false>("v_fma_f32");
  run<F_VALU, 4, true, true>("v_fma_f32");     run<F_VALU, 8, true, true>("v_fma_f32");
  return 0;
}
