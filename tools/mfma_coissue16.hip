// mfma_coissue16.hip -- tools/mfma_coissue.hip's question for v_mfma_f32_16x16x4_f32 (32 cycles; the instruction of
// conv_wino4.hip.h): what does one more instruction between two matrix instructions cost, with the accumulators in VGPRs or in
// AGPRs, from the same wave or from the SIMD's other wave?
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_coissue16.hip -o tools/mfma_coissue16
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
enum { F_NONE = 0, F_VALU = 1, F_LDSR128 = 2, F_LDSW = 3, F_PK = 4, F_LDSR32 = 5 };

template <int KIND, int N, bool AGPR, bool SPLIT, bool BOTH>
__global__ __launch_bounds__(512) void kern(float *out, unsigned long long *cyc, int iters, float a0) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int tid = threadIdx.x, wave = tid >> 6;
  for (int i = tid; i < 8192; i += blockDim.x) lds[i] = a0;
  __syncthreads();
  f32x4 acc[16];
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a = a0 + tid * 1e-3f, b = a0 * 0.5f + tid * 2e-3f;
  float f[8];
  for (int i = 0; i < 8; ++i) f[i] = a + i;
  f32x2 p[4];
  for (int i = 0; i < 4; ++i) p[i] = f32x2{a + i, b - i};
  f32x4 q[4];
  for (int i = 0; i < 4; ++i) q[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const bool do_mfma = !SPLIT || BOTH || wave < 4, do_fill = !SPLIT || wave >= 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      if (do_mfma) {
        if (AGPR) asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc[i]) : "v"(a), "v"(b));
        else acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (do_fill) {
#pragma unroll
        for (int k = 0; k < N; ++k) {
          if (KIND == F_VALU) f[k & 7] = f[k & 7] * 1.0001f + 0.5f;
          if (KIND == F_PK) p[k & 3] = p[k & 3] * f32x2{1.0001f, 1.0002f} + f32x2{0.5f, 0.25f};
          if (KIND == F_LDSR128) q[k & 3] += *reinterpret_cast<const f32x4 *>(lds + ((tid * 4 + 256 * k + 4 * it) & 8188));
          if (KIND == F_LDSR32) f[k & 7] += lds[(tid + 64 * k + it) & 1023];
          if (KIND == F_LDSW) lds[(tid + 64 * k) & 1023] = f[k & 7];
        }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) for (int r = 0; r < 4; ++r) s += acc[i][r];
  for (int i = 0; i < 8; ++i) s += f[i];
  for (int i = 0; i < 4; ++i) s += p[i][0] + p[i][1] + q[i][0] + q[i][1] + q[i][2] + q[i][3];
  out[blockIdx.x * 512 + tid] = s + lds[(tid * 7) & 1023];
  if ((tid & 63) == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

template <int KIND, int N, bool AGPR, bool SPLIT, bool BOTH = false>
void run(const char *what) {
  const int blocks = 256, iters = 1000, threads = SPLIT ? 512 : 256;
  float *d; unsigned long long *c;
  hipMalloc(&d, blocks * 512 * 4); hipMalloc(&c, blocks * 8 * 8); hipMemset(c, 0, blocks * 8 * 8);
  for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((kern<KIND, N, AGPR, SPLIT, BOTH>), dim3(blocks), dim3(threads), 0, 0, d, c, iters, 0.5f);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(blocks * 8);
  hipMemcpy(h.data(), c, h.size() * 8, hipMemcpyDeviceToHost);
  double sum = 0; int n = 0;
  for (int b = 0; b < blocks; ++b) for (int w = 0; w < 4; ++w) { sum += (double)h[b * 8 + w]; ++n; }
  printf("16x16x4 acc in %s, %-16s N=%2d, %s: %7.1f cycles per matrix instruction of waves 0-3\n", AGPR ? "AGPRs" : "VGPRs", what, N,
         BOTH ? "two matrix waves per SIMD, fillers in the second" : SPLIT ? "fillers in the SIMD's other wave" : "fillers in the same wave", sum / n / (iters * 16.0));
  hipFree(d); hipFree(c);
}

int main() {
  run<F_NONE, 0, false, false>("none");        run<F_NONE, 0, true, false>("none");
  run<F_VALU, 1, false, false>("v_fma_f32");   run<F_VALU, 2, false, false>("v_fma_f32");   run<F_VALU, 4, false, false>("v_fma_f32");
  run<F_VALU, 1, true, false>("v_fma_f32");    run<F_VALU, 2, true, false>("v_fma_f32");    run<F_VALU, 4, true, false>("v_fma_f32");
  run<F_PK, 1, false, false>("v_pk_fma_f32");  run<F_PK, 2, false, false>("v_pk_fma_f32");  run<F_PK, 2, true, false>("v_pk_fma_f32");
  run<F_LDSR128, 1, false, false>("ds_read_b128"); run<F_LDSR128, 2, false, false>("ds_read_b128"); run<F_LDSR128, 1, true, false>("ds_read_b128");
  run<F_LDSR32, 1, false, false>("ds_read_b32");   run<F_LDSR32, 2, false, false>("ds_read_b32");
  run<F_LDSW, 2, false, false>("ds_write_b32");
  run<F_VALU, 2, false, true>("v_fma_f32");    run<F_VALU, 4, false, true>("v_fma_f32");    run<F_VALU, 4, true, true>("v_fma_f32");
  run<F_LDSR128, 1, false, true>("ds_read_b128"); run<F_LDSR128, 2, false, true>("ds_read_b128");
  run<F_NONE, 0, false, true, true>("none");   run<F_VALU, 2, false, true, true>("v_fma_f32"); run<F_VALU, 4, false, true, true>("v_fma_f32");
  run<F_LDSR128, 1, false, true, true>("ds_read_b128");
  return 0;
}
