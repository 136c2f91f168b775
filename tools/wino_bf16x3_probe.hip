// Round 6, review item 4: what would the F(4x4, 3x3) item cost with its products on the bf16 matrix pipe (three bf16 pieces per fp32 operand,
// six partial products, fp32 accumulation) instead of the fp32 one?  Not the kernel -- its INNER LOOP's issue stream, operands in registers:
//   arm A  (today):    per wave and item 36 x v_mfma_f32_16x16x4_f32 (288 per workgroup of 8 waves = 294 912 multiply-adds) + NV vector instructions
//   arm B  (bf16 x 3): per wave and item 18 x v_mfma_f32_32x32x16_bf16 (144 per workgroup: 6 products x 4 channels = K 24 padded to 32) + NV
//   arm B' (bf16 x 3, items of 8 channels: K = 48 = 3 x 16, no padding): 27 per wave and TWO items
// NV = the item's non-matrix work as vector instructions (input transform, operand reads, staging: the value that reproduces the real
// kernel's ~4000 cycles per item in arm A is the calibration), plus ~30 per wave and item for the split into three bf16 pieces in B / B'.
// One 512-thread workgroup per CU, two waves per SIMD as in conv_wino4_kernel; cycles per item from the shader clock.
// hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/wino_bf16x3_probe.hip -o tools/wino_bf16x3_probe ; usage: wino_bf16x3_probe [items = 2000]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); std::exit(1); } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));

template <int ARM, int NV>
__global__ __launch_bounds__(512) void probe(float *out, unsigned long long *cyc, int items, float seed) {
  const int lane = threadIdx.x & 63;
  float a = seed + lane * 1e-3f, b = seed * 0.5f + lane * 2e-3f;
  f4 acc4[12];
  f16v acc16[3];
  for (int i = 0; i < 12; ++i) acc4[i] = f4{0, 0, 0, 0};
  for (int i = 0; i < 3; ++i) for (int k = 0; k < 16; ++k) acc16[i][k] = 0;
  bf8 ba, bb;
  for (int k = 0; k < 8; ++k) { ba[k] = (__bf16)(a + k); bb[k] = (__bf16)(b - k); }
  float v[8];
  for (int k = 0; k < 8; ++k) v[k] = a * (k + 1);
  __syncthreads();
  const unsigned long long t0 = __builtin_readcyclecounter();
  constexpr int NM = ARM == 0 ? 36 : (ARM == 1 ? 18 : 27);            // matrix instructions per wave and loop body (ARM 2: body = two items)
  constexpr int NVB = ARM == 2 ? 2 * NV : NV;
  constexpr int PER = (NVB + NM - 1) / NM;
  for (int it = 0; it < items; it += (ARM == 2 ? 2 : 1)) {
#pragma unroll
    for (int m = 0; m < NM; ++m) {
      if constexpr (ARM == 0) acc4[m % 12] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc4[m % 12], 0, 0, 0);
      else acc16[m % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ba, bb, acc16[m % 3], 0, 0, 0);
      // the vector work of the item spread between the matrix instructions
#pragma unroll
      for (int q = 0; q < PER; ++q)
        if (m * PER + q < NVB) { const int r = (m + q) & 7; asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[r]) : "v"(a), "v"(b)); }   // exactly one instruction
    }
    __builtin_amdgcn_s_barrier();   // one barrier per item, as in the kernel
  }
  const unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int i = 0; i < 12; ++i) s += acc4[i][0] + acc4[i][3];
  for (int i = 0; i < 3; ++i) s += acc16[i][0] + acc16[i][15];
  for (int k = 0; k < 8; ++k) s += v[k];
  out[blockIdx.x * 512 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int ARM, int NV>
static void run(const char *name, int items, float *d_out, unsigned long long *d_cyc, int ncu) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((probe<ARM, NV>), dim3(ncu), dim3(512), 0, 0, d_out, d_cyc, 64, 1.f);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((probe<ARM, NV>), dim3(ncu), dim3(512), 0, 0, d_out, d_cyc, items, 1.f);
  CK(hipEventRecord(e1));
  CK(hipDeviceSynchronize());
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  std::vector<unsigned long long> h(ncu);
  CK(hipMemcpy(h.data(), d_cyc, ncu * 8, hipMemcpyDeviceToHost));
  double mean = 0;
  for (auto c : h) mean += (double)c;
  mean /= ncu;
  std::printf("%-46s NV = %3d: %7.1f ns per item (wall), %8.0f counter ticks per item\n", name, NV, 1e6 * ms / items, mean / items);
}

int main(int argc, char **argv) {
  const int items = argc > 1 ? std::atoi(argv[1]) : 2000;
  hipDeviceProp_t p;
  CK(hipGetDeviceProperties(&p, 0));
  const int ncu = p.multiProcessorCount;
  float *d_out; unsigned long long *d_cyc;
  CK(hipMalloc(&d_out, (size_t)ncu * 512 * 4)); CK(hipMalloc(&d_cyc, ncu * 8));
  std::printf("%d CUs at %d MHz, one 512-thread workgroup each, %d items\n", ncu, p.clockRate / 1000, items);
  run<0, 0>("A  fp32 16x16x4 x 36 per wave", items, d_out, d_cyc, ncu);
  run<0, 106>("A  fp32 16x16x4 x 36 per wave", items, d_out, d_cyc, ncu);
  run<0, 212>("A  fp32 16x16x4 x 36 per wave", items, d_out, d_cyc, ncu);
  run<0, 320>("A  fp32 16x16x4 x 36 per wave", items, d_out, d_cyc, ncu);
  run<1, 0>("B  bf16 32x32x16 x 18 per wave (K 24 -> 32)", items, d_out, d_cyc, ncu);
  run<1, 136>("B  bf16 32x32x16 x 18 per wave (K 24 -> 32)", items, d_out, d_cyc, ncu);
  run<1, 242>("B  bf16 32x32x16 x 18 per wave (K 24 -> 32)", items, d_out, d_cyc, ncu);
  run<1, 350>("B  bf16 32x32x16 x 18 per wave (K 24 -> 32)", items, d_out, d_cyc, ncu);
  run<2, 0>("B' bf16 32x32x16 x 13.5 per wave (K 48)", items, d_out, d_cyc, ncu);
  run<2, 242>("B' bf16 32x32x16 x 13.5 per wave (K 48)", items, d_out, d_cyc, ncu);
  run<2, 350>("B' bf16 32x32x16 x 13.5 per wave (K 48)", items, d_out, d_cyc, ncu);
  return 0;
}
