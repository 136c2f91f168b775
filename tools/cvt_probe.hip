// How does v_cvt_pk_u8_f32 round?  (If to nearest even, the v_rndne_f32 in front of it in the INT8 requantisation is redundant.)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const float *x, unsigned *o, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) o[i] = __builtin_amdgcn_cvt_pk_u8_f32(x[i], 0, 0u);
}
int main() {
  const int n = 4096;
  float *hx = new float[n]; unsigned *ho = new unsigned[n];
  for (int i = 0; i < n; ++i) hx[i] = i / 16.0f - 1.0f;   // -1 .. 255 in steps of 1/16: every x.5 included
  float *dx; unsigned *dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, n * 4);
  hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  hipMemcpy(ho, dout, n * 4, hipMemcpyDeviceToHost);
  int rne = 0, trunc = 0, rhu = 0;
  for (int i = 0; i < n; ++i) {
    const float v = hx[i];
    const float c = fminf(fmaxf(v, 0.f), 255.f);
    rne += ho[i] == (unsigned)nearbyintf(c);
    trunc += ho[i] == (unsigned)c;
    rhu += ho[i] == (unsigned)floorf(c + 0.5f);
  }
  printf("v_cvt_pk_u8_f32 over %d values: equals round-to-nearest-even %d, truncation %d, round-half-up %d\n", n, rne, trunc, rhu);
  printf("samples: 0.5 -> %u, 1.5 -> %u, 2.5 -> %u, 2.4375 -> %u, 2.5625 -> %u, -0.5 -> %u, 126.5 -> %u, 127.5 -> %u\n", ho[24], ho[40], ho[56], ho[55], ho[57], ho[8], ho[2040], ho[2056]);
  return 0;
}
