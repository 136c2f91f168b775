// heads_i8.hip.h -- INT8 engines (BASELINE config 5): the tail of the SuperPoint graphs in ONE launch -- convPb 256 -> 65 (fp32 detector
// planes), convDb 256 -> 256, ReduceL2 + Div, NHWC -- where the plan's ops were three launches of ~12-14 us each on a 45 x 147 map that
// holds 13 230 pixels (conv_i8_kernel twice, l2norm_nhwc_kernel: 39 of the 331 us of a forward pass, round 5).  The counterpart of
// heads.hip.h for C16 int8 activations.
//
// The job is 2.2 GOP -- nothing on v_mfma_i32_16x16x64_i8 -- and 24 MB of HBM traffic (6.8 MB of int8 in, 17 MB of fp32 out): it is a
// load / store kernel with a latency chain, so it is built flat: no LDS staging, no loader wave.
//   * the pixels of all images are one flat sequence cut into tiles of 32 (two sub-tiles of 16); one workgroup of four waves per tile;
//   * a lane's operand of the matrix instruction is 16 consecutive input channels of one pixel = one 16-byte piece of a C16 group, read
//     straight from global memory (lane l: pixel l & 15 of the sub-tile, group 4 ks + (l >> 4) of k-step ks): 8 pieces per branch and lane;
//   * output channels in 21 units of 16 (5 detector units = 80 >= 65, 16 descriptor units), dealt 6 / 5 / 5 / 5 to the waves; weights
//     [unit][k-step][lane][16 bytes] straight from global memory (86 KB: L2-resident), the next unit's under the current unit's instructions;
//   * detector units with the operands swapped (D[px][co]: a lane holds four consecutive pixels of its channel = one 16-byte store into
//     the plane), descriptor units D[co][px] (four consecutive channels of its pixel = 16 bytes of the pixel's [256] row);
//   * epilogue: r = fma(f32(acc), qm[co], bias[co]) -- oracle/net_int8.py's arithmetic for a layer with an fp32 output, so the detector
//     planes and the un-normalised descriptors are the oracle's BIT FOR BIT; squared norms per (unit, sub-tile)
//     through 2 KB of LDS, summed in unit order (a pixel's result does not depend on where its tile falls), one reciprocal per pixel.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "conv_i8.hip.h"

namespace spvo {

struct HeadsArgs8 {
  const int8_t *in_det, *in_desc;          // C16 tensors [g][hp][wp][16]: image 0, first group of each branch's 256 input channels
  size_t det_in_per_image, desc_in_per_image;   // bytes
  int in_hp, in_wp;
  const int8_t *wpack;                      // pack_heads_weights_i8(): [unit 21][k-step 4][lane 64][16]
  const float *qm, *bias;                   // [21 x 16] per output channel of the unit sequence: ws[co] * s_in of its branch, bias (0 beyond a branch's cout)
  float *det;                               // [img][65][hp][wp] padded planes
  size_t det_per_image;                     // floats
  float *desc_raw;                          // [img][256][hp][wp] un-normalised descriptor planes, or NULL (only the synchronous entry points keep them)
  size_t raw_per_image;
  float *desc;                              // [img][H][W][256] normalised, dense
  int H, W, batch;
};

constexpr int HEADS8_CIN = 256, HEADS8_DET_UNITS = 5, HEADS8_UNITS = 21, HEADS8_THREADS = 256;

// quantised OIHW 1x1 weights of both heads (quantize_conv_weights) -> the A / B operand of v_mfma_i32_16x16x64_i8 per (unit, k-step, lane):
// lane l = output channel 16 u + (l & 15) of the unit's branch, its 16 bytes = input channels 64 ks + 16 (l >> 4) + e
inline std::vector<int8_t> pack_heads_weights_i8(const int8_t *wq_det, int cout_det, const int8_t *wq_desc) {
  std::vector<int8_t> out((size_t)HEADS8_UNITS * 4 * 64 * 16, 0);
  for (int u = 0; u < HEADS8_UNITS; ++u)
    for (int o = 0; o < 16; ++o) {
      const bool det = u < HEADS8_DET_UNITS;
      const int co = det ? 16 * u + o : 16 * (u - HEADS8_DET_UNITS) + o;
      if (det && co >= cout_det) continue;
      const int8_t *w = (det ? wq_det : wq_desc) + (size_t)co * HEADS8_CIN;
      for (int ci = 0; ci < HEADS8_CIN; ++ci) {
        const int ks = ci >> 6, q = (ci >> 4) & 3, e = ci & 15;
        out[(((size_t)u * 4 + ks) * 64 + 16 * q + o) * 16 + e] = w[ci];
      }
    }
  return out;
}

typedef float heads8_f4 __attribute__((ext_vector_type(4)));

template <int DUMMY = 0>
__global__ __launch_bounds__(HEADS8_THREADS) void heads_i8_kernel(const HeadsArgs8 a) {
  __shared__ float sred[16][2][16];   // [descriptor unit][sub-tile][px]: squared norm of the unit's 16 channels
  const int tid = threadIdx.x, lane = tid & 63, px = lane & 15, q = lane >> 4;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int hw = a.H * a.W, npx = a.batch * hw;
  const int t32 = blockIdx.x * 32;
  const size_t plane = (size_t)a.in_hp * a.in_wp;

  // this lane's pixel of each sub-tile: image and offset in a padded plane
  int img[2];
  size_t opix[2];
  bool okp[2];
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    const int f = t32 + 16 * h + px;
    okp[h] = f < npx;
    const int fc = okp[h] ? f : 0;
    img[h] = fc / hw;
    const int rem = fc - img[h] * hw, y = rem / a.W, x = rem - y * a.W;
    opix[h] = (size_t)(y + PADY) * a.in_wp + (x + PADX);
  }
  // activations: wave 0 multiplies both branches (detector units 0..4 and descriptor unit 5), the others the descriptor branch
  i32x4 bdet[2][4], bdesc[2][4];
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const size_t at = ((size_t)(4 * ks + q) * plane + opix[h]) * 16;
      bdesc[h][ks] = okp[h] ? *reinterpret_cast<const i32x4 *>(a.in_desc + (size_t)img[h] * a.desc_in_per_image + at) : i32x4{0, 0, 0, 0};
      if (w == 0) bdet[h][ks] = okp[h] ? *reinterpret_cast<const i32x4 *>(a.in_det + (size_t)img[h] * a.det_in_per_image + at) : i32x4{0, 0, 0, 0};
    }
  const int u0 = w == 0 ? 0 : w == 1 ? 6 : w == 2 ? 11 : 16, nu = w == 0 ? 6 : 5;
  const i32x4 *wp = reinterpret_cast<const i32x4 *>(a.wpack) + lane;   // (unit u, k-step ks): + (u * 4 + ks) * 64

  // ---- detector units (wave 0): D[px][co], lane = channel 16 u + px, registers = pixels 4 q .. 4 q + 3 of the sub-tile
  if (w == 0) {
    // where the lane's four pixels go in a padded plane: the flat sequence wraps to the next row (at most once inside four pixels) and image
    size_t doff[2][4];
    bool dok[2][4], dvec[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f0 = t32 + 16 * h + 4 * q, fc = f0 < npx ? f0 : 0;
      const int im = fc / hw, rem = fc - im * hw, y = rem / a.W, x = rem - y * a.W;
      dvec[h] = f0 + 3 < npx && x + 3 < a.W;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int fr = f0 + r < npx ? f0 + r : 0;
        const int ir = fr / hw, yr = (fr - ir * hw) / a.W, xr = fr - ir * hw - yr * a.W;
        dok[h][r] = f0 + r < npx;
        doff[h][r] = (size_t)ir * a.det_per_image + (size_t)(yr + PADY) * a.in_wp + (xr + PADX);
      }
    }
#pragma unroll 1
    for (int u = 0; u < HEADS8_DET_UNITS; ++u) {
      i32x4 wv[4];
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wv[ks] = wp[(u * 4 + ks) * 64];
      const int co = 16 * u + px;
      const float qm = a.qm[co], bi = a.bias[co];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        i32x4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(bdet[h][ks], wv[ks], acc, 0, 0, 0);
        if (co < 65) {
          heads8_f4 v;
#pragma unroll
          for (int r = 0; r < 4; ++r) v[r] = __builtin_fmaf((float)acc[r], qm, bi);
          float *dco = a.det + (size_t)co * plane;
          if (dvec[h]) *reinterpret_cast<heads8_f4 *>(dco + doff[h][0]) = v;
          else {
#pragma unroll
            for (int r = 0; r < 4; ++r)
              if (dok[h][r]) dco[doff[h][r]] = v[r];
          }
        }
      }
    }
  }

  // ---- descriptor units: D[co][px], lane = pixel px, registers = channels 16 ud + 4 q .. + 3
  const int d0 = w == 0 ? 5 : u0, nd = w == 0 ? 1 : nu;   // this wave's descriptor units d0 .. d0 + nd - 1 (unit index in the 21)
  heads8_f4 val[5][2];
#pragma unroll
  for (int s = 0; s < 5; ++s) {
    if (s >= nd) break;
    const int u = d0 + s;
    i32x4 wv[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) wv[ks] = wp[(u * 4 + ks) * 64];
    const heads8_f4 qm4 = *reinterpret_cast<const heads8_f4 *>(a.qm + 16 * u + 4 * q), bi4 = *reinterpret_cast<const heads8_f4 *>(a.bias + 16 * u + 4 * q);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      i32x4 acc = {0, 0, 0, 0};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_i32_16x16x64_i8(wv[ks], bdesc[h][ks], acc, 0, 0, 0);
      float ss = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = __builtin_fmaf((float)acc[r], qm4[r], bi4[r]);
        val[s][h][r] = v;
        ss = fmaf(v, v, ss);
      }
      ss += __shfl_xor(ss, 16);
      ss += __shfl_xor(ss, 32);
      if (q == 0) sred[u - HEADS8_DET_UNITS][h][px] = ss;
      if (a.desc_raw && okp[h]) {
        const int cd = 16 * (u - HEADS8_DET_UNITS) + 4 * q;
#pragma unroll
        for (int r = 0; r < 4; ++r) a.desc_raw[(size_t)img[h] * a.raw_per_image + (size_t)(cd + r) * plane + opix[h]] = val[s][h][r];
      }
    }
  }
  __syncthreads();
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    // 1 / ||d|| once per pixel (one correctly rounded division), then one multiply per channel: within 1.5 ulp of the ONNX graph's d / ||d||
    float ss = sred[0][h][px];
#pragma unroll
    for (int u = 1; u < 16; ++u) ss += sred[u][h][px];   // the 16 units in order
    const float inv = 1.0f / sqrtf(ss);
    const int fh = t32 + 16 * h + px;
    float *op = a.desc + (size_t)fh * 256;
#pragma unroll
    for (int s = 0; s < 5; ++s) {
      if (s >= nd) break;
      const int cd = 16 * (d0 + s - HEADS8_DET_UNITS) + 4 * q;
      heads8_f4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = val[s][h][r] * inv;
      if (fh < npx) *reinterpret_cast<heads8_f4 *>(op + cd) = v;
    }
  }
}

}  // namespace spvo
