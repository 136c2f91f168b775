// heads.hip.h -- K3 + K6 in one launch: the two 1x1 head convolutions of the SuperPoint graphs (convPb 256 -> 65 on the detector
// branch, convDb 256 -> 256 on the descriptor branch) and the descriptor tail (ONNX ReduceL2 + Div, no epsilon, fused with the
// NCHW -> NHWC transpose of feature_detection_neural_network.cpp:339-342).  Replaces three launches of the plan -- two 1x1
// instances of conv_mfma_kernel and l2norm_nhwc_kernel, 30 + 30 + 15 us for 2.2 GFLOP -- on the tail stream, where their CU-time
// is taken from the next pair's trunk.
//
// Workgroup = 4 waves = 32 consecutive pixels of one row x ALL 321 output channels (K = 256 input channels per head, fp32
// v_mfma_f32_32x32x2_f32, D[co][pixel]): wave 0 owns the detector branch (co blocks 0..2: 65 channels padded to 96), waves
// 1..3 the descriptor branch (co blocks of 32: 3 + 3 + 2).  The weights are read ONCE per workgroup, lane-linear 16-byte
// pieces straight into the A operands (each value feeds one matrix instruction: no LDS copy would be reused); the activations
// of both branches are staged through LDS in chunks of 64 channels (global -> registers -> ds_write, next chunk in flight under
// the current one).  Epilogue: bias; detector planes leave as 128-byte row pieces; the descriptor branch reduces the squared norm
// over its three waves through LDS, divides, and leaves through an LDS transpose as [pixel][256] rows (1 KiB per pixel,
// coalesced) -- the layout descriptor sampling (K11) reads.  66 KB of LDS, 2 workgroups per CU, 450 workgroups at 45 x 147 x 2.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "conv_mfma.hip.h"

#ifndef HEADS_ABL
#define HEADS_ABL 0   // measurement builds only (tools/heads_bench.hip): 1 no epilogue, 2 weights read once, 4 no activation staging, 8 no B-operand reads
#endif

namespace spvo {

struct HeadsArgs {
  const float *in;          // padded planes of the tensor both branches read (batch 0, channel 0)
  size_t in_per_image;      // floats
  int in_hp, in_wp;
  int coff_det, coff_desc;  // first input channel of the detector / descriptor branch (256 each)
  const float *wpack;       // pack_heads_weights()
  float *det;               // [img][65][hp][wp] padded planes (same level: hp, wp as the input)
  size_t det_per_image;
  float *desc_raw;          // [img][256][hp][wp] un-normalised descriptor planes, or NULL (only the synchronous entry points keep them)
  size_t raw_per_image;
  float *desc;              // [img][H][W][256] normalised, dense
  int H, W;
};

constexpr int HEADS_BLOCKS = 11;           // 3 detector + 8 descriptor blocks of 32 output channels
constexpr int HEADS_CIN = 256, HEADS_CK = 64, HEADS_PX = 32;
constexpr int HEADS_TP = 256 + 4;          // pitch of the transposed descriptor tile
constexpr int HEADS_LDS_BYTES = (2 * 2 * HEADS_CK * HEADS_PX + HEADS_PX * HEADS_TP + 4 * HEADS_PX) * 4;

// OIHW 1x1 weights + biases of both heads -> [block 11][s4 32][lane 64][4] (lane l: output channel 32 b + (l & 31), input
// channel 2 s + (l >> 5), s = 4 s4 + e) followed by [11 x 32] biases; channels beyond 65 of the detector branch are zero.
inline std::vector<float> pack_heads_weights(const float *w_det, const float *b_det, int cout_det, const float *w_desc, const float *b_desc) {
  std::vector<float> out((size_t)HEADS_BLOCKS * 32 * 64 * 4 + HEADS_BLOCKS * 32, 0.f);
  float *bias = out.data() + (size_t)HEADS_BLOCKS * 32 * 64 * 4;
  for (int b = 0; b < HEADS_BLOCKS; ++b)
    for (int o = 0; o < 32; ++o) {
      const bool det = b < 3;
      const int co = det ? 32 * b + o : 32 * (b - 3) + o;
      if (det && co >= cout_det) continue;
      const float *w = (det ? w_det : w_desc) + (size_t)co * HEADS_CIN;
      bias[32 * b + o] = det ? b_det[co] : b_desc[co];
      for (int ci = 0; ci < HEADS_CIN; ++ci) {
        const int s = ci >> 1, lane = 32 * (ci & 1) + o;
        out[(((size_t)b * 32 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] = w[ci];
      }
    }
  return out;
}

template <int UNUSED = 0>   // (a template so that every translation unit may include this header)
__global__ __launch_bounds__(256, 2) void heads_fused_kernel(const HeadsArgs a) {
  typedef float f32x4v __attribute__((ext_vector_type(4)));
  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *sx = smem;                                            // [buf 2][head 2][ci 64][px 32]
  float *st = smem + 2 * 2 * HEADS_CK * HEADS_PX;              // [px 32][HEADS_TP] normalised descriptors, transposed
  float *sred = st + HEADS_PX * HEADS_TP;                      // [wave 4][px 32] partial squared norms
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int x0 = blockIdx.x * HEADS_PX, y = blockIdx.y, img = blockIdx.z;
  const size_t plane = (size_t)a.in_hp * a.in_wp;
  const int head = wave == 0 ? 0 : 1;
  const int b0 = 3 * wave;                                     // first co block of this wave
  const int nb = wave == 3 ? 2 : 3;

  // ---- activation staging: 1024 16-byte pieces per chunk (2 heads x 64 channels x 8 pieces), 4 per thread
  const float *in_img = a.in + (size_t)img * a.in_per_image + (size_t)(y + PADY) * a.in_wp + (x0 + PADX);
  f32x4v pre[4];
  auto load_chunk = [&](int c) {
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const int p = it * 256 + tid, h = p >> 9, ci = (p >> 3) & 63, q = p & 7;
      pre[it] = *(const f32x4v *)(in_img + (size_t)((h ? a.coff_desc : a.coff_det) + c * HEADS_CK + ci) * plane + q * 4);
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int it = 0; it < 4; ++it) *(f32x4v *)(sx + buf * (2 * HEADS_CK * HEADS_PX) + (it * 256 + tid) * 4) = pre[it];
  };
  const f32x4v *wp4 = reinterpret_cast<const f32x4v *>(a.wpack) + (size_t)b0 * 32 * 64 + lane;   // block b, group s4: + ((b - b0) * 32 + s4) * 64
  f32x16 acc[3];
#pragma unroll
  for (int b = 0; b < 3; ++b)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
  f32x4v a_cur[3], a_nxt[3];   // (two groups in flight instead of one: measured, no change)
#pragma unroll
  for (int b = 0; b < 3; ++b) a_cur[b] = wp4[((b < nb ? b : 0) * 32 + 0) * 64];
  load_chunk(0);
  store_chunk(0);
  __syncthreads();
#pragma unroll 1
  for (int c = 0; c < HEADS_CIN / HEADS_CK; ++c) {
    if (!(HEADS_ABL & 4) && c + 1 < HEADS_CIN / HEADS_CK) load_chunk(c + 1);
    const float *xb = sx + (c & 1) * (2 * HEADS_CK * HEADS_PX) + head * (HEADS_CK * HEADS_PX) + half * HEADS_PX + j;   // k-step s: + 2 s * 32
#pragma unroll
    for (int s4 = 0; s4 < 8; ++s4) {
      const int g = 8 * c + s4 + 1;                            // next group of four k-steps
      if (!(HEADS_ABL & 2) && g < 32) {
#pragma unroll
        for (int b = 0; b < 3; ++b) a_nxt[b] = wp4[((b < nb ? b : 0) * 32 + g) * 64];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const float bv = (HEADS_ABL & 8) ? xb[0] : xb[(2 * (4 * s4 + e)) * HEADS_PX];
#pragma unroll
        for (int b = 0; b < 3; ++b)
          if (b < nb) acc[b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[b][e], bv, acc[b], 0, 0, 0);
      }
#pragma unroll
      for (int b = 0; b < 3; ++b) a_cur[b] = a_nxt[b];
    }
    if (!(HEADS_ABL & 4) && c + 1 < HEADS_CIN / HEADS_CK) {
      store_chunk((c + 1) & 1);   // (the other buffer: its readers passed the barrier at the end of chunk c - 1)
      __syncthreads();
    }
  }

  if (HEADS_ABL & 1) {
    float t = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) t += acc[b][r];
    if (t == 12345.f) a.desc[0] = t;
    return;
  }
  // ---- epilogue.  Register r of a block: output channel (r & 3) + 8 (r >> 2) + 4 half, lane j: pixel x0 + j
  const float *bias = a.wpack + (size_t)HEADS_BLOCKS * 32 * 64 * 4 + 32 * b0;
  const bool px_ok = x0 + j < a.W;
  const size_t opix = (size_t)(y + PADY) * a.in_wp + (x0 + PADX) + j;
  if (wave == 0) {
    float *dp = a.det + (size_t)img * a.det_per_image + opix;
#pragma unroll
    for (int b = 0; b < 3; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int co = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (px_ok && co < 65) dp[(size_t)co * plane] = acc[b][r] + bias[co];
      }
  } else {
    float ss = 0.f;
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if (b < nb) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int cl = 32 * b + (r & 3) + 8 * (r >> 2) + 4 * half;   // channel within this wave's blocks
          const float v = acc[b][r] + bias[cl];
          acc[b][r] = v;
          ss = fmaf(v, v, ss);
          if (a.desc_raw && px_ok) a.desc_raw[(size_t)img * a.raw_per_image + (size_t)(32 * (b0 - 3) + cl) * plane + opix] = v;
        }
      }
    ss += __shfl_xor(ss, 32);
    if (half == 0) sred[wave * HEADS_PX + j] = ss;
  }
  __syncthreads();
  if (wave != 0) {
    const float nrm = sqrtf((sred[1 * HEADS_PX + j] + sred[2 * HEADS_PX + j]) + sred[3 * HEADS_PX + j]);   // fixed order
#pragma unroll
    for (int b = 0; b < 3; ++b)
      if (b < nb) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          f32x4v v;
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = acc[b][4 * q + e] / nrm;
          *(f32x4v *)(st + j * HEADS_TP + 32 * (b0 - 3 + b) + 8 * q + 4 * half) = v;
        }
      }
  }
  __syncthreads();
  // [pixel][256] rows: 8 threads per pixel, 8 x 16 bytes each
  {
    const int p = tid >> 3, part = tid & 7;
    if (x0 + p < a.W) {
      float *op = a.desc + (((size_t)img * a.H + y) * a.W + x0 + p) * 256;
#pragma unroll
      for (int i = 0; i < 8; ++i) *(f32x4v *)(op + (i * 8 + part) * 4) = *(const f32x4v *)(st + p * HEADS_TP + (i * 8 + part) * 4);
    }
  }
}

}  // namespace spvo
