// heads.hip.h -- K3 + K6 in one launch: the two 1x1 head convolutions of the SuperPoint graphs (convPb 256 -> 65 on the detector
// branch, convDb 256 -> 256 on the descriptor branch) and the descriptor tail (ONNX ReduceL2 + Div, no epsilon, fused with the
// NCHW -> NHWC transpose of feature_detection_neural_network.cpp:339-342).  Replaces three launches of the plan (two 1x1 instances of
// conv_mfma_kernel and l2norm_nhwc_kernel).
//
// Round 5 form.  The job is small (2.17 GFLOP per stereo pair: 13.8 us at the fp32 matrix peak) and was lost to granularity: 32-pixel
// row tiles (450 workgroups on 512 slots, every row's last tile 40 % empty), the detector's 65 channels padded to 96, one of four
// waves a third idle.  Now (45 x 147, two / four images: 40 / 68 us where the round-4 kernel took 45 / 85):
//   * the pixels of ALL images are one flat sequence cut into tiles of 16; every CU -- one workgroup -- takes an equal share of the
//     tiles (+- 1) in steps of two tiles (32 pixels; a share's last step may have one);
//   * v_mfma_f32_16x16x4_f32: the output channels are 21 units of 16 (5 detector units = 80 >= 65, 16 descriptor units); the 42
//     (tile, unit) pairs of a step are dealt 11 / 11 / 10 / 10 to the four computing waves (one tile: 6 / 5 / 5 / 5 units) -- which
//     pairs a wave owns is a template parameter, no branch sits between matrix instructions;
//   * weights straight from global memory into the operands as before (pack_heads_weights: one 16-byte piece per lane = four
//     k-steps; every value feeds the matrix instructions of both tiles), the next group of four k-steps in flight under the current one;
//   * activations through LDS in chunks of 64 input channels, in operand order (a lane's four consecutive k-steps are one
//     conflict-free ds_read_b128), double-buffered, staged by a FIFTH wave, the loader, one chunk ahead of the k-loop; one barrier
//     per chunk.  A wave of its own because `s_waitcnt vmcnt` counts a wave's loads and stores IN ORDER: with the activation loads
//     in the computing waves every weight fetch issued behind them waited for their HBM latency, and a chunked pipeline hid nothing;
//   * epilogue in registers.  Descriptor units are D[co][px]: a lane holds four consecutive channels of its pixel -- squares summed
//     per lane, across the four lane groups by two shuffles, across the 16 units through 2 KB of LDS (in unit order, whichever wave owned a unit), ONE reciprocal per
//     pixel, 16-byte stores into the pixel's [256] row.  Detector units run with the operands SWAPPED, D[px][co]: a lane holds four
//     consecutive pixels of its channel = one 16-byte store into the plane.  Every load of the phase (biases, the next step's first
//     weights) goes out before its first store, and the stores are never waited for: the next step's k-loop runs while they drain.
// Measured on the way (tools/heads_bench, 45 x 147, two / four images; none of it kept): everything staged at once, k-loop, epilogue
// one after the other in two 4-wave workgroups per CU 46 / 77 us (22 / 44 of it matrix instructions); one 8-wave workgroup whose
// halves work half a step apart (k-loop of one beside staging + stores of the other) 48 / 77 -- the loads of one half delay the
// other half's weight fetches in the CU's memory pipeline; chunked staging by the computing waves themselves 49 / 80 (vmcnt order,
// above); two 5-wave workgroups per CU: 168 registers per lane, the compiler spills the epilogue's addresses and every reload between
// two stores waits for the stores before it (50 / 83).
// The two branches may read one tensor (VGG plan: channels 0..255 / 256..511 of the merged convPa + convDa output) or two.
//
// FP16 engines (round 6): the same kernel with F16 = true runs the products on v_mfma_f32_16x16x16_f16 -- a lane's four k-steps of a
// group are four CONSECUTIVE input channels, which is that instruction's operand (four halves per lane, k = 4 (lane >> 4) + i), so ONE
// matrix instruction replaces the four fp32 ones of a group; the weights are packed as fp16 (they were fp16-rounded values stored as
// fp32 before: half the bytes per fetch), the loader moves the C8 activations into LDS as they are (8 bytes per lane instead of 16, no
// conversion), accumulation, bias, norms and stores stay fp32.  Rounds 4-5 converted to fp32 and used the fp32 instruction: 22.7 us at
// 192 x 640, the longest launch of config 3's forward pass, at 0.011 of the fp16 peak the engine is entitled to.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "conv_mfma.hip.h"

#ifndef HEADS_ABL
#define HEADS_ABL 0   // measurement builds only (tools/heads_bench.hip): 1 no epilogue stores, 2 no weight loads in the loop, 4 no activation staging
#endif

namespace spvo {

struct HeadsArgs {
  const float *in_det, *in_desc;        // padded planes: image 0, first of the 256 input channels of each branch (F16IN: C8 fp16 groups [g][hp][wp][8])
  size_t det_in_per_image, desc_in_per_image;   // in 4-byte units
  int in_hp, in_wp;
  const float *wpack;                   // pack_heads_weights()
  float *det;                           // [img][65][hp][wp] padded planes (same level: hp, wp as the input)
  size_t det_per_image;
  float *desc_raw;                      // [img][256][hp][wp] un-normalised descriptor planes, or NULL (only the synchronous entry points keep them)
  size_t raw_per_image;
  float *desc;                          // [img][H][W][256] normalised, dense
  int H, W, batch;
};

constexpr int HEADS_CIN = 256;
constexpr int HEADS_DET_UNITS = 5, HEADS_UNITS = 21;   // units of 16 output channels: detector 0..4 (65 of 80 used), descriptor 5..20
constexpr int HEADS_CHUNK_FLOATS = 2 * 4 * 2 * 64 * 4; // one chunk of 64 input channels: [head 2][group of four k-steps 4][tile 2][lane 64][4]
constexpr int HEADS_SRED_FLOATS = 16 * 2 * 16;         // [descriptor unit 16][tile 2][px 16]
constexpr int HEADS_LDS_BYTES = (2 * HEADS_CHUNK_FLOATS + HEADS_SRED_FLOATS) * 4;

// OIHW 1x1 weights + biases of both heads -> [unit 21][s4 16][lane 64][e 4] (lane l: output channel 16 u + (l & 15) of the unit's
// branch, input channel 16 s4 + 4 e + (l >> 4): the A operand of k-step 4 s4 + e) followed by [21 x 16] biases; detector channels
// beyond `cout_det` are zero.  `f16` (FP16 engines, whose activations are C8 fp16): the weights rounded to fp16 as conv_f16.hip.h's
// packer rounds them (biases stay fp32), and the lane's four k-steps are four CONSECUTIVE input channels -- 16 s4 + 4 (l >> 4) + e --
// so that the loader's lane fetches them as one 8-byte piece of a C8 group.
inline std::vector<float> pack_heads_weights(const float *w_det, const float *b_det, int cout_det, const float *w_desc, const float *b_desc, bool f16 = false) {
  // f16: [unit 21][s4 16][lane 64][4 halves] = two floats' worth of bits per lane, then the biases (fp32)
  const size_t per_lane = f16 ? 2 : 4;
  std::vector<float> out((size_t)HEADS_UNITS * 16 * 64 * per_lane + HEADS_UNITS * 16, 0.f);
  float *bias = out.data() + (size_t)HEADS_UNITS * 16 * 64 * per_lane;
  _Float16 *out_h = reinterpret_cast<_Float16 *>(out.data());
  for (int u = 0; u < HEADS_UNITS; ++u)
    for (int o = 0; o < 16; ++o) {
      const bool det = u < HEADS_DET_UNITS;
      const int co = det ? 16 * u + o : 16 * (u - HEADS_DET_UNITS) + o;
      if (det && co >= cout_det) continue;
      const float *w = (det ? w_det : w_desc) + (size_t)co * HEADS_CIN;
      bias[16 * u + o] = det ? b_det[co] : b_desc[co];
      for (int ci = 0; ci < HEADS_CIN; ++ci) {
        const int s4 = ci >> 4, e = f16 ? ci & 3 : (ci >> 2) & 3, lane = 16 * (f16 ? (ci >> 2) & 3 : ci & 3) + o;
        const size_t at = (((size_t)u * 16 + s4) * 64 + lane) * 4 + e;
        if (f16) out_h[at] = (_Float16)w[ci];
        else out[at] = w[ci];
      }
    }
  return out;
}

typedef float heads_f4 __attribute__((ext_vector_type(4)));
typedef _Float16 heads_h4 __attribute__((ext_vector_type(4)));
template <bool F16> struct HeadsOperand { typedef heads_f4 T; };       // one lane's operand of a group of four k-steps: four floats ...
template <> struct HeadsOperand<true> { typedef heads_h4 T; };          // ... or four halves (FP16 engines)
template <bool F16> __device__ __forceinline__ typename HeadsOperand<F16>::T heads_abl_operand() {   // (HEADS_ABL & 2: constants instead of weight fetches)
  typename HeadsOperand<F16>::T v;
  v[0] = 1; v[1] = 2; v[2] = 3; v[3] = 4;
  return v;
}

struct HeadsPix { int f[2], img[2]; size_t opix[2]; };   // the lane's pixel in the step's two tiles: flat index, image, offset in a padded plane

__device__ __forceinline__ HeadsPix heads_pix(const HeadsArgs &a, int t, int lane, int npx) {
  HeadsPix p;
  const int hw = a.H * a.W;
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    p.f[h] = (t + h) * 16 + (lane & 15);
    const int fc = p.f[h] < npx ? p.f[h] : 0;
    p.img[h] = fc / hw;
    const int rem = fc - p.img[h] * hw, y = rem / a.W, x = rem - y * a.W;
    p.opix[h] = (size_t)(y + PADY) * a.in_wp + (x + PADX);
  }
  return p;
}

// One chunk (64 input channels = four groups of four k-steps) of one wave's k-loop.  HM = which tiles of its six unit slots the wave
// computes, two bits per slot (bit 0: tile 0, bit 1: tile 1); slot s = unit u0 + s.  W0: slots 0..4 are detector units (wave 0), slot 5
// and every slot of the other waves descriptor units.  `wa` holds the weights of the chunk's first group on entry and those of the
// next chunk's first group on exit (LAST: nothing is fetched behind the step's last group).
template <unsigned HM, bool W0, bool F16>
__device__ __forceinline__ void heads_wave_chunk(const HeadsArgs &a, const float *sx, const int u0, const int lane, const int c, heads_f4 (&acc)[6][2],
                                                 typename HeadsOperand<F16>::T (&wa)[6]) {
  typedef typename HeadsOperand<F16>::T op_t;
  constexpr auto hm = [](int s) { return (HM >> (2 * s)) & 3u; };
  constexpr unsigned any = hm(0) | hm(1) | hm(2) | hm(3) | hm(4) | hm(5);
  constexpr unsigned any_desc = W0 ? hm(5) : any;
  const op_t *wp4 = reinterpret_cast<const op_t *>(a.wpack) + (size_t)u0 * 16 * 64 + lane;   // slot s, group s4: + (s * 16 + s4) * 64
  const op_t *xb4 = reinterpret_cast<const op_t *>(sx) + lane;                               // [head][s4 of the chunk][tile]: + ((head * 4 + s4l) * 2 + tile) * 64
  op_t wb[6], ba[4], bb[4];   // activations: detector tile 0 / 1, descriptor tile 0 / 1
  auto load_w = [&](op_t (&wv)[6], int s4) {
#pragma unroll
    for (int s = 0; s < 6; ++s)
      if (hm(s)) wv[s] = (HEADS_ABL & 2) ? heads_abl_operand<F16>() : wp4[(s * 16 + s4) * 64];
  };
  auto load_b = [&](op_t (&bv)[4], int s4l) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (W0 && (any >> h & 1)) bv[h] = xb4[((0 * 4 + s4l) * 2 + h) * 64];
      if (any_desc >> h & 1) bv[2 + h] = xb4[((1 * 4 + s4l) * 2 + h) * 64];
    }
  };
  auto mfmas = [&](const op_t (&wv)[6], const op_t (&bv)[4]) {
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      // descriptor units D[co][px]: a lane ends up with four consecutive CHANNELS of its pixel (the [256] rows); detector units with
      // the operands swapped, D[px][co]: four consecutive PIXELS of its channel (the planes) -- 16-byte stores both
      if constexpr (F16) {   // the group's four k-steps in ONE instruction: k = 4 (lane >> 4) + i
#pragma unroll
        for (int h = 0; h < 2; ++h)
          if (hm(s) >> h & 1) {
            if (W0 && s < 5) acc[s][h] = __builtin_amdgcn_mfma_f32_16x16x16f16(bv[h], wv[s], acc[s][h], 0, 0, 0);
            else acc[s][h] = __builtin_amdgcn_mfma_f32_16x16x16f16(wv[s], bv[2 + h], acc[s][h], 0, 0, 0);
          }
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
          for (int h = 0; h < 2; ++h)
            if (hm(s) >> h & 1) {
              if (W0 && s < 5) acc[s][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[h][e], wv[s][e], acc[s][h], 0, 0, 0);
              else acc[s][h] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv[s][e], bv[2 + h][e], acc[s][h], 0, 0, 0);
            }
      }
    }
  };
  // (the fences keep every fetch where it is written: one group of weights and activations in flight under the group being
  // multiplied -- the scheduler otherwise hoists all of a step's fetches to its top and spills)
  load_b(ba, 0);
  load_w(wb, 4 * c + 1);
  load_b(bb, 1);
  __builtin_amdgcn_sched_barrier(0);
  mfmas(wa, ba);
  __builtin_amdgcn_sched_barrier(0);
  load_w(wa, 4 * c + 2);
  load_b(ba, 2);
  __builtin_amdgcn_sched_barrier(0);
  mfmas(wb, bb);
  __builtin_amdgcn_sched_barrier(0);
  load_w(wb, 4 * c + 3);
  load_b(bb, 3);
  __builtin_amdgcn_sched_barrier(0);
  mfmas(wa, ba);
  __builtin_amdgcn_sched_barrier(0);
  if (c < 3) load_w(wa, 4 * c + 4);
  __builtin_amdgcn_sched_barrier(0);
  mfmas(wb, bb);
  __builtin_amdgcn_sched_barrier(0);
}

// behind a step's last chunk: bias on the descriptor units and, per (unit, tile), the sum of the squares of the unit's 16 channels for every
// pixel (read behind the next barrier).  Per UNIT, not per wave: which wave owns a unit depends on the step (two tiles or one, first
// or second tile), and a pixel's result must not depend on where its tile falls in the launch -- one image alone, in a pair, or in a
// group of four give bit-identical descriptors (tests/test_gpu_network.py, test_trunk_pairing_does_not_change_results)
template <unsigned HM, bool W0, bool F16>
__device__ __forceinline__ void heads_wave_norms(const HeadsArgs &a, float *sred, const int u0, const int lane, heads_f4 (&acc)[6][2]) {
  constexpr auto hm = [](int s) { return (HM >> (2 * s)) & 3u; };
  const int px = lane & 15, kk = lane >> 4;
  const float *bias = a.wpack + (size_t)HEADS_UNITS * 16 * 64 * (F16 ? 2 : 4);
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    if (!hm(s) || (W0 && s < 5)) continue;
    const heads_f4 bv = *reinterpret_cast<const heads_f4 *>(bias + 16 * (u0 + s) + 4 * kk);
#pragma unroll
    for (int h = 0; h < 2; ++h)
      if (hm(s) >> h & 1) {
        float ss = 0.f;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[s][h][r] + bv[r];
          acc[s][h][r] = v;
          ss = fmaf(v, v, ss);
        }
        ss += __shfl_xor(ss, 16);
        ss += __shfl_xor(ss, 32);
        if (kk == 0) sred[((u0 + s - HEADS_DET_UNITS) * 2 + h) * 16 + px] = ss;
      }
  }
}

// ... behind that barrier: detector planes, normalisation, descriptor rows
template <unsigned HM, bool W0, bool F16>
__device__ __forceinline__ void heads_wave_finish(const HeadsArgs &a, const float *sred, const int u0, const int lane, const int npx, const int t,
                                                  const heads_f4 (&acc)[6][2]) {
  const int hw = a.H * a.W;
  constexpr auto hm = [](int s) { return (HM >> (2 * s)) & 3u; };
  constexpr unsigned any = hm(0) | hm(1) | hm(2) | hm(3) | hm(4) | hm(5);
  constexpr unsigned any_desc = W0 ? hm(5) : any;
  const int px = lane & 15;
  int kk = lane >> 4, plane_i = a.in_hp * a.in_wp;
  // (opaque per step: the compiler otherwise hoists the ~100 channel indices and plane offsets of this phase out of the step loop as 64-bit
  // values and SPILLS them -- and a reload between two global stores waits, vmcnt being in order, for every store before it)
  asm volatile("" : "+v"(kk), "+s"(plane_i));
  const float *bias = a.wpack + (size_t)HEADS_UNITS * 16 * 64 * (F16 ? 2 : 4);
  const size_t plane = (size_t)plane_i;
  // register r of a unit: output channel 16 u + 4 kk + r, the lane's pixel px of tile h.
  // Every load of this phase goes out BEFORE its first store: vmcnt counts loads and stores in order, so a load's data is only there
  // when every store issued before it has been acknowledged -- the detector biases fetched slot by slot between the stores made this
  // phase 15 us instead of 1.5 (the acknowledgement of a store takes microseconds when every CU stores at once)
  float bvd[5];
#pragma unroll
  for (int s = 0; s < 5; ++s)
    if (W0 && hm(s)) bvd[s] = bias[16 * (u0 + s) + px];
  // detector units (operands swapped in the k-loop): lane = channel 16 u + px, registers = pixels 4 kk .. 4 kk + 3 of the tile.  Where the
  // four pixels go in a padded plane is the same for every unit: worked out ONCE per tile, with two divisions (the flat pixel sequence
  // wraps to the next row, at most once inside four pixels, and to the next image) -- per unit and pixel it was 7 of this kernel's 40 us
  // on wave 0, for which the other waves wait at the next barrier
  size_t doff[2][4];
  bool dok[2][4], dvec[2];
  if (W0) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int f0 = (t + h) * 16 + 4 * kk, fc = f0 < npx ? f0 : 0;
      const int im = fc / hw, rem = fc - im * hw, y = rem / a.W, x = rem - y * a.W;
      dvec[h] = f0 + 3 < npx && x + 3 < a.W;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        int xr = x + r, yr = y, ir = im;
        if (a.W >= 4) {   // at most one wrap inside four pixels
          if (xr >= a.W) { xr -= a.W; yr += 1; }
          if (yr >= a.H) { yr = 0; ir += 1; }
        } else {          // (maps narrower than four cells: the general form)
          const int fr = f0 + r < npx ? f0 + r : 0;
          ir = fr / hw;
          yr = (fr - ir * hw) / a.W;
          xr = fr - ir * hw - yr * a.W;
        }
        dok[h][r] = f0 + r < npx;
        doff[h][r] = (size_t)ir * a.det_per_image + (size_t)(yr + PADY) * a.in_wp + (xr + PADX);
      }
    }
  }
#pragma unroll
  for (int s = 0; s < 6; ++s) {
    if (!hm(s)) continue;
    const int u = u0 + s;
    if (W0 && s < 5) {
      const int co = 16 * u + px;
      float *dco = a.det + (size_t)co * plane;
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if (hm(s) >> h & 1) {
          if (co < 65) {
            heads_f4 v;
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = acc[s][h][r] + bvd[s];
            // one 16-byte store when the four pixels lie in one image row (dword stores leave a wave at a fraction of the rate)
            if (HEADS_ABL & 1) { if (v[0] == 12345.f) dco[doff[h][0]] = v[1]; }   // (the values stay live: the matrix instructions are not optimised away)
            else if (dvec[h]) *reinterpret_cast<heads_f4 *>(dco + doff[h][0]) = v;
            else {
#pragma unroll
              for (int r = 0; r < 4; ++r)
                if (dok[h][r]) dco[doff[h][r]] = v[r];
            }
          }
        }
    } else if (a.desc_raw) {   // (the synchronous entry points only: spvo_forward / spvo_debug_tensor expose the un-normalised planes)
      const int cd = 16 * (u - HEADS_DET_UNITS) + 4 * kk;
      const HeadsPix p = heads_pix(a, t, lane, npx);
#pragma unroll
      for (int h = 0; h < 2; ++h)
        if ((hm(s) >> h & 1) && p.f[h] < npx) {
#pragma unroll
          for (int r = 0; r < 4; ++r) a.desc_raw[(size_t)p.img[h] * a.raw_per_image + (size_t)(cd + r) * plane + p.opix[h]] = acc[s][h][r];
        }
    }
  }
#pragma unroll
  for (int h = 0; h < 2; ++h) {
    if (!(any_desc >> h & 1)) continue;
    // 1 / ||d|| once per pixel (one correctly rounded division), then one multiply per channel: within 1.5 ulp of the ONNX graph's d / ||d||
    float ss = sred[(0 * 2 + h) * 16 + px];
#pragma unroll
    for (int u = 1; u < 16; ++u) ss += sred[(u * 2 + h) * 16 + px];   // the 16 units in order
    const float inv = 1.0f / sqrtf(ss);
    const int fh = (t + h) * 16 + px;   // the lane's pixel in the flat sequence = its row of the dense [pixel][256] output
    float *op = a.desc + (size_t)fh * 256;
#pragma unroll
    for (int s = 0; s < 6; ++s) {
      if (!(hm(s) >> h & 1) || (W0 && s < 5)) continue;
      const int cd = 16 * (u0 + s - HEADS_DET_UNITS) + 4 * kk;
      heads_f4 v;
#pragma unroll
      for (int r = 0; r < 4; ++r) v[r] = acc[s][h][r] * inv;
      if (HEADS_ABL & 1) { if (v[0] == 12345.f) *reinterpret_cast<heads_f4 *>(op + cd) = v; }
      else if (fh < npx) *reinterpret_cast<heads_f4 *>(op + cd) = v;
    }
  }
}

constexpr unsigned heads_hm(unsigned s0, unsigned s1, unsigned s2, unsigned s3, unsigned s4, unsigned s5) {
  return s0 | s1 << 2 | s2 << 4 | s3 << 6 | s4 << 8 | s5 << 10;
}

// the four waves: which (tile, unit) pairs each owns in a step of two tiles (11 / 11 / 10 / 10 pairs) and of one tile (6 / 5 / 5 / 5)
constexpr unsigned HEADS_A2 = heads_hm(3, 3, 3, 3, 3, 1), HEADS_B2 = heads_hm(2, 3, 3, 3, 3, 3), HEADS_C2 = heads_hm(3, 3, 3, 3, 3, 0);
constexpr unsigned HEADS_A1 = heads_hm(1, 1, 1, 1, 1, 1), HEADS_B1 = heads_hm(0, 1, 1, 1, 1, 1), HEADS_C1 = heads_hm(1, 1, 1, 1, 1, 0);

// The LOADER wave (wave 4 of the workgroup): stages every chunk of 64 input channels of both branches -- 32 pixels x 128 channels = 16 KB
// -- global -> registers -> LDS in operand order, one chunk ahead of the k-loop.  A wave of its own because `s_waitcnt vmcnt` counts a
// wave's loads and stores IN ORDER: with the activation loads in the computing waves (round 5's second form) every weight fetch issued
// behind them waited for their HBM latency, and the chunked pipeline hid nothing (49 us; with the loader 3x us).  Pass v (0..3) covers the
// quarter of the chunk the v-th computing wave would have staged: tile v & 1, input channels 16 (2 i + (v >> 1)) + 4 e + (lane >> 4).
template <bool F16IN>
struct HeadsLoader {
  typedef unsigned heads_u2 __attribute__((ext_vector_type(2)));
  const float *src[2][2];   // [tile][branch]: this lane's pixel, channel lane >> 4 (F16IN: channels 4 (lane >> 4) .. + 3 of the branch's first C8 groups)
  size_t plane;
  int lane, nh;
  bool ok[2];               // the lane's pixel of each tile exists in the images
  heads_f4 sv[F16IN ? 1 : 4][4];   // [pass][branch * 2 + i]: the chunk in flight
  heads_u2 hv[F16IN ? 4 : 1][4];   // the same of an FP16 engine: four fp16 values per piece
  __device__ __forceinline__ void pixels(const HeadsArgs &a, int t, int nh_, int npx) {
    const int hw = a.H * a.W;
    nh = nh_;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int fpx = (t + h) * 16 + (lane & 15);
      ok[h] = h < nh && fpx < npx;
      const int fc = ok[h] ? fpx : 0;
      const int img = fc / hw, rem = fc - img * hw, y = rem / a.W, x = rem - y * a.W;
      const size_t at = (size_t)(y + PADY) * a.in_wp + (x + PADX);
      // F16IN: a pixel of a C8 group is 16 bytes = four 4-byte units; the lane's four channels are its low or high half
      const size_t pix = F16IN ? ((size_t)(lane >> 5) * plane + at) * 4 + ((lane >> 4) & 1) * 2 : at + (size_t)(lane >> 4) * plane;
      src[h][0] = a.in_det + (size_t)img * a.det_in_per_image + pix;
      src[h][1] = a.in_desc + (size_t)img * a.desc_in_per_image + pix;
    }
  }
  __device__ __forceinline__ void fetch(int c) {
    if (HEADS_ABL & 4) return;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if ((v & 1) >= nh) continue;
#pragma unroll
      for (int head = 0; head < 2; ++head)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if constexpr (F16IN) {   // input channels 64 c + 16 (2 i + (v >> 1)) + 4 (lane >> 4) + e: C8 group 8 c + 2 (2 i + (v >> 1)) + (lane >> 5)
            hv[v][head * 2 + i] = *reinterpret_cast<const heads_u2 *>(src[v & 1][head] + (size_t)(8 * c + 2 * (2 * i + (v >> 1))) * plane * 4);
          } else {
#pragma unroll
            for (int e = 0; e < 4; ++e) sv[v][head * 2 + i][e] = src[v & 1][head][(size_t)(64 * c + 16 * (2 * i + (v >> 1)) + 4 * e) * plane];
          }
        }
    }
  }
  __device__ __forceinline__ void store(float *sx) const {
    if (HEADS_ABL & 4) return;
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if ((v & 1) >= nh) continue;
#pragma unroll
      for (int head = 0; head < 2; ++head)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const size_t slot = (size_t)((head * 4 + 2 * i + (v >> 1)) * 2 + (v & 1)) * 64 + lane;   // operand order: [head][s4 of the chunk][tile][lane]
          if constexpr (F16IN) {   // four halves = the f16 matrix instruction's operand, as they come
            reinterpret_cast<heads_u2 *>(sx)[slot] = ok[v & 1] ? hv[v][head * 2 + i] : heads_u2{0u, 0u};
          } else {
            reinterpret_cast<heads_f4 *>(sx)[slot] = ok[v & 1] ? sv[v][head * 2 + i] : heads_f4{0.f, 0.f, 0.f, 0.f};
          }
        }
    }
  }
};

// one barrier per chunk.  Only LDS traffic is ordered (lgkmcnt): the global stores of a step's results are not waited for
__device__ __forceinline__ void heads_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// One step of one computing wave: four chunks, the results.  Every wave of the workgroup passes the same barriers -- one per chunk --
// each in its own instantiation (the branch that selects it is wave-uniform); the loader's are in the kernel body.
// the weights of a step's first group of four k-steps, for every unit slot of the wave that exists
template <bool F16>
__device__ __forceinline__ void heads_first_weights(const HeadsArgs &a, const int u0, const int lane, typename HeadsOperand<F16>::T (&wa)[6]) {
  typedef typename HeadsOperand<F16>::T op_t;
  const op_t *wp4 = reinterpret_cast<const op_t *>(a.wpack) + (size_t)u0 * 16 * 64 + lane;
#pragma unroll
  for (int s = 0; s < 6; ++s)
    if (u0 + s < HEADS_UNITS) wa[s] = (HEADS_ABL & 2) ? heads_abl_operand<F16>() : wp4[(s * 16) * 64];
}

template <unsigned HM, bool W0, bool F16>
__device__ __forceinline__ void heads_wave_step(const HeadsArgs &a, float *smem, int &buf, const int u0, const int lane, const int w, const int t, const int npx,
                                                const bool more, typename HeadsOperand<F16>::T (&wa)[6]) {
  float *sred = smem + 2 * HEADS_CHUNK_FLOATS;
  heads_f4 acc[6][2];
#pragma unroll
  for (int s = 0; s < 6; ++s)
#pragma unroll
    for (int h = 0; h < 2; ++h) acc[s][h] = heads_f4{0.f, 0.f, 0.f, 0.f};
#pragma unroll 1
  for (int c = 0; c < 4; ++c) {
    heads_wave_chunk<HM, W0, F16>(a, smem + buf * HEADS_CHUNK_FLOATS, u0, lane, c, acc, wa);
    if (c == 3) heads_wave_norms<HM, W0, F16>(a, sred, u0, lane, acc);
    heads_barrier();   // the buffer just read may be refilled, the other one may be read, the partial norms are complete
    buf ^= 1;
  }
  if (more) heads_first_weights<F16>(a, u0, lane, wa);   // the next step's first weights: in front of this step's stores (vmcnt is in order)
  heads_wave_finish<HM, W0, F16>(a, sred, u0, lane, npx, t, acc);
  // (the next step's partial norms are written three barriers from here: `sred` is free by then)
}

constexpr int HEADS_THREADS = 320;   // four computing waves + the loader

template <bool F16IN = false>   // F16IN: the activations are C8 fp16 (FP16 engines): products on the f16 matrix instruction, fp32 accumulation and epilogue (the chunk buffers are then half full)
__global__ __launch_bounds__(HEADS_THREADS) void heads_fused_kernel(const HeadsArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];   // two chunk buffers, then [wave 4][tile 2][px 16] partial squared norms
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int npx = a.batch * a.H * a.W, ntiles = (npx + 15) >> 4;
  // tiles of this workgroup (one per CU): an equal share [tb, te) of the flat tile sequence, in steps of two
  const int tb = (int)((long long)blockIdx.x * ntiles / gridDim.x), te = (int)((long long)(blockIdx.x + 1) * ntiles / gridDim.x);
  if (tb >= te) return;
  int buf = 0;
  if (w == 4) {   // ---- the loader: chunk k + 1 is fetched and stored while the computing waves multiply chunk k
    HeadsLoader<F16IN> ld;
    ld.plane = (size_t)a.in_hp * a.in_wp;
    ld.lane = lane;
    ld.pixels(a, tb, te - tb >= 2 ? 2 : 1, npx);
    ld.fetch(0);
    ld.store(smem);
    heads_barrier();
    for (int t = tb; t < te; t += 2) {
      const bool more = t + 2 < te;
#pragma unroll 1
      for (int c = 0; c < 4; ++c) {
        if (c == 3 && more) ld.pixels(a, t + 2, te - (t + 2) >= 2 ? 2 : 1, npx);
        if (c < 3 || more) {
          ld.fetch(c < 3 ? c + 1 : 0);
          ld.store(smem + (buf ^ 1) * HEADS_CHUNK_FLOATS);
        }
        heads_barrier();
        buf ^= 1;
      }
    }
    return;
  }
  const int u0 = w == 0 ? 0 : w == 1 ? 5 : w == 2 ? 11 : 16;
  typename HeadsOperand<F16IN>::T wa[6];
  heads_first_weights<F16IN>(a, u0, lane, wa);
  heads_barrier();   // the first chunk is in LDS
  for (int t = tb; t < te; t += 2) {
    const int nh = te - t >= 2 ? 2 : 1;
    const bool more = t + 2 < te;
    // the instantiation that owns this wave's (tile, unit) pairs: wave-uniform branch
    if (nh == 2) {
      if (w == 0) heads_wave_step<HEADS_A2, true, F16IN>(a, smem, buf, u0, lane, w, t, npx, more, wa);
      else if (w == 1) heads_wave_step<HEADS_B2, false, F16IN>(a, smem, buf, u0, lane, w, t, npx, more, wa);
      else heads_wave_step<HEADS_C2, false, F16IN>(a, smem, buf, u0, lane, w, t, npx, more, wa);
    } else {
      if (w == 0) heads_wave_step<HEADS_A1, true, F16IN>(a, smem, buf, u0, lane, w, t, npx, more, wa);
      else if (w == 1) heads_wave_step<HEADS_B1, false, F16IN>(a, smem, buf, u0, lane, w, t, npx, more, wa);
      else heads_wave_step<HEADS_C1, false, F16IN>(a, smem, buf, u0, lane, w, t, npx, more, wa);
    }
  }
}

}  // namespace spvo
