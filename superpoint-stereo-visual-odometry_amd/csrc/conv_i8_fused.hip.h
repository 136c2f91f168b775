// conv_i8_fused.hip.h -- INT8 engines (BASELINE config 5), MobileNet blocks as ONE launch: depthwise 3x3 -> ReLU -> requantise ->
// pointwise 1x1 (v_mfma_i32_32x32x32_i8) -> ReLU [-> BatchNorm -> ReLU] [-> 2x2 max-pool] -> requantise; and, for the first block
// of the sp_mbv1 graph, the fp32 stem in front of it as well: 3x3 conv 1 -> 1, ReLU, 1x1 conv 1 -> 64, ReLU, BatchNorm, ReLU,
// quantise (ops 0..3 of the plan: one launch that reads the one-channel input plane and writes the pooled 64-channel tensor).
//
// Why: as separate launches every block wrote its depthwise result (one full int8 plane set) to HBM and read it straight back, and
// the depthwise kernel ran at 1.2 TB/s: at 360 x 1176 ops 1..3 moved 4 x 54 MB in 27 + 88 + 47 us.  Fused, the depthwise result of a
// tile lives in LDS in C16 order -- which IS the B operand of the matrix instruction (lane = one pixel x 16 consecutive k) -- and the
// stem's 64-channel full-resolution tensor never exists.  The block is then bound by vector instructions (requantisation chains: eight
// separately rounded operations per value), so the depthwise multiplies are taken off the vector pipe: they run on the matrix cores too,
// against a DIAGONAL weight operand (below), and the requantised bytes are packed by v_cvt_pk_u8_f32 (integers in [0, 127] by then).
//
// The arithmetic is oracle/net_int8.py's, bit for bit (tests/test_gpu_network.py::test_int8_engine_is_bit_exact): exact int32
// accumulation; r = fma(f32(acc), m, bias); ReLU; [r = fma(r, bn_scale, bn_shift); ReLU]; [2x2 max]; q = min(rint(r), 127) -- one fused
// multiply-add per affine, 1 / s_out folded into the chain's last affine by the loader (round 6; rounds 2-5: separately rounded multiplies and
// adds and a final multiply, nine vector instructions per value of the stem where there are five now).  The fp32 stem: bias first, then one
// fused multiply-add per tap in row-major order.  The synchronous entry points (spvo_forward / spvo_debug_tensor) pass pointers for the tensors a fused block
// skips (stem plane, stem output, depthwise output): the kernel then stores them too, so the test sees every tensor of the graph
// computed by the kernels the pipeline runs.
//
// Workgroup = 4 computing waves = 8 x 32 pixels x ALL output channels (one or two tiles of 64), persistent over tiles, + a LOADER wave
// that brings the next tile's input halo into LDS (LDS-DMA; the stem variant: the raw one-channel tile) under the current tile's
// stages.  A wave of its own because `s_waitcnt vmcnt` counts a wave's loads and stores in order: a computing wave that waited for its
// next input would wait for the acknowledgement of its previous tile's output stores as well (csrc/heads.hip.h has the measurement).
// Pointwise weights stay resident in LDS.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "conv_i8.hip.h"

namespace spvo {

struct DwPwArgs8 {
  const int8_t *in = nullptr;        // C16 input of the depthwise layer, image 0, group 0 (not with STEM)
  const float *in_f32 = nullptr;     // STEM: the network's one-channel fp32 input plane (padded plane), image 0
  size_t in_per_image = 0;           // bytes (C16) / floats (STEM)
  int hp = 0, wp = 0;                // padded geometry of the block's input level (depthwise in and out)
  int H = 0, W = 0;
  // STEM (ops 0, 1 of the plan): 3x3 1 -> 1 + ReLU, then 1x1 1 -> 64 + ReLU + BatchNorm + ReLU, quantised with inv_s_stem
  const float *w0 = nullptr, *b0 = nullptr;       // [9], [1]
  const float *w1 = nullptr, *b1 = nullptr, *bn1_scale = nullptr, *bn1_shift = nullptr;   // [64] each
  float inv_s_stem = 0.f;
  // depthwise layer
  const int *dw_wsel = nullptr;      // [G][9][16]: quantised weight of (channel, tap) shifted into byte (channel & 3): the dot4 operand
  const float *dw_qm = nullptr, *dw_bias = nullptr;   // [C]
  float inv_s_dw = 0.f;
  // pointwise layer
  const int8_t *pw_w = nullptr;      // [co_tiles][G][co 64][16]  (pack_conv_weights_i8 with ks = 1, ckg = G)
  const float *pw_qm = nullptr, *pw_bias = nullptr, *bn_scale = nullptr, *bn_shift = nullptr;   // [co_tiles * 64]
  float inv_s_out = 0.f;
  int8_t *out = nullptr;             // C16 output (pooled: level + 1), image 0
  size_t out_per_image = 0;          // bytes
  int out_hp = 0, out_wp = 0, cout = 0, co_tiles = 0;
  int tiles_x = 0, tiles_y = 0, batch = 0;
  // tensors a fused block skips, stored only when a pointer is given (synchronous entry points)
  float *dbg_stem_plane = nullptr;   // STEM: output of op 0 (fp32 padded plane), image 0
  size_t dbg_stem_plane_per_image = 0;
  int8_t *dbg_stem_out = nullptr;    // STEM: output of op 1 (C16), image 0
  int8_t *dbg_dw_out = nullptr;      // output of the depthwise layer (C16), image 0
  size_t dbg_c16_per_image = 0;      // bytes per image of those two
};

template <int G, bool STEM>
struct DwPwTile {
  static constexpr int TH = 8, TW = 32, LH = 10, LW = 34;
  static constexpr int IN_P = G * LH * LW;                 // 16-byte pieces of one input halo tile
  static constexpr int IN_P_PAD = (IN_P + 63) / 64 * 64;   // ... rounded up to whole LDS-DMA instructions (64 lanes x 16 bytes)
  static constexpr int IN_BUFS = STEM ? 1 : 2;             // C16 input: the next tile lands while the current one is read; STEM: the tile is computed in place
  static constexpr int RAW_F = STEM ? 12 * 36 : 0;         // STEM: raw input tile with a halo of two (floats), double-buffered
  static constexpr int W_P_MAX = 2 * G * CO_TILE;          // pointwise weights: at most 128 output channels x 16 G bytes
  static constexpr int PAR_F = 4 * 128 + 2 * 16 * G + 9 * 16 * G;   // pointwise qm / bias / BatchNorm scale / shift [128] each; depthwise qm / bias [C]; depthwise dot operands [G][9][16]
  static constexpr int LDS_BYTES = (IN_BUFS * IN_P_PAD + W_P_MAX) * 16 + PAR_F * 4 + 2 * RAW_F * 4;
};
constexpr int DWPW_THREADS = 320;   // four computing waves + the loader

// quantised depthwise weights [C][9] -> the dot4 operands [G][9][16]: byte (c & 3) of entry c holds the weight, the other bytes 0
inline std::vector<int> pack_dw_wsel(const int8_t *wq, int channels) {
  std::vector<int> out((size_t)(channels / 16) * 9 * 16, 0);
  for (int c = 0; c < channels; ++c)
    for (int t = 0; t < 9; ++t)
      out[((size_t)(c / 16) * 9 + t) * 16 + (c & 15)] = (int)((unsigned)(uint8_t)wq[(size_t)c * 9 + t] << (8 * (c & 3)));
  return out;
}

// Layer constants that are the same for every lane (depthwise weights, per-channel scales and biases indexed by loop counters) are read
// through the CONSTANT address space: the compiler then fetches them with scalar loads into SGPRs (s_load_dwordx16) instead of one
// vector load per lane -- with plain pointers it cannot (the kernel also stores to global memory), and the depthwise stage spent its time
// waiting for 36 broadcast vector loads per 16 channels
typedef float dwpw_f4 __attribute__((ext_vector_type(4)));
typedef const int __attribute__((address_space(4))) *dwpw_cint;
typedef const float __attribute__((address_space(4))) *dwpw_cfloat;

// q = min(rint(x), 127) of a NON-NEGATIVE x (everything here is behind a ReLU), packed into byte `e` of `u`: v_cvt_pk_u8_f32 rounds to
// nearest even itself (tools/cvt_probe.hip: all 4096 sixteenths in [-1, 255], ties included) and saturates at 0 and 255, and
// min(rint(x), 127) = rint(min(x, 127)) because 127 is an integer: two instructions instead of the oracle's rint, clip, convert, pack
__device__ __forceinline__ unsigned dwpw_pack_q(float x, int e, unsigned u) { return __builtin_amdgcn_cvt_pk_u8_f32(fminf(x, 127.f), e, u); }

#ifndef DWPW_ABL
#define DWPW_ABL 0   // measurement builds only (make EXTRA=-DDWPW_ABL=n): 1 no stem stage, 2 no depthwise stage, 4 no pointwise epilogue (results are then wrong)
#endif

__device__ __forceinline__ void dwpw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }   // LDS traffic only: global stores are never waited for

template <int G, bool STEM, bool POOL, int EPI>
__global__ __launch_bounds__(DWPW_THREADS) void dwpw_i8_kernel(const DwPwArgs8 a) {
  using T = DwPwTile<G, STEM>;
  constexpr int TH = T::TH, TW = T::TW, LH = T::LH, LW = T::LW;
  static_assert(!STEM || G == 4, "the stem feeds 64 channels");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_f[];
  i32x4 *s_in = reinterpret_cast<i32x4 *>(smem_f);                      // [buf][G][LH][LW] (+ pad)
  i32x4 *s_w = s_in + T::IN_BUFS * T::IN_P_PAD;                         // [co_tiles][G][64]
  float *s_par = reinterpret_cast<float *>(s_w + T::W_P_MAX);           // [4][128]: pointwise qm, bias, BatchNorm scale, shift per output channel
  float *s_dwpar = s_par + 4 * 128;                                      // [2][16 G]: depthwise qm, bias per channel
  int *s_wsel = reinterpret_cast<int *>(s_dwpar + 2 * 16 * G);           // [G][9][16]: depthwise weights, each shifted into its channel's byte of a dword
  float *s_raw = reinterpret_cast<float *>(s_wsel + 9 * 16 * G);         // STEM: [buf 2][12][36]

  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t plane = (size_t)a.hp * a.wp, out_plane = (size_t)a.out_hp * a.out_wp;
  const int n_tiles = a.tiles_x * a.tiles_y * a.batch;
  if ((int)blockIdx.x >= n_tiles) return;

  struct TileRef { int x0, y0, img; };
  auto decode = [&](int id) {
    TileRef t;
    t.x0 = (id % a.tiles_x) * TW;
    id /= a.tiles_x;
    t.y0 = (id % a.tiles_y) * TH;
    t.img = id / a.tiles_y;
    return t;
  };
  // pointwise weights and the pointwise layer's per-channel constants: resident for the whole launch (visible behind the first barrier)
  for (int p = tid; p < a.co_tiles * G * CO_TILE; p += DWPW_THREADS) s_w[p] = reinterpret_cast<const i32x4 *>(a.pw_w)[p];
  for (int p = tid; p < a.co_tiles * CO_TILE; p += DWPW_THREADS) {
    s_par[p] = a.pw_qm[p];
    s_par[128 + p] = a.pw_bias[p];
    if constexpr (EPI == 1) { s_par[256 + p] = a.bn_scale[p]; s_par[384 + p] = a.bn_shift[p]; }
  }
  for (int p = tid; p < 16 * G; p += DWPW_THREADS) { s_dwpar[p] = a.dw_qm[p]; s_dwpar[16 * G + p] = a.dw_bias[p]; }
  for (int p = tid; p < 9 * 16 * G; p += DWPW_THREADS) s_wsel[p] = a.dw_wsel[p];
  const dwpw_cfloat c_w0 = (dwpw_cfloat)a.w0, c_b0 = (dwpw_cfloat)a.b0, c_w1 = (dwpw_cfloat)a.w1, c_b1 = (dwpw_cfloat)a.b1, c_bn1s = (dwpw_cfloat)a.bn1_scale, c_bn1h = (dwpw_cfloat)a.bn1_shift;

  if (wave == 4) {
    // ================================================================ the loader
    // tile k + 1 is fetched while the computing waves work on tile k; it passes the same barriers (one at the top of a tile, STEM: one
    // behind the stem stage) and arrives at a tile's top barrier with that tile's input landed
    // C16 halo tile: LDS-DMA, 16 bytes per lane; the planes' zero border is the convolution's padding.  Where a lane's pieces lie relative to
    // the tile's first piece does not depend on the tile: worked out once (two divisions per piece -- per tile they were most of this
    // wave's time, and the computing waves waited for it at every tile's top barrier)
    constexpr int NPC = STEM ? 1 : T::IN_P_PAD / 64;
    unsigned piece_off[NPC];
    if constexpr (!STEM) {
#pragma unroll
      for (int it = 0; it < NPC; ++it) {
        const int idx = min(it * 64 + lane, T::IN_P - 1);   // (the last instruction's surplus lanes re-load the last piece into the pad)
        const int g = idx / (LH * LW), rem = idx - g * (LH * LW), r = rem / LW, c = rem - r * LW;
        piece_off[it] = (unsigned)(((size_t)g * plane + (size_t)r * a.wp + c) * 16);
      }
    }
    auto fetch = [&](const TileRef &t, int b) {
      if constexpr (!STEM) {
        const int8_t *base = a.in + (size_t)t.img * a.in_per_image + ((size_t)(t.y0 + PADY - 1) * a.wp + (t.x0 + PADX - 1)) * 16;
#pragma unroll
        for (int it = 0; it < NPC; ++it)
          glds16(reinterpret_cast<const float *>(base + piece_off[it]), reinterpret_cast<float *>(s_in + (size_t)b * T::IN_P_PAD + it * 64));
      } else {
        // raw fp32 tile with a halo of two, bounds-checked (the plane has one row of padding above the image, the stem needs two)
        const float *pl = a.in_f32 + (size_t)t.img * a.in_per_image;
        float v[7];
#pragma unroll
        for (int i = 0; i < 7; ++i) {
          const int p = i * 64 + lane, r = p / 36, c = p - r * 36, y = t.y0 - 2 + r, x = t.x0 - 2 + c;
          v[i] = (p < 12 * 36 && y >= 0 && y < a.H && x >= 0 && x < a.W) ? pl[(size_t)(y + PADY) * a.wp + (x + PADX)] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < 7; ++i)
          if (i * 64 + lane < 12 * 36) s_raw[b * T::RAW_F + i * 64 + lane] = v[i];
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    int buf = 0;
    fetch(decode(blockIdx.x), 0);
    for (int tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x, buf ^= 1) {
      dwpw_barrier();                                              // top of tile k: its input is in LDS
      if (tile_id + (int)gridDim.x < n_tiles) fetch(decode(tile_id + gridDim.x), buf ^ 1);
      if constexpr (STEM) dwpw_barrier();
    }
    return;
  }

  // ================================================================ the computing waves
  int buf = 0;
  for (int tile_id = blockIdx.x; tile_id < n_tiles; tile_id += gridDim.x, buf ^= 1) {
    const TileRef cur = decode(tile_id);
    dwpw_barrier();   // this tile's input has landed; the previous tile's readers of the stem's s_in are done
    const int ctid = tid;   // 0 .. 255
    i32x4 *tin = s_in + (size_t)(STEM ? 0 : buf) * T::IN_P_PAD;

    if constexpr (STEM) {
      if (!(DWPW_ABL & 1)) {
      // ---- ops 0 and 1 on the tile's 10 x 34 halo pixels: x1 = ReLU(3x3 stem), 64 channels of ReLU(BN(ReLU(w1 x1 + b1))) quantised to C16.
      // Pixels outside the image are the depthwise layer's zero padding (q = 0), whatever the stem would give there
      const float *raw = s_raw + buf * T::RAW_F;
#pragma unroll 1
      for (int p = ctid; p < LH * LW; p += 256) {
        const int r = p / LW, c = p - r * LW, y = cur.y0 - 1 + r, x = cur.x0 - 1 + c;
        const bool inside = y >= 0 && y < a.H && x >= 0 && x < a.W;
        float x1 = c_b0[0];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) x1 = __builtin_fmaf(c_w0[ky * 3 + kx], raw[(r + ky) * 36 + c + kx], x1);
        x1 = fmaxf(x1, 0.f);
        const bool own = inside && r >= 1 && r <= TH && c >= 1 && c <= TW;   // interior of the tile: this workgroup's pixel
        if (a.dbg_stem_plane && own) a.dbg_stem_plane[(size_t)cur.img * a.dbg_stem_plane_per_image + (size_t)(y + PADY) * a.wp + (x + PADX)] = x1;
#pragma unroll 1
        for (int g = 0; g < 4; ++g) {
          i32x4 pk;
#pragma unroll
          for (int d = 0; d < 4; ++d) {
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              const int ch = 16 * g + 4 * d + e;
              // five instructions per value (rounds 2-5: nine): fma, max, fma with 1 / s folded into the BatchNorm constants, min, convert --
              // the second ReLU is the conversion's saturation at 0
              const float v = fmaxf(__builtin_fmaf(c_w1[ch], x1, c_b1[ch]), 0.f);
              u = dwpw_pack_q(__builtin_fmaf(v, c_bn1s[ch], c_bn1h[ch]), e, u);
            }
            pk[d] = inside ? (int)u : 0;
          }
          tin[(g * LH + r) * LW + c] = pk;
          if (a.dbg_stem_out && own)
            reinterpret_cast<i32x4 *>(a.dbg_stem_out + (size_t)cur.img * a.dbg_c16_per_image)[(size_t)g * plane + (size_t)(y + PADY) * a.wp + (x + PADX)] = pk;
        }
      }
      }
      dwpw_barrier();
    }

    // ---- depthwise 3x3 ON THE MATRIX CORES: for 32 channels (two C16 groups) and one tap, D[c][px] += diag(w[c][tap]) x X[c'][px + tap] --
    // v_mfma_i32_32x32x32_i8 with a DIAGONAL A operand (row c holds its weight at k = c, zeros elsewhere) and the input tile, shifted by
    // the tap, as B (lane = pixel, 16 consecutive k = the bytes of its C16 piece: read straight from LDS).  1/32 of the multiplies are
    // useful, and it is still twice the rate of v_dot4c on the vector pipe -- which is left to the requantisation.  Wave w: rows 2 w,
    // 2 w + 1 of the tile, all channels: what it produces here is exactly what it consumes as the pointwise layer's B operand below, so
    // the depthwise result stays in registers (two v_permlane32_swap per 32 channels put a pixel's 16 bytes of a group into one lane).
    i32x4 bvdw[2][G / 2];
    {
      const int diag_dword = ((j >> 4) == half) ? ((j & 15) >> 2) : -1;   // lane (row j of the 32, k-half `half`): which of its four k dwords holds the diagonal byte
#pragma unroll
      for (int s2 = 0; s2 < G / 2; ++s2) {
        i32x16 dacc[2];
#pragma unroll
        for (int rr = 0; rr < 2; ++rr)
#pragma unroll
          for (int q = 0; q < 16; ++q) dacc[rr][q] = 0;
#pragma unroll
        for (int t = 0; t < ((DWPW_ABL & 2) ? 1 : 9); ++t) {
          const int wv = s_wsel[((2 * s2 + (j >> 4)) * 9 + t) * 16 + (j & 15)];   // channel 32 s2 + j, tap t: the weight in byte (j & 3)
          i32x4 av;
#pragma unroll
          for (int d = 0; d < 4; ++d) av[d] = diag_dword == d ? wv : 0;
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const i32x4 bv = tin[((2 * s2 + half) * LH + 2 * wave + rr + t / 3) * LW + j + t % 3];
            dacc[rr] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av, bv, dacc[rr], 0, 0, 0);
          }
        }
        // requantisation: register q of the lane (pixel j) = channel 32 s2 + (q & 3) + 8 (q >> 2) + 4 half
#pragma unroll
        for (int rr = 0; rr < 2; ++rr) {
          unsigned dq[4];
#pragma unroll
          for (int qq = 0; qq < 4; ++qq) {
            const int ch = 32 * s2 + 8 * qq + 4 * half;
            const dwpw_f4 qm4 = *reinterpret_cast<const dwpw_f4 *>(s_dwpar + ch), bi4 = *reinterpret_cast<const dwpw_f4 *>(s_dwpar + 16 * G + ch);
            unsigned u = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
              u = dwpw_pack_q(__builtin_fmaf((float)dacc[rr][4 * qq + e], qm4[e], bi4[e]), e, u);   // (1 / s folded into qm, bias; ReLU = the conversion's saturation at 0)
            }
            dq[qq] = u;
          }
          // half 0 holds channels {0-3, 8-11, 16-19, 24-27} + 32 s2 of its pixel, half 1 the other four quadruples: after two swaps
          // lane (j, 0) holds the 16 bytes of group 2 s2, lane (j, 1) those of group 2 s2 + 1 -- the matrix instruction's B operand
          typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
          const u32x2s sw0 = __builtin_amdgcn_permlane32_swap(dq[0], dq[2], false, false);
          const u32x2s sw1 = __builtin_amdgcn_permlane32_swap(dq[1], dq[3], false, false);
          bvdw[rr][s2] = i32x4{(int)sw0[0], (int)sw0[1], (int)sw1[0], (int)sw1[1]};
          const int y = cur.y0 + 2 * wave + rr, x = cur.x0 + j;
          if (a.dbg_dw_out && y < a.H && x < a.W)
            reinterpret_cast<i32x4 *>(a.dbg_dw_out + (size_t)cur.img * a.dbg_c16_per_image)[(size_t)(2 * s2 + half) * plane + (size_t)(y + PADY) * a.wp + (x + PADX)] = bvdw[rr][s2];
        }
      }
    }

    // ---- pointwise 1x1 on the matrix cores + epilogue, one tile of 64 output channels after the other.
    // Wave w: rows 2 w, 2 w + 1 of the tile; lane: pixel column j, k-half `half` (the 16 channels of group 2 s + half)
#pragma unroll 1
    for (int ct = 0; ct < a.co_tiles; ++ct) {
      i32x16 acc[2][2];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < 2; ++n)
#pragma unroll
          for (int q = 0; q < 16; ++q) acc[m][n][q] = 0;
#pragma unroll
      for (int s2 = 0; s2 < G / 2; ++s2) {
        i32x4 av[2];
#pragma unroll
        for (int m = 0; m < 2; ++m) av[m] = s_w[(ct * G + 2 * s2 + half) * CO_TILE + 32 * m + j];
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) acc[m][rr] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[m], bvdw[rr][s2], acc[m][rr], 0, 0, 0);
      }
      // epilogue: register q of row block m = channel 32 m + (q & 3) + 8 (q >> 2) + 4 half of the tile
      const int co_t = ct * CO_TILE;
      auto tail = [&](int accv, int m, int q) -> float {
        const int co = co_t + 32 * m + (q & 3) + 8 * (q >> 2) + 4 * half;
        // the chain's last affine carries 1 / s_out (folded by the loader); the ReLU behind it is the conversion's saturation at 0
        float v = __builtin_fmaf((float)accv, s_par[co], s_par[128 + co]);
        if constexpr (EPI == 1) v = __builtin_fmaf(fmaxf(v, 0.f), s_par[256 + co], s_par[384 + co]);
        return v;
      };
      int8_t *g_base = a.out + (size_t)cur.img * a.out_per_image + (size_t)ct * (CO_TILE / 16) * out_plane * 16;
      const int groups_valid = (a.cout - co_t + 15) / 16;
      auto store_tile = [&](const float (&v)[16], int m, bool ok, size_t pix) {
#pragma unroll
        for (int gq = 0; gq < 4; ++gq) {
          if (2 * m + (gq >> 1) < groups_valid && ok && !((DWPW_ABL & 4) && v[0] != 12345.f)) {
            unsigned pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk = dwpw_pack_q(v[4 * gq + e], e, pk);
            *reinterpret_cast<unsigned *>(g_base + ((size_t)(2 * m + (gq >> 1)) * out_plane + pix) * 16 + 8 * (gq & 1) + 4 * half) = pk;
          }
        }
      };
      if constexpr (!POOL) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int rr = 0; rr < 2; ++rr) {
            const int y = cur.y0 + 2 * wave + rr, x = cur.x0 + j;
            float v[16];
#pragma unroll
            for (int q = 0; q < 16; ++q) v[q] = tail(acc[m][rr][q], m, q);
            store_tile(v, m, y < a.H && x < a.W, (size_t)(y + PADY) * a.out_wp + (x + PADX));
          }
      } else {
        const int OH = a.H >> 1, OW = a.W >> 1;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
          const int y = (cur.y0 >> 1) + wave, x = (cur.x0 + j) >> 1;
          float v[16];
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const float p2 = fmaxf(tail(acc[m][0][q], m, q), tail(acc[m][1][q], m, q));
            v[q] = fmaxf(p2, __shfl_xor(p2, 1));
          }
          store_tile(v, m, y < OH && x < OW && !(j & 1), (size_t)(y + PADY) * a.out_wp + (x + PADX));
        }
      }
    }
  }
}

}  // namespace spvo
