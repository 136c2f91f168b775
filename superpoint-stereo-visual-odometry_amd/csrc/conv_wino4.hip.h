// conv_wino4.hip.h -- K2x: 3x3 convolution by the Winograd minimal-filtering algorithm F(4x4, 3x3) on the gfx950 fp32 matrix
// cores, bias + ReLU (+ 2x2 max-pool) fused.  fp32 throughout: operands, products and accumulation.
//
// Replaces, like the other conv_*.hip.h, the TensorRT engine the reference enqueues at
// feature_detection_neural_network.cpp:169 for the 3x3 Conv/Relu/MaxPool nodes of the SuperPoint graphs.
//
// Y(4x4) = A^T [ (G g G^T) .* (B^T d B) ] A per 6x6 input patch d (patches overlap by 2) and 3x3 filter g, interpolation
// points {0, +-1, +-2, inf}:
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//   G   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
// 36 multiplies per 4x4 outputs and channel pair instead of 144: 1/4 of the direct method's matrix work, 9/16 of
// F(2x2, 3x3)'s (conv_wino2.hip.h).  The price is numerical: the transforms amplify rounding (coefficients up to 8 instead
// of 1); on this network's tensors the error against a float64 evaluation is 0.4 - 2.7 x the direct kernel's (measured per
// tensor at 360x1176, 376x1240 and 192x640 by tests/test_gpu_network.py::test_winograd_layers_stay_at_fp32_rounding_level, bar
// 4 x + 2e-7: worst tensor 2.5e-6 of its maximum against 0.9e-6 direct and 0.5e-6 for F(2x2)), 40 x inside the 1e-4 bar
// against the oracle.
//
// Structure = conv_wino2.hip.h's (8 waves, two per SIMD, LDS-DMA staging, input transform of the next item between the matrix
// instructions of the current one, one barrier per item) with these differences:
//   * workgroup tile = 64 output channels x 32 Winograd tiles (4 x 8 tiles = 16 rows x 32 columns of output); wave (cq, tb) owns
//     16 channels x 16 tiles x all 36 positions on v_mfma_f32_16x16x4_f32: 144 accumulator registers;
//   * item = (tile, chunk of FOUR input channels): one matrix instruction per position and item; filters U [pos/4 9][cq 4]
//     [lane 64][4] and transformed input V [h 2][q/4 5][tb 2][lane 64][4] in LDS, both read by ds_read_b128 (nine + ten reads per item).
//     The matrix instructions run in the order p' = 18 h + 3 i + c of the positions (i, j = 3 h + c): the half h of the columns
//     that one transforming thread produces is contiguous, so its eighteen stores carry compile-time offsets.  (The 4-byte
//     stores hit every fourth bank -- SQ_LDS_BANK_CONFLICT: 40 % of the LDS cycles -- but pieces of 3 positions + 1 pad written
//     by conflict-free 16-byte stores and read by twelve instead of ten instructions measured 1-2 % SLOWER: stores are the one
//     thing that is nearly free beside an fp32 matrix instruction, reads are not);
//   * the input transform of an item is 128 patches of 6 x 6: TWO threads per patch, rows {0,1,2} / {3,4,5} for the row pass,
//     nine v_permlane32_swap exchanges, columns {0,1,2} / {3,4,5} for the column pass (the partner sits 32 lanes away, so the
//     swap leaves "rows 0-2" and "rows 3-5" in the same registers of both halves: no selects); waves 0-3 transform the even
//     items, waves 4-7 the odd ones, each transform in two halves in two consecutive items (rows + row pass two items ahead,
//     exchange + column pass + stores one item ahead; 18 registers carry it across the barrier), so that in every item BOTH
//     waves of a SIMD do half a transform;
//   * a wave's issue priority falls as it advances through an item (WINO4_PRIO below), so that the two waves of a SIMD reach the
//     item's barrier together;
//   * the inverse transform (36 -> 16 per output channel and tile) runs in registers, as before.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <vector>
#include "conv_mfma.hip.h"

#ifndef WINO4_LEAD      // tuning (tools/wino_bench.hip): operand reads LEAD slots ahead; transform step s in slot XF_START + s * XF_STRIDE
#define WINO4_LEAD 6
#endif
// The input transform of an item is done in two halves, in two consecutive items, by the same waves: H1 (steps 0..5: the patch rows
// from the raw tile, row pass) two items ahead, H2 (steps 6..9: exchange, column pass, stores into V) one item ahead.  In every item one
// group of waves does H1 of item k + 2 and the other H2 of item k + 1: the same extra work for both waves of a SIMD.  (With a whole
// transform per item by alternating groups, the transforming wave's matrix stream took 3900 cycles and the other's 2630, which then
// waited 1300 cycles at the barrier while the transforming wave ran alone: tools/wino_bench -DWINO_STAMPS -DWINO_STAMPS_ROLES.)
// A wave's issue priority falls as it advances through an item (s_setprio 3, 2, 1, 0 at slots 0, 9, 18, 27): of the two waves of a SIMD
// the one that is BEHIND wins the matrix pipe.  Without it the older wave of each SIMD (waves 0-3) wins every arbitration, finishes
// its item ~1000 cycles ahead and stands at the barrier while the younger one runs on alone with every stall of its own exposed
// (stamps by wave, tools/wino_bench -DWINO_STAMPS: barrier wait 1180 / 180 cycles per item -> 560 / 200).  0: off; 2..4: other
// schedules that measured no better.
#ifndef WINO4_PRIO
#define WINO4_PRIO 1
#endif
#ifndef WINO4_H1_STEPS
#define WINO4_H1_STEPS 6   // steps of the first half (6: up to the row pass; 7: + the exchange -- no better)
#endif
#ifndef WINO4_H1_START
#define WINO4_H1_START 3
#endif
#ifndef WINO4_H1_STRIDE
#define WINO4_H1_STRIDE 5
#endif
#ifndef WINO4_H2_START
#define WINO4_H2_START 4
#endif
#ifndef WINO4_H2_STRIDE
#define WINO4_H2_STRIDE 8
#endif
#ifndef WINO4_ABL
#define WINO4_ABL 0   // measurement builds only (tools/wino_bench.hip): 1 no input transform, 2 no filter staging, 4 no raw staging, 8 operands read once, 16 no stores
#endif

#ifndef SPVO_STATIC_BANDS
#define SPVO_STATIC_BANDS 1   // measurement builds: 0 = tile = blockIdx.x in launches without a.sched
#endif

namespace spvo {

// TB = tile blocks of 16 Winograd tiles (2 x 8 tiles = 8 rows x 32 columns of output) a workgroup covers: 2 = the 8-wave form (16 x 32
// outputs), 1 = the 4-wave form (8 x 32 outputs, one wave per SIMD) for layers whose 16 x 32 tiles would leave half the CUs without a
// workgroup (conv4a / conv4b: 45 x 147 = 15 tiles per image and 64 output channels).  Same filter slabs, same per-wave work.
template <int TB>
struct Wino4TileT {
  static constexpr int CK = 4, TH = 8 * TB, TW = 32, LW = TW + 8, LH = TH + 2, NT = 16 * TB, NTH = 256 * TB;
  static constexpr int IN_FLOATS = CK * LH * LW;        // 2880 (1600): raw halo tile of one chunk, row = x0-4 .. x0+35
  static constexpr int U_FLOATS = 36 * CK * CO_TILE;    // 9216: one filter slab
  static constexpr int V_FLOATS = 10 * TB * 64 * 4;     // 5120 (2560): transformed input of one chunk, per column half 18 positions in 5 pieces of four
  static constexpr int RAW_OFF = 0, U_OFF = 2 * IN_FLOATS, V_OFF = U_OFF + 2 * U_FLOATS;
  static constexpr int LDS_BYTES = (V_OFF + 2 * V_FLOATS) * 4 + 16;   // 137 744 (107 024) (+ the slot through which a tile's successor is published)
};
using Wino4Tile = Wino4TileT<2>;

// OIHW weights + bias -> slabs [co_tile][chunk][p'/4 9][cq 4][lane 64][4] of U = G g G^T (double), then [co_tiles * 64] biases.
// p' = 18 (j / 3) + 3 i + j % 3 (the kernel's instruction order); lane = 16 (ci & 3) + (co & 15): the A operand of
// v_mfma_f32_16x16x4_f32 (row co & 15, k = ci).
inline std::vector<float> pack_conv_weights_wino4(const float *w, const float *bias, int cout, int cin) {
  constexpr int CK = Wino4Tile::CK;
  const int co_tiles = (cout + CO_TILE - 1) / CO_TILE, nch = cin / CK;
  std::vector<float> out((size_t)co_tiles * nch * Wino4Tile::U_FLOATS + (size_t)co_tiles * CO_TILE, 0.f);
  static const double G[6][3] = {{0.25, 0, 0}, {-1.0 / 6, -1.0 / 6, -1.0 / 6}, {-1.0 / 6, 1.0 / 6, -1.0 / 6},
                                 {1.0 / 24, 1.0 / 12, 1.0 / 6}, {1.0 / 24, -1.0 / 12, 1.0 / 6}, {0, 0, 1}};
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int o = 0; o < CO_TILE; ++o) {
      const int co = ct * CO_TILE + o;
      if (co >= cout) continue;
      out[(size_t)co_tiles * nch * Wino4Tile::U_FLOATS + co] = bias[co];
      for (int ci = 0; ci < cin; ++ci) {
        const float *g = w + ((size_t)co * cin + ci) * 9;
        double t[6][3];
        for (int a = 0; a < 6; ++a)
          for (int k = 0; k < 3; ++k) t[a][k] = G[a][0] * g[0 * 3 + k] + G[a][1] * g[1 * 3 + k] + G[a][2] * g[2 * 3 + k];
        float *slab = out.data() + ((size_t)ct * nch + ci / CK) * Wino4Tile::U_FLOATS;
        const int lane = 16 * (ci % CK) + (o & 15), cq = o >> 4;
        for (int a = 0; a < 6; ++a)
          for (int b = 0; b < 6; ++b) {
            const int pos = 18 * (b / 3) + 3 * a + b % 3;
            slab[(((pos >> 2) * 4 + cq) * 64 + lane) * 4 + (pos & 3)] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
          }
      }
    }
  return out;
}

// 1-D transforms.  in6: B^T x (12 operations), out4: A^T m (10 operations).
__device__ __forceinline__ void wino4_in6(const float x0, const float x1, const float x2, const float x3, const float x4, const float x5, float (&o)[6]) {
  const float e = fmaf(-4.f, x2, x4), od = fmaf(-4.f, x1, x3), e2 = x4 - x2, o2 = x3 - x1;
  o[0] = fmaf(4.f, x0, fmaf(-5.f, x2, x4));
  o[1] = e + od;
  o[2] = e - od;
  o[3] = fmaf(2.f, o2, e2);
  o[4] = fmaf(-2.f, o2, e2);
  o[5] = fmaf(4.f, x1, fmaf(-5.f, x3, x5));
}
__device__ __forceinline__ void wino4_out4(const float m0, const float m1, const float m2, const float m3, const float m4, const float m5, float (&y)[4]) {
  const float a = m1 + m2, b = m1 - m2, c = m3 + m4, d = m3 - m4;
  y[0] = (m0 + a) + c;
  y[1] = fmaf(2.f, d, b);
  y[2] = fmaf(4.f, c, a);
  y[3] = fmaf(8.f, d, b) + m5;
}

// the same on two output channels at once (v_pk_add_f32 / v_pk_fma_f32: the epilogue has no matrix instructions to hide behind)
typedef float wino4_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wino4_out4(const wino4_f32x2 m0, const wino4_f32x2 m1, const wino4_f32x2 m2, const wino4_f32x2 m3, const wino4_f32x2 m4,
                                           const wino4_f32x2 m5, wino4_f32x2 (&y)[4]) {
  const wino4_f32x2 a = m1 + m2, b = m1 - m2, c = m3 + m4, d = m3 - m4;
  y[0] = (m0 + a) + c;
  y[1] = __builtin_elementwise_fma(wino4_f32x2{2.f, 2.f}, d, b);
  y[2] = __builtin_elementwise_fma(wino4_f32x2{4.f, 4.f}, c, a);
  y[3] = __builtin_elementwise_fma(wino4_f32x2{8.f, 8.f}, d, b) + m5;
}

template <bool POOL, bool RELU, int TAG = 0, int TB = 2>
__global__ __launch_bounds__(256 * TB, TB) void conv_wino4_kernel(const ConvArgs a) {
  using T = Wino4TileT<TB>;
  constexpr int CK = T::CK, LW = T::LW, LH = T::LH, LW4 = LW / 4, NTH = T::NTH;
  constexpr int IN_V4 = T::IN_FLOATS / 4;        // 720 (400) 16-byte pieces per raw tile
  constexpr int U_V4 = T::U_FLOATS / 4;          // 2304 per filter slab
  constexpr int NIT_U = (U_V4 + NTH - 1) / NTH;  // 5, the last one by threads 0..255 (9, all full)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cq = wave & 3, tb = wave >> 2;
  const size_t in_plane = (size_t)a.in_hp * a.in_wp;
  const size_t out_plane = (size_t)a.out_hp * a.out_wp;
  const int n_tiles = a.tiles_x * a.tiles_y * a.co_tiles * a.batch;

  struct TileRef { const float *in_base, *w_base; int x0, y0, ct, img; };
  auto decode = [&](int id) {
    TileRef t;
    const int tx = id % a.tiles_x;
    id /= a.tiles_x;
    const int ty = id % a.tiles_y;
    id /= a.tiles_y;
    t.ct = id % a.co_tiles;
    t.img = id / a.co_tiles;
    t.x0 = tx * T::TW;
    t.y0 = ty * T::TH;
    t.in_base = a.in + ((size_t)t.img * a.in_ctot + a.in_coff) * in_plane + (size_t)(t.y0 + PADY - 1) * a.in_wp + (t.x0 + PADX - 4);
    t.w_base = a.wpack + (size_t)t.ct * a.n_chunks * T::U_FLOATS;
    return t;
  };

  // ---- staging plans (wave-uniform 64-bit base + 32-bit lane offset).  Raw tile: pieces 0..511 by every thread, pieces
  // 512..719 by threads 256.. (waves 4-7, which carry one filter piece less: six LDS-DMA instructions per wave and item everywhere);
  // 4-wave form: pieces 0..255 by every thread, 256..399 by threads 0..143
  auto raw_piece_off = [&](int idx) {
    idx = min(idx, IN_V4 - 1);
    const int ci = idx / (LH * LW4);
    const int rem = idx - ci * (LH * LW4);
    const int r = rem / LW4;
    const int q = rem - r * LW4;
    return 4u * (unsigned)(ci * (int)in_plane + r * a.in_wp + q * 4);
  };
  constexpr int R0 = NTH - 256;   // first thread that carries a second raw piece
  const unsigned roff0 = raw_piece_off(tid), roff1 = raw_piece_off(NTH + (tid - R0));
  const bool raw2 = tid >= R0 && NTH + (tid - R0) < IN_V4;
  auto issue_raw = [&](const TileRef &t, int chunk, float *buf) {
    const char *inb = reinterpret_cast<const char *>(t.in_base + (size_t)chunk * CK * in_plane);
    glds16(reinterpret_cast<const float *>(inb + roff0), buf + (wave * 64) * 4);
    if (raw2) glds16(reinterpret_cast<const float *>(inb + roff1), buf + (NTH + (wave - R0 / 64) * 64) * 4);
  };
  const unsigned uoff = 16u * (unsigned)tid;
  auto issue_u = [&](const TileRef &t, int chunk, float *buf) {
    const char *wb = reinterpret_cast<const char *>(t.w_base + (size_t)chunk * T::U_FLOATS);
#pragma unroll
    for (int it = 0; it < NIT_U; ++it)
      if (it < NIT_U - 1 || tid < U_V4 - (NIT_U - 1) * NTH) glds16(reinterpret_cast<const float *>(wb + (uoff + 16u * NTH * it)), buf + (it * NTH + wave * 64) * 4);
  };

  // ---- input transform: two threads per patch.  u = tid & 255: input channel = u >> 6 (= wave & 3), tile = lane & 31 (tile row
  // tile >> 3, tile column tile & 7), half h = lane >> 5 (patch rows 3 h .. 3 h + 2 in the row pass, columns 3 h .. in the column pass)
  // (4-wave form: 64 patches = two waves; input channel = 2 (wave & 1) + bit 4 of the lane, tile = lane & 15)
  const int x_ci = TB == 2 ? (wave & 3) : 2 * (wave & 1) + ((lane >> 4) & 1), x_tile = TB == 2 ? (lane & 31) : (lane & 15), x_h = lane >> 5;
  const int x_trow = x_tile >> 3, x_tcol = x_tile & 7;
  const int raw_off = x_ci * (LH * LW) + (4 * x_trow + 3 * x_h) * LW + 4 * x_tcol + 3;   // LDS row 0 = output row y0 - 1, LDS column 4 = output column x0
  // V[h][q / 4][tb][lane = 16 ci + (tile & 15)][q & 3], q = 3 i + c: this thread writes (h, i = 0..5, c = 0..2)
  const int v_off = x_h * (1280 * TB) + (x_tile >> 4) * 256 + (16 * x_ci + (x_tile & 15)) * 4;   // + (q >> 2) * 256 TB + (q & 3) floats
  // Steps of one patch half: 0..2 read row r (b32, b128, b32 = columns 3, 4 .. 7, 8 of the halo row), 3..5 row pass of row r,
  // 6 the exchange, 7..9 column pass of column c + stores
  float xr[3][6];        // rows after the row pass; xr[r][0..2] stay, xr[r][3..5] are swapped with the partner's
  f32x4v xa[3];
  float xs0[3], xs1[3];
  auto xf_step = [&](const float *raw, float *vb, int st) {
    if (st < 3) {
      const float *d = raw + raw_off + st * LW;
      xs0[st] = d[0];
      xa[st] = *reinterpret_cast<const f32x4v *>(d + 1);
      xs1[st] = d[5];
    } else if (st < 6) {
      const int r = st - 3;
      wino4_in6(xs0[r], xa[r][0], xa[r][1], xa[r][2], xa[r][3], xs1[r], xr[r]);
    } else if (st == 6) {
      // h = 0 keeps columns 0-2 and needs the partner's rows 3-5 of them; h = 1 keeps columns 3-5 and needs rows 0-2.  One swap per
      // value: v0 = the value in column c, v1 = the value in column c + 3; lanes 32-63 of v0 <-> lanes 0-31 of v1.  Afterwards, in
      // BOTH halves, v0 holds rows 0-2 and v1 rows 3-5 of the thread's own three columns.
#pragma unroll
      for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          typedef unsigned u32x2s __attribute__((ext_vector_type(2)));
          const u32x2s sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(xr[r][c]), __float_as_uint(xr[r][c + 3]), false, false);
          xr[r][c] = __uint_as_float(sw[0]);
          xr[r][c + 3] = __uint_as_float(sw[1]);
        }
    } else {
      const int c = st - 7;
      float o[6];
      wino4_in6(xr[0][c], xr[1][c], xr[2][c], xr[0][c + 3], xr[1][c + 3], xr[2][c + 3], o);
#pragma unroll
      for (int i = 0; i < 6; ++i) vb[v_off + ((3 * i + c) >> 2) * (256 * TB) + ((3 * i + c) & 3)] = o[i];
    }
  };

  const int a_lane = cq * 64 + lane;   // 16-byte pieces in a filter slab: + (pos / 4) * 256
  const int b_lane = tb * 64 + lane;   // 16-byte pieces in a V buffer:     + (5 h + q / 4) * 64 TB

  // ---- tile assignment: conv_wino2.hip.h's XCD-banded counters (a.sched), or blockIdx.x + k gridDim.x
  const int band = blockIdx.x & 7;
  auto wgs_before = [&](int b) { return min(b, (int)gridDim.x & 7) + b * ((int)gridDim.x >> 3); };
  auto band_lo = [&](int b) { return (int)((long)n_tiles * wgs_before(b) / (int)gridDim.x); };
  auto band_hi = [&](int b) { return band_lo(b + 1); };
  auto band_wgs = [&](int b) { return wgs_before(b + 1) - wgs_before(b); };
  auto steal = [&]() {
    for (int k = 1; k < 8; ++k) {
      const int b = (band + k) & 7;
      if (band_lo(b) + band_wgs(b) >= band_hi(b)) continue;
      const int v = band_lo(b) + band_wgs(b) + atomicAdd(a.sched + b, 1);
      if (v < band_hi(b)) return v;
    }
    return n_tiles;
  };
  auto all_done = [&]() {
    if (a.sched && tid == 0 && atomicAdd(a.sched + 8, 1) == (int)gridDim.x - 1)
      for (int k = 0; k < 9; ++k) a.sched[k] = 0;
  };
  int *const sched_slot = reinterpret_cast<int *>(smem + (T::LDS_BYTES - 16) / 4);
  // static assignment (a.sched == nullptr: single-round launches, short K loops): the same XCD bands without the counters -- workgroup
  // blockIdx.x takes tile (workgroups of lower bands) + (its rank in its band), a permutation of 0 .. gridDim.x - 1, then + k gridDim.x.
  // With tile = blockIdx.x neighbouring tiles sat on different XCDs and every L2 fetched its own copy of the halos and of the lines
  // tiles share: conv3a / conv3b / conv4a / conv4b / convPa+Da moved 2.8 - 6 x their algorithmic bytes (profiles/r04_pmc_layers.json)
  int tile_id = SPVO_STATIC_BANDS ? wgs_before(band) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (a.sched) {
    tile_id = band_lo(band) + (blockIdx.x >> 3);
    if (tile_id >= band_hi(band)) {
      if (tid == 0) *sched_slot = steal();
      __syncthreads();
      tile_id = *sched_slot;
      __syncthreads();
    }
  }
  if (tile_id >= n_tiles) {
    all_done();
    return;
  }
  TileRef cur = decode(tile_id);

  // prefetch cursors over the item sequence: filters one item ahead, raw tiles two items ahead
  int nxt_id = tile_id + gridDim.x;
  int dyn_fetch = 0;
  struct Cursor { TileRef t; int chunk, id; };
  auto advance = [&](Cursor &q) {
    if (++q.chunk == a.n_chunks) {
      asm volatile("" ::: "memory");   // a real branch, taken once per tile
      q.chunk = 0;
      q.id = a.sched ? nxt_id : q.id + (int)gridDim.x;
      if (q.id < n_tiles) q.t = decode(q.id);
    }
  };
  Cursor cu{cur, 0, tile_id};
  issue_raw(cu.t, 0, smem + T::RAW_OFF);
  issue_u(cu.t, 0, smem + T::U_OFF);
  advance(cu);                                   // item 1
  if (cu.id < n_tiles) issue_raw(cu.t, cu.chunk, smem + T::RAW_OFF + T::IN_FLOATS);
  Cursor cr = cu;
  advance(cr);                                   // item 2
  asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
  if (wave < 2 * TB) {                           // item 0's transform has nothing to hide behind ...
#pragma unroll
    for (int st = 0; st < 10; ++st) xf_step(smem + T::RAW_OFF, smem + T::V_OFF, st);
  } else {                                       // ... nor has the first half of item 1's (the other group: its second half follows in item 0)
#pragma unroll
    for (int st = 0; st < WINO4_H1_STEPS; ++st) xf_step(smem + T::RAW_OFF + T::IN_FLOATS, smem + T::V_OFF + T::V_FLOATS, st);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // raw tile 0 is read: item 2's goes there (first half of its transform: in item 0)
  if (cr.id < n_tiles) issue_raw(cr.t, cr.chunk, smem + T::RAW_OFF);
  advance(cr);                                   // item 3

#ifdef WINO_STAMPS   // diagnostic build (tools/wino_bench.hip): shader-clock cycles per wave waiting for LDS-DMA / stores at an item's start (o[7]: of
                     // which in a tile's SECOND item, the first wait behind an epilogue's stores), at the barrier, in the matrix stream, in epilogues
  unsigned long long st_dma = 0, st_bar = 0, st_mfma = 0, st_epi = 0, st_items = 0, st_post = 0, st_bar_xf = 0, st_mfma_xf = 0;   // _xf: in the items in which this wave does the transform's first half
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
  int st_c = 0;
#endif
  int k = 0;                    // items done: selects the buffers and the transforming half
  bool drained = false;         // the LDS-DMA this item needs has been waited for already
  constexpr unsigned OOB = 0xFFFFFFFFu;

  while (tile_id < n_tiles) {
    if (a.sched) {
      if (tid == 0) dyn_fetch = band_lo(band) + band_wgs(band) + atomicAdd(a.sched + band, 1);
    } else {
      nxt_id = tile_id + gridDim.x;
    }

    // acc[pos]: position pos = 6 i + j of this wave's 16 tiles; register r = output channel 16 cq + 4 g4 + r, lane c16 = tile
    f32x4v acc[36];
    auto item = [&](auto first_tag, bool publish) {
      constexpr bool FIRST = decltype(first_tag)::value;   // the tile's first chunk: C = 0 in every accumulator's (only) instruction
      // group k & 1 (waves 0-3 / 4-7) does the first half of item k + 2's transform, the other group the second half of item k + 1's
      // (wave-uniform: a scalar branch per step)
      const bool XF = __builtin_amdgcn_readfirstlane((wave >> TB) == (k & 1) ? 1 : 0) != 0;
#ifdef WINO_STAMPS
      const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
      if (!drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      drained = false;
#ifdef WINO_STAMPS
      const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
      if (publish && tid == 0) *sched_slot = dyn_fetch < band_hi(band) ? dyn_fetch : n_tiles;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (publish) nxt_id = __builtin_amdgcn_readfirstlane(*sched_slot);
#ifdef WINO_STAMPS
      const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
      st_dma += ts1 - ts0; st_bar += ts2 - ts1;
      if (st_c == 1) st_post += ts1 - ts0;
      if (XF) st_bar_xf += ts2 - ts1;
#endif
      const float *ub = smem + T::U_OFF + (k & 1) * T::U_FLOATS;
      const float *vb = smem + T::V_OFF + (k & 1) * T::V_FLOATS;
      float *u_next = smem + T::U_OFF + ((k + 1) & 1) * T::U_FLOATS;
      const float *raw_next2 = smem + T::RAW_OFF + (k & 1) * T::IN_FLOATS;    // raw(k+2): landed before this item, read by H1 now
      float *raw_next3 = smem + T::RAW_OFF + ((k + 1) & 1) * T::IN_FLOATS;    // raw(k+3) replaces raw(k+1), whose H1 ran during item k-1
      float *v_next = smem + T::V_OFF + ((k + 1) & 1) * T::V_FLOATS;
      const f32x4v *ub4 = reinterpret_cast<const f32x4v *>(ub) + a_lane;
      const f32x4v *vb4 = reinterpret_cast<const f32x4v *>(vb) + b_lane;
      // 36 matrix instructions in the order p' = 18 h + q (q = 3 i + c; position 6 i + 3 h + c); operands: A piece p' / 4 (9 reads),
      // B piece 5 h + q / 4 (10 reads), each read LEAD instructions before its first use
      constexpr int LEAD = WINO4_LEAD;
      f32x4v av[9], bv[10];
      auto b_piece = [](int p) { return 5 * (p / 18) + ((p % 18) >> 2); };
#pragma unroll
      for (int p = 0; p < ((WINO4_ABL & 8) ? 36 : LEAD); ++p) {
        if ((WINO4_ABL & 8) && k > 0) break;
        if ((p & 3) == 0) av[p >> 2] = ub4[(p >> 2) * 256];
        if (p == 0 || b_piece(p) != b_piece(p - 1)) bv[b_piece(p)] = vb4[b_piece(p) * (64 * TB)];
      }
#pragma unroll
      for (int p = 0; p < 36; ++p) {
        const int ph = p / 18, pq = p % 18, pos = 6 * (pq / 3) + 3 * ph + pq % 3;
        const float a_op = av[p >> 2][p & 3], b_op = bv[b_piece(p)][pq & 3];
#if WINO4_PRIO
#if WINO4_PRIO == 1
        if (p == 0) asm volatile("s_setprio 3");
        if (p == 9) asm volatile("s_setprio 2");
        if (p == 18) asm volatile("s_setprio 1");
        if (p == 27) asm volatile("s_setprio 0");
#elif WINO4_PRIO == 2   // the younger wave of a SIMD (tb = 1) keeps each level longer
        if (p == 0) asm volatile("s_setprio 3");
        if (tb == 0) { if (p == 6) asm volatile("s_setprio 2"); if (p == 15) asm volatile("s_setprio 1"); if (p == 24) asm volatile("s_setprio 0"); }
        else { if (p == 12) asm volatile("s_setprio 2"); if (p == 24) asm volatile("s_setprio 1"); }
#elif WINO4_PRIO == 3   // two levels only: high in the first half of the item
        if (p == 0) asm volatile("s_setprio 1");
        if (p == 18) asm volatile("s_setprio 0");
#elif WINO4_PRIO == 4   // older wave one level below the younger one throughout, both falling
        if (tb == 0) { if (p == 0) asm volatile("s_setprio 2"); if (p == 12) asm volatile("s_setprio 1"); if (p == 24) asm volatile("s_setprio 0"); }
        else { if (p == 0) asm volatile("s_setprio 3"); if (p == 12) asm volatile("s_setprio 2"); if (p == 24) asm volatile("s_setprio 1"); }
#endif
#endif
        if (FIRST) acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op, b_op, f32x4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        else acc[pos] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op, b_op, acc[pos], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (!(WINO4_ABL & 2) && p == 1 && cu.id < n_tiles) issue_u(cu.t, cu.chunk, u_next);
        if (!(WINO4_ABL & 4) && p == 4 && cr.id < n_tiles) issue_raw(cr.t, cr.chunk, raw_next3);
        const int q = p + LEAD;
        if (!(WINO4_ABL & 8) && q < 36) {
          if ((q & 3) == 0) av[q >> 2] = ub4[(q >> 2) * 256];
          if (b_piece(q) != b_piece(q - 1)) bv[b_piece(q)] = vb4[b_piece(q) * (64 * TB)];
        }
        if (!(WINO4_ABL & 1) && XF && p >= WINO4_H1_START && p < WINO4_H1_START + WINO4_H1_STEPS * WINO4_H1_STRIDE && (p - WINO4_H1_START) % WINO4_H1_STRIDE == 0)
          xf_step(raw_next2, v_next, (p - WINO4_H1_START) / WINO4_H1_STRIDE);                          // H1 of item k + 2
        if (!(WINO4_ABL & 1) && !XF && p >= WINO4_H2_START && p < WINO4_H2_START + (10 - WINO4_H1_STEPS) * WINO4_H2_STRIDE && (p - WINO4_H2_START) % WINO4_H2_STRIDE == 0)
          xf_step(raw_next2, v_next, WINO4_H1_STEPS + (p - WINO4_H2_START) / WINO4_H2_STRIDE);        // H2 of item k + 1 (registers of this wave's H1 in the item before)
        __builtin_amdgcn_sched_barrier(0);
      }
      advance(cu);
      advance(cr);
      ++k;
#ifdef WINO_STAMPS
      { const unsigned long long dt = __builtin_amdgcn_s_memtime() - ts2; st_mfma += dt; if (XF) st_mfma_xf += dt; }
      ++st_items; ++st_c;
#endif
    };
#ifdef WINO_STAMPS
    st_c = 0;
#endif
    item(std::true_type{}, false);
    for (int c = 1; c < a.n_chunks; ++c) item(std::false_type{}, a.sched != nullptr && c == 1);

    // everything in flight for the next item has landed before this tile's stores queue up behind it
#ifdef WINO_STAMPS
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drained = true;

    // ---------------------------------------------------------------- epilogue: Y = A^T M A, bias, ReLU, (pool), store
    const int co0 = cur.ct * CO_TILE + cq * 16 + 4 * g4;   // this lane's four output channels co0 .. co0 + 3
    float *co_base = a.out + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)cur.ct * CO_TILE + cq * 16) * out_plane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
    const int oplane = (int)out_plane;
    const int kmax = a.cout - co0;                                             // channels r < kmax of this lane's 4 exist
    const float *bias_p = a.wpack + (size_t)a.co_tiles * a.n_chunks * T::U_FLOATS + min(co0, a.cout - 1);
    const int trow = 2 * tb + (c16 >> 3), tcol = c16 & 7;
    auto relu = [](float v) { return RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v; };
    const int oy = cur.y0 + 4 * trow, ox = cur.x0 + 4 * tcol;                  // first output pixel of this lane's tile
#pragma unroll
    for (int rp = 0; rp < 2; ++rp) {   // channels 2 rp, 2 rp + 1 of this lane's four: two per vector instruction
      typedef wino4_f32x2 f32x2;
      const f32x2 bias_v = {bias_p[2 * rp < kmax ? 2 * rp : 0], bias_p[2 * rp + 1 < kmax ? 2 * rp + 1 : 0]};
      auto pr = [&](int p) { return f32x2{acc[p][2 * rp], acc[p][2 * rp + 1]}; };
      f32x2 z[6][4];   // column pass: z[j][.] = A^T M[., j]
#pragma unroll
      for (int j = 0; j < 6; ++j) wino4_out4(pr(0 + j), pr(6 + j), pr(12 + j), pr(18 + j), pr(24 + j), pr(30 + j), z[j]);
      f32x2 y2[4][4];   // row pass: y[i][.] = A^T z[., i]
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        wino4_out4(z[0][i], z[1][i], z[2][i], z[3][i], z[4][i], z[5][i], y2[i]);
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) y2[i][jj] += bias_v;
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
      const int r = 2 * rp + e;
      float y[4][4];
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) y[i][jj] = relu(y2[i][jj][e]);
      const bool ch_ok = r < kmax;
      // Stores: one 16-byte piece per tile row (8 bytes per pooled row).  A tile that straddles the right edge writes ZEROS into
      // the plane's right padding (at least PADX = 4 columns, zero by construction and read as such by the next layer's halo),
      // so that odd widths and widths that are not multiples of 4 take the same wide stores.
      if constexpr (POOL) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const int ow = a.W >> 1;
#pragma unroll
        for (int pi = 0; pi < 2; ++pi) {
          float pv[2];
#pragma unroll
          for (int pj = 0; pj < 2; ++pj)
            pv[pj] = fmaxf(fmaxf(y[2 * pi][2 * pj], y[2 * pi][2 * pj + 1]), fmaxf(y[2 * pi + 1][2 * pj], y[2 * pi + 1][2 * pj + 1]));
          const int py = (oy >> 1) + pi, px = ox >> 1;
          const unsigned vo = (ch_ok && py < (a.H >> 1) && px < ow) ? 4u * (unsigned)(4 * g4 * oplane + (py + PADY) * a.out_wp + (px + PADX)) : OOB;
          const u32x2 v = {__float_as_uint(pv[0]), __float_as_uint(px + 1 < ow ? pv[1] : 0.f)};
          __builtin_amdgcn_raw_buffer_store_b64(v, rsrc, (WINO4_ABL & 16) ? OOB : vo, r * oplane * 4, 0);
        }
      } else {
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int yy = oy + i;
          const unsigned vo = (ch_ok && yy < a.H && ox < a.W) ? 4u * (unsigned)(4 * g4 * oplane + (yy + PADY) * a.out_wp + (ox + PADX)) : OOB;
          const u32x4 v = {__float_as_uint(y[i][0]), __float_as_uint(ox + 1 < a.W ? y[i][1] : 0.f), __float_as_uint(ox + 2 < a.W ? y[i][2] : 0.f),
                           __float_as_uint(ox + 3 < a.W ? y[i][3] : 0.f)};
          __builtin_amdgcn_raw_buffer_store_b128(v, rsrc, (WINO4_ABL & 16) ? OOB : vo, r * oplane * 4, 0);
        }
      }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#ifdef WINO_STAMPS
    st_epi += __builtin_amdgcn_s_memtime() - te0;
#endif
    tile_id = nxt_id;
    if (tile_id < n_tiles) cur = decode(tile_id);
  }
  all_done();   // the last workgroup out resets the counters for the next launch
#ifdef WINO_STAMPS
  if (lane == 0 && a.stamps) {
    unsigned long long *o = a.stamps + 8 * (blockIdx.x * (4 * TB) + wave);
    o[0] = st_dma; o[1] = st_bar; o[2] = st_mfma; o[3] = st_epi; o[4] = st_items;
    o[5] = __builtin_amdgcn_s_memtime() - st_t0; o[6] = __builtin_amdgcn_s_memrealtime() - st_r0; o[7] = st_post;
#ifdef WINO_STAMPS_ROLES   // o[0], o[7]: matrix stream and barrier wait (in front) of the items in which this wave did the transform's FIRST half: half of its items
    o[0] = st_mfma_xf; o[7] = st_bar_xf;
#endif
  }
#endif
}

}  // namespace spvo
