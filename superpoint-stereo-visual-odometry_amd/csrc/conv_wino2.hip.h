// conv_wino2.hip.h -- K2w: 3x3 convolution by the Winograd minimal-filtering algorithm F(2x2, 3x3) on the gfx950 fp32
// matrix cores, bias + ReLU (+ 2x2 max-pool) fused.  fp32 throughout: operands, products and accumulation.
//
// Replaces, like conv_mfma.hip.h, the TensorRT engine the reference enqueues at
// feature_detection_neural_network.cpp:169 for the 3x3 Conv/Relu/MaxPool nodes of the SuperPoint graphs; TensorRT's
// own fp32 tactics for 3x3 stride-1 layers are Winograd kernels as well.
//
// Y(2x2) = A^T [ (G g G^T) .* (B^T d B) ] A  per 4x4 input patch d (patches overlap by 2) and 3x3 filter g, with
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1],  G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1],  A^T = [1 1 1 0; 0 1 -1 -1].
// Summed over input channels the element-wise product is 16 independent GEMMs, one per position xi = (a, b) of the 4x4
// transform domain:   M[xi][co][tile] = sum_ci U[xi][co][ci] * V[xi][ci][tile]
// i.e. 16 multiplies per 2x2 outputs and channel pair instead of 36: 4/9 of the direct method's matrix work.
//
// Form: EIGHT waves, two per SIMD (round 2; the round-1 form -- 4 waves, each a 32 x 32 block of all 16 positions = 256 accumulator
// registers, one wave per SIMD -- is gone: every instruction of that one wave which was not a matrix instruction stopped the SIMD's
// matrix pipe, ~6000 cycles per item for 4096 of matrix work).  Workgroup = 64 output channels x 64 Winograd tiles (8 rows x 32
// columns of output); its work is a linear sequence of items k = (tile, chunk of 8 input channels), software-pipelined three deep:
// LDS-DMA (global_load_lds) of the host-transformed filters of item k+1 and of the raw halo tile (8 x 10 x 40 floats) of item k+2
// from inside the matrix stream of item k, the input transform of item k+1 in micro-steps between the matrix instructions, one
// barrier per item.  The kernel must not spill: a scratch reload's s_waitcnt vmcnt(0) also waits for the LDS-DMA in flight.
// Epilogue in registers: inverse transform, bias through position (1,1) (inverse-transform weight +1 for all four outputs), 2x2
// max-pool = the maximum of the tile's own four outputs.
//
// Numerics: the transforms use coefficients 0, +-1, +-1/2 only; against a float64 evaluation the layer's error is of the
// same order as the direct kernel's accumulated rounding, in fact smaller (tests/test_gpu_network.py: 1e-4 bar on every tensor
// against the oracle, and test_winograd_layers_stay_at_fp32_rounding_level against float64).
//
// Who holds the accumulators.  The round-1 form gave each of 4 waves a 32 x 32 block of all 16 transform positions: 256 accumulator
// registers, so ONE wave per SIMD, and every instruction that wave issues which is not a matrix instruction -- 13 LDS-DMA
// pieces, 40 transform micro-steps, 32 operand reads, the barrier -- is time the matrix pipe of that SIMD may stand still
// (measured: ~6000 cycles per item for 4096 cycles of matrix work).  Here the workgroup has 8 waves; wave (cq, tb) owns 16
// output channels x 32 tiles x 16 positions on v_mfma_f32_16x16x4_f32 (two 16 x 16 blocks per position): 128 accumulator
// registers, 256 registers per wave, two waves per SIMD.  Each wave issues half the matrix instructions and half of
// everything else; while one of a SIMD's two waves waits -- on an LDS-DMA issue, a transform step, the barrier -- the other
// one's matrix instructions keep the pipe busy.
//
// LDS images (the staging loads stay lane-linear 16-byte pieces):
//   U  [xi/2 8][cq 4][lane 64][xi&1][s 2]   lane = 16 (ci & 3) + (co & 15), s = ci >> 2: a lane's two k-steps of TWO positions
//                                           are one 16-byte piece (ds_read_b128, conflict-free): 8 A reads per item, not 16 --
//                                           every LDS read next to an fp32 matrix instruction costs ~8 cycles of matrix time
//   V  [xi 16][tb 2][lane 64][blk 2][s 2]   lane = 16 (ci & 3) + (tile & 15), blk = (tile >> 4) & 1: both 16-tile blocks and
//                                           both k-steps of one position are one 16-byte piece (ds_read_b128)
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <vector>
#include "conv_mfma.hip.h"

#ifndef SPVO_STATIC_BANDS
#define SPVO_STATIC_BANDS 1   // measurement builds: 0 = tile = blockIdx.x in launches without a.sched
#endif

namespace spvo {

#ifndef WINO_STORE_AUX
#define WINO_STORE_AUX 0   // cache policy of the output stores (experiments: 16 = sc1, write-through)
#endif

struct WinoTile {
  static constexpr int CK = 8, TH = 8, TW = 32, LW = TW + 8, LH = TH + 2;
  static constexpr int IN_FLOATS = CK * LH * LW;            // 3200: raw halo tile, row = x0-4 .. x0+35
  static constexpr int U_FLOATS = 16 * CK * CO_TILE;        // 8192
  static constexpr int W_FLOATS = U_FLOATS + CO_TILE;       // + the bias row (chunk 0's slab)
  static constexpr int V_FLOATS = 16 * CK * 64;             // transformed input
  // LDS: two raw tiles, two filter slabs, two transformed tiles (everything double-buffered)
  static constexpr int RAW_OFF = 0, U_OFF = 2 * IN_FLOATS, V_OFF = U_OFF + 2 * W_FLOATS;
  static constexpr int LDS_BYTES = (V_OFF + 2 * V_FLOATS) * 4;   // 157 184
};

constexpr int WINO2_LDS_BYTES = WinoTile::LDS_BYTES + 16;   // + the slot through which a tile's successor is published (a.sched)

// OIHW weights + bias -> slabs [co_tile][chunk][xi/2 8][cq][lane 64][xi&1][s 2] of U = G g G^T (double) + a bias row; cot = output
// channels per workgroup: 64 (cq 4) or, for the narrow form, 32 (cq 2)
inline std::vector<float> pack_conv_weights_wino2(const float *w, const float *bias, int cout, int cin, int cot = CO_TILE) {
  constexpr int CK = WinoTile::CK;
  const int co_tiles = (cout + cot - 1) / cot, nch = cin / CK, ncq = cot / 16;
  const int u_floats = 16 * CK * cot, w_floats = u_floats + cot;
  std::vector<float> out((size_t)co_tiles * nch * w_floats, 0.f);
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int ch = 0; ch < nch; ++ch) {
      float *slab = out.data() + ((size_t)ct * nch + ch) * w_floats;
      for (int o = 0; o < cot; ++o) {
        const int co = ct * cot + o;
        if (co >= cout) continue;
        for (int c = 0; c < CK; ++c) {
          const float *g = w + ((size_t)co * cin + ch * CK + c) * 9;
          double t[4][3];
          for (int a = 0; a < 4; ++a)
            for (int k = 0; k < 3; ++k) t[a][k] = G[a][0] * g[0 * 3 + k] + G[a][1] * g[1 * 3 + k] + G[a][2] * g[2 * 3 + k];
          const int lane = 16 * (c & 3) + (o & 15), s = c >> 2, cq = o >> 4;
          for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
              slab[((((a * 4 + b) >> 1) * ncq + cq) * 64 + lane) * 4 + 2 * ((a * 4 + b) & 1) + s] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
        }
        if (ch == 0) slab[u_floats + o] = bias[co];
      }
    }
  return out;
}

// NARROW: 32 output channels per workgroup instead of 64 -- wave (cq of 2, tb of 4) owns 16 channels x 16 tiles x 16 positions
// (64 accumulator registers, 32 matrix instructions per item).  Twice the workgroups for the same layer: for the layers at 1/8
// resolution (conv4a / conv4b at 45 x 147: 120 workgroups of the wide form on 256 CUs) the chip is filled and a tile's chain
// of items -- the layer's duration -- is made of items with half the matrix work.  The input transform is done once per
// 32 instead of once per 64 output channels, which is why the wide form stays the choice wherever it fills the chip.
template <bool POOL, bool RELU, int TAG = 0, bool ODD = false, bool NARROW = false>
__global__ __launch_bounds__(512, 2) void conv_wino2_kernel(const ConvArgs a) {
  using T = WinoTile;
  constexpr int CK = T::CK, LW = T::LW, LH = T::LH, LW4 = LW / 4;
  constexpr int COT = NARROW ? 32 : 64, NCQ = COT / 16, NBLK = NARROW ? 1 : 2;
  constexpr int U_FL = 16 * CK * COT, W_FL = U_FL + COT;                    // a filter slab in global memory
  constexpr int IN_V4 = T::IN_FLOATS / 4, W_V4 = W_FL / 4;
  constexpr int NIT_R = (IN_V4 + 511) / 512, NIT_U = (W_V4 + 511) / 512;   // 2, 5 (narrow: 3)
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int c16 = lane & 15, g4 = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cq = wave % NCQ, tb = wave / NCQ;
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
    t.w_base = a.wpack + (size_t)t.ct * a.n_chunks * W_FL;
    return t;
  };

  // staging plans: this thread's 16-byte pieces of a raw tile / of a filter slab (wave-uniform 64-bit base + 32-bit lane offset)
  unsigned roff[NIT_R];
#pragma unroll
  for (int it = 0; it < NIT_R; ++it) {
    const int idx = min(it * 512 + tid, IN_V4 - 1);
    const int ci = idx / (LH * LW4);
    const int rem = idx - ci * (LH * LW4);
    const int r = rem / LW4;
    const int q = rem - r * LW4;
    roff[it] = 4u * (unsigned)(ci * (int)in_plane + r * a.in_wp + q * 4);
  }
  auto issue_raw = [&](const TileRef &t, int chunk, float *buf) {
    const char *inb = reinterpret_cast<const char *>(t.in_base + (size_t)chunk * CK * in_plane);
#pragma unroll
    for (int it = 0; it < NIT_R; ++it)
      if (it < NIT_R - 1 || it * 512 + tid < IN_V4) glds16(reinterpret_cast<const float *>(inb + roff[it]), buf + (it * 512 + wave * 64) * 4);
  };
  const unsigned uoff = 16u * (unsigned)tid;
  auto issue_u = [&](const TileRef &t, int chunk, float *buf) {
    const char *wb = reinterpret_cast<const char *>(t.w_base + (size_t)chunk * W_FL);
#pragma unroll
    for (int it = 0; it < NIT_U; ++it)
      if (it < NIT_U - 1 || it * 512 + tid < W_V4) glds16(reinterpret_cast<const float *>(wb + (uoff + 8192u * it)), buf + (it * 512 + wave * 64) * 4);
  };

  auto issue_raw_piece = [&](const TileRef &t, int chunk, float *buf, int it) {
    const char *inb = reinterpret_cast<const char *>(t.in_base + (size_t)chunk * CK * in_plane);
    if (it < NIT_R - 1 || it * 512 + tid < IN_V4) glds16(reinterpret_cast<const float *>(inb + roff[it]), buf + (it * 512 + wave * 64) * 4);
  };
  auto issue_u_piece = [&](const TileRef &t, int chunk, float *buf, int it) {
    const char *wb = reinterpret_cast<const char *>(t.w_base + (size_t)chunk * W_FL);
    if (it < NIT_U - 1 || it * 512 + tid < W_V4) glds16(reinterpret_cast<const float *>(wb + (uoff + 8192u * it)), buf + (it * 512 + wave * 64) * 4);
  };

  // input transform: ONE patch per thread -- tile t_tile of input channel `wave` of the chunk
  const int t_tile = tid & 63;
  const int t_trow = t_tile >> 4, t_tcol = t_tile & 15;
  const int raw_off = wave * (LH * LW) + (2 * t_trow) * LW + 3 + 2 * t_tcol;   // LDS row 0 = output row y0 - 1, LDS column 4 = output column x0
  const int v_off = NARROW ? (((t_tile >> 4) * 64 + 16 * (wave & 3) + t_tcol) * 2) + (wave >> 2)                              // [xi][tb 4][lane][s]
                           : (((t_tile >> 5) * 64 + 16 * (wave & 3) + t_tcol) * 4) + 2 * ((t_tile >> 4) & 1) + (wave >> 2);   // + xi * 512
  auto transform = [&](const float *raw, float *vb) {
    const float *d = raw + raw_off;
    float t[4][4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float d0 = d[0 * LW + q], d1 = d[1 * LW + q], d2 = d[2 * LW + q], d3 = d[3 * LW + q];
      t[0][q] = d0 - d2; t[1][q] = d1 + d2; t[2][q] = d2 - d1; t[3][q] = d1 - d3;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float *v = vb + v_off + (r * 4) * 512;
      v[0 * 512] = t[r][0] - t[r][2];
      v[1 * 512] = t[r][1] + t[r][2];
      v[2 * 512] = t[r][2] - t[r][1];
      v[3 * 512] = t[r][1] - t[r][3];
    }
  };
  // the same in 20 micro-steps for the matrix stream: 0..11 read patch row u / 3 as three aligned 8-byte pairs (columns c0-1 ..
  // c0+4, of which c0 .. c0+3 are the patch); 12..15 column q of B^T d; 16..19 row r of (B^T d) B, stored
  f32x2 dp[12];
  float tt[16];
  auto xf_step = [&](const float *raw, float *vb, int st) {
    if (st < 12) {
      dp[st] = *reinterpret_cast<const f32x2 *>(raw + raw_off - 1 + (st / 3) * LW + 2 * (st % 3));
    } else if (st < 16) {
      const int q = st - 12;
      auto d = [&](int r) { return q == 0 ? dp[3 * r][1] : q == 1 ? dp[3 * r + 1][0] : q == 2 ? dp[3 * r + 1][1] : dp[3 * r + 2][0]; };
      tt[0 * 4 + q] = d(0) - d(2);
      tt[1 * 4 + q] = d(1) + d(2);
      tt[2 * 4 + q] = d(2) - d(1);
      tt[3 * 4 + q] = d(1) - d(3);
    } else {
      const int r = st - 16;
      float *v = vb + v_off + (r * 4) * 512;
      v[0 * 512] = tt[r * 4 + 0] - tt[r * 4 + 2];
      v[1 * 512] = tt[r * 4 + 1] + tt[r * 4 + 2];
      v[2 * 512] = tt[r * 4 + 2] - tt[r * 4 + 1];
      v[3 * 512] = tt[r * 4 + 1] - tt[r * 4 + 3];
    }
  };
  const int a_lane = cq * 64 + lane;   // 16-byte pieces in a filter slab: + (xi / 2) * 64 NCQ
  const int b_lane = tb * 64 + lane;   // 16-byte (narrow: 8-byte) pieces in a V buffer:  + xi * 128 (256)

  // a.sched (dynamic tile assignment, see below): the tiles are cut into 8 contiguous bands, one per group of workgroups that
  // share an XCD (blockIdx.x % 8: MI355X_MICROARCH.md, workgroup dispatch) -- an XCD's ~30 workgroups then work on ~30
  // consecutive tiles = most of one row of tiles at a time, and the halo columns / rows and the 128-byte lines that neighbouring
  // tiles share are fetched into that XCD's L2 once instead of once per XCD.  Bands are sized in proportion to their workgroups
  // (238 workgroups are 6 groups of 30 and 2 of 29), so all of them run dry together and nobody needs to look into another band
  // (tried: the futile atomics of 238 workgroups on 8 counters at the end cost more than the balance they bought).  A band's first
  // tiles go to its workgroups by rank (no start-up atomic), the rest through the band's counter a.sched[band].  a.sched[8]
  // counts the workgroups that are done; the last one clears all nine.
  const int band = blockIdx.x & 7;
  auto wgs_before = [&](int b) { return min(b, (int)gridDim.x & 7) + b * ((int)gridDim.x >> 3); };   // workgroups with blockIdx.x % 8 < b
  auto band_lo = [&](int b) { return (int)((long)n_tiles * wgs_before(b) / (int)gridDim.x); };
  auto band_hi = [&](int b) { return band_lo(b + 1); };
  auto band_wgs = [&](int b) { return wgs_before(b + 1) - wgs_before(b); };
  auto steal = [&]() {   // thread 0, blocking: a workgroup without a first tile of its own (more workgroups than tiles in its band)
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
  int *const sched_slot = reinterpret_cast<int *>(smem + T::LDS_BYTES / 4);   // 16 bytes behind the V buffers (WINO2_LDS_BYTES)
  // static assignment (a.sched == nullptr: single-round launches, short K loops): the same XCD bands without the counters -- workgroup
  // blockIdx.x takes tile (workgroups of lower bands) + (its rank in its band), a permutation of 0 .. gridDim.x - 1, then + k gridDim.x.
  // With tile = blockIdx.x neighbouring tiles sat on different XCDs and every L2 fetched its own copy of the halos and of the lines
  // tiles share: conv3a / conv3b / conv4a / conv4b / convPa+Da moved 2.8 - 6 x their algorithmic bytes (profiles/r04_pmc_layers.json)
  int tile_id = SPVO_STATIC_BANDS ? wgs_before(band) + ((int)blockIdx.x >> 3) : (int)blockIdx.x;
  if (a.sched) {
    tile_id = band_lo(band) + (blockIdx.x >> 3);
    if (tile_id >= band_hi(band)) {   // (uniform) more workgroups than tiles in this band: look elsewhere before anything starts
      if (tid == 0) *sched_slot = steal();
      __syncthreads();
      tile_id = *sched_slot;
      __syncthreads();
    }
  }
  if (tile_id >= n_tiles) {   // (the product's grids never exceed the tile count; a workgroup without work still counts as done)
    all_done();
    return;
  }
  TileRef cur = decode(tile_id);

  // prefetch cursors over the item sequence: filters one item ahead, raw tiles two items ahead
  // Which tile comes after the one being computed: blockIdx.x + rounds x gridDim.x (static), or -- a.sched -- whatever the
  // launch's tile counter hands out next.  A workgroup of this kernel owns its CU; when a small kernel of another stream sits
  // on that CU as the layer starts, the workgroup starts late, and with equal shares the whole layer ends late.  With the
  // counter the late workgroup simply takes fewer tiles.  The counter is read (one atomic by thread 0) when a tile begins and
  // published through LDS at the tile's second item; the prefetch cursors cross into the next tile at the end of item
  // n_chunks - 3 at the earliest, so the host selects this mode for n_chunks >= 4 only.
  int nxt_id = tile_id + gridDim.x;
  int dyn_fetch = 0;
  struct Cursor { TileRef t; int chunk, id; };
  auto advance = [&](Cursor &q) {
    if (++q.chunk == a.n_chunks) {
      asm volatile("" ::: "memory");   // keeps this a real branch, taken once per tile: the tile decode is three integer divisions on the scalar unit
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
  transform(smem + T::RAW_OFF, smem + T::V_OFF);   // item 0's transform has nothing to hide behind

#ifdef WINO_STAMPS   // diagnostic build: shader-clock cycles per wave spent waiting for LDS-DMA, at the barrier, in the matrix stream, in epilogues
  unsigned long long st_dma = 0, st_bar = 0, st_mfma = 0, st_epi = 0, st_items = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif
  int k = 0;                    // items done: selects the buffers
  bool drained = true;          // the LDS-DMA this item needs has been waited for already
  constexpr unsigned OOB = 0xFFFFFFFFu;

  while (tile_id < n_tiles) {
    if (a.sched) {
      if (tid == 0) dyn_fetch = band_lo(band) + band_wgs(band) + atomicAdd(a.sched + band, 1);
    } else {
      nxt_id = tile_id + gridDim.x;
    }

    // acc[xi][blk]: position xi, tiles 16 blk .. 16 blk + 15 of this wave's 32; register r of a block = output channel 4 g4 + r
    f32x4v acc[16][NBLK];
    auto item = [&](auto first_tag, bool publish) {
      constexpr bool FIRST = decltype(first_tag)::value;   // the tile's first chunk: C = 0 in every accumulator's first instruction
#ifdef WINO_STAMPS
      const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
      if (!drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      drained = false;
#ifdef WINO_STAMPS
      const unsigned long long ts1 = __builtin_amdgcn_s_memtime();
#endif
      if (publish && tid == 0) *sched_slot = dyn_fetch < band_hi(band) ? dyn_fetch : n_tiles;   // (the wait above covered the atomic's return)
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      if (publish) nxt_id = __builtin_amdgcn_readfirstlane(*sched_slot);
#ifdef WINO_STAMPS
      const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
      st_dma += ts1 - ts0; st_bar += ts2 - ts1;
#endif
      const float *ub = smem + T::U_OFF + (k & 1) * T::W_FLOATS;
      const float *vb = smem + T::V_OFF + (k & 1) * T::V_FLOATS;
      float *u_next = smem + T::U_OFF + ((k + 1) & 1) * T::W_FLOATS;
      float *raw_next2 = smem + T::RAW_OFF + (k & 1) * T::IN_FLOATS;          // raw(k+2) replaces raw(k), transformed during item k-1
      const float *raw_next = smem + T::RAW_OFF + ((k + 1) & 1) * T::IN_FLOATS;
      float *v_next = smem + T::V_OFF + ((k + 1) & 1) * T::V_FLOATS;
      const f32x4v *vb4 = reinterpret_cast<const f32x4v *>(vb);
      {
        // 64 matrix instructions in 4 groups of 4 positions; per position one ds_read_b64 (A: two k-steps) and one ds_read_b128
        // (B: two blocks x two k-steps), read while the previous group multiplies.  Inside a group: k-step outer, position, block
        // inner -- an accumulator is revisited after 7 other instructions.
        f32x4v avq[2];   // [pair]: positions 4 g + 2 p, 4 g + 2 p + 1, two k-steps each
        f32x4v bv[4];
        const f32x4v *ub4 = reinterpret_cast<const f32x4v *>(ub);
        auto lda = [&](int g, int p) { avq[p] = ub4[a_lane + (2 * g + p) * (64 * NCQ)]; };
        auto ld = [&](int g, int x) {
          if constexpr (NARROW) {
            const f32x2 b2 = reinterpret_cast<const f32x2 *>(vb)[b_lane + (4 * g + x) * 256];
            bv[x][0] = b2[0]; bv[x][1] = b2[1];
          } else {
            bv[x] = vb4[b_lane + (4 * g + x) * 128];
          }
        };
        lda(0, 0); lda(0, 1);
#pragma unroll
        for (int x = 0; x < 4; ++x) ld(0, x);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          constexpr int QN = NARROW ? 8 : 16;   // matrix instructions per group of 4 positions
#pragma unroll
          for (int q = 0; q < QN; ++q) {
            const int s = NARROW ? q >> 2 : q >> 3, x = NARROW ? q & 3 : (q >> 1) & 3, blk = NARROW ? 0 : q & 1;
            const float a_op = avq[x >> 1][2 * (x & 1) + s];
            if (FIRST && s == 0) acc[4 * g + x][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op, bv[x][2 * blk + s], f32x4v{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
            else acc[4 * g + x][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(a_op, bv[x][2 * blk + s], acc[4 * g + x][blk], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#ifndef WINO2_ABL
#define WINO2_ABL 0   // timing experiments only (results are wrong when != 0): 1 no transform steps, 2 no LDS-DMA, 4 no operand reads in the loop
#endif
#ifndef WINO2_SPREAD
#define WINO2_SPREAD 0   // 1: one LDS-DMA piece per matrix instruction instead of two bursts
#endif
            const int slot = QN * g + q;
            if (!(WINO2_ABL & 2)) {
              if (WINO2_SPREAD) {
                if (slot >= 1 && slot < 1 + 2 * NIT_U && (slot & 1) && cu.id < n_tiles) issue_u_piece(cu.t, cu.chunk, u_next, (slot - 1) >> 1);
                if (slot >= 1 + 2 * NIT_U && slot < 1 + 2 * NIT_U + 2 * NIT_R && (slot & 1) && cr.id < n_tiles) issue_raw_piece(cr.t, cr.chunk, raw_next2, (slot - 1 - 2 * NIT_U) >> 1);
              } else {
                if (slot == 1 && cu.id < n_tiles) issue_u(cu.t, cu.chunk, u_next);
                if (slot == 5 && cr.id < n_tiles) issue_raw(cr.t, cr.chunk, raw_next2);
              }
            }
            if (!(WINO2_ABL & 4) && g < 3 && q >= QN - 4) {   // position x's registers were last read by instruction 9 + 2 x (narrow: 4 + x)
              ld(g + 1, q - (QN - 4));
              if (q == QN - 4 + (NARROW ? 1 : 0)) lda(g + 1, 0);   // positions 0, 1 are through after instruction 11 (narrow: 5)
              if (q == QN - 1) lda(g + 1, 1);                      // positions 2, 3 after this one
            }
            // input transform of item k+1: 20 micro-steps on every other instruction slot (staggering them between the two waves of
            // a SIMD, or packing them densely, measured slower)
            if constexpr (NARROW) {
              if (!(WINO2_ABL & 1) && slot >= 8 && slot < 28) xf_step(raw_next, v_next, slot - 8);
            } else {
              if (!(WINO2_ABL & 1) && slot >= 16 && slot < 56 && !(slot & 1)) xf_step(raw_next, v_next, (slot - 16) >> 1);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (FIRST) {   // bias through position (1,1) (inverse-transform weight +1 for all four outputs): A = (bias, 0, 0, 0), B = 1
        const float bias_a = g4 ? 0.f : ub[U_FL + cq * 16 + c16];
#pragma unroll
        for (int blk = 0; blk < NBLK; ++blk) acc[5][blk] = __builtin_amdgcn_mfma_f32_16x16x4f32(bias_a, 1.0f, acc[5][blk], 0, 0, 0);
      }
      advance(cu);
      advance(cr);
      ++k;
#ifdef WINO_STAMPS
      st_mfma += __builtin_amdgcn_s_memtime() - ts2;
      ++st_items;
#endif
    };
    item(std::true_type{}, false);
    for (int c = 1; c < a.n_chunks; ++c) item(std::false_type{}, a.sched != nullptr && c == 1);

    // everything in flight for the next item has landed before this tile's stores queue up behind it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drained = true;

#ifdef WINO_STAMPS
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
    // ---------------------------------------------------------------- epilogue: Y = A^T M A, ReLU, (pool), store
    float *co_base = a.out + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)cur.ct * COT + cq * 16) * out_plane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
    const int oplane = (int)out_plane;
    const int kmax = a.cout - (cur.ct * COT + cq * 16 + 4 * g4);   // channels r < kmax of this lane's 4 exist
    auto relu = [](float v) { return RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v; };
#pragma unroll
    for (int blk = 0; blk < NBLK; ++blk) {
      const int trow = NARROW ? tb : 2 * tb + blk, tcol = c16;
      unsigned voff, voff10 = OOB;
      bool col1 = true;   // ODD: the tile's second column is inside the image
      if constexpr (POOL) {
        const int y = (cur.y0 >> 1) + trow, x = (cur.x0 >> 1) + tcol;
        voff = (y < (a.H >> 1) && x < (a.W >> 1)) ? 4u * (unsigned)(4 * g4 * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
      } else {
        const int y = cur.y0 + 2 * trow, x = cur.x0 + 2 * tcol;
        voff = (y < a.H && x < a.W) ? 4u * (unsigned)(4 * g4 * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
        if constexpr (ODD) {
          col1 = x + 1 < a.W;
          voff10 = (voff != OOB && y + 1 < a.H) ? voff + 4u * (unsigned)a.out_wp : OOB;
        }
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        // column pass per row a of M (positions 4 a + b), then the row pass -- single adds in the association the tests pin
        // ((m0 + m1) + m2, (m1 - m2) - m3): the two-wide form needed two register moves per packed add to pair its operands,
        // and every vector instruction here is paid in matrix time
        float sa[4], sb[4];
#pragma unroll
        for (int ar = 0; ar < 4; ++ar) {
          const float m0 = acc[4 * ar + 0][blk][r], m1 = acc[4 * ar + 1][blk][r], m2 = acc[4 * ar + 2][blk][r], m3 = acc[4 * ar + 3][blk][r];
          sa[ar] = (m0 + m1) + m2;
          sb[ar] = (m1 - m2) - m3;
        }
        const float y00 = relu((sa[0] + sa[1]) + sa[2]), y01 = relu((sb[0] + sb[1]) + sb[2]);
        const float y10 = relu((sa[1] - sa[2]) - sa[3]), y11 = relu((sb[1] - sb[2]) - sb[3]);
        const unsigned vo = r < kmax ? voff : OOB;
        if constexpr (POOL) {
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11))), rsrc, vo, r * oplane * 4, 0);
        } else if constexpr (!ODD) {
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 r0 = {__float_as_uint(y00), __float_as_uint(y01)}, r1 = {__float_as_uint(y10), __float_as_uint(y11)};
          __builtin_amdgcn_raw_buffer_store_b64(r0, rsrc, vo, r * oplane * 4, 0);
          __builtin_amdgcn_raw_buffer_store_b64(r1, rsrc, vo == OOB ? OOB : vo + 4u * (unsigned)a.out_wp, r * oplane * 4, 0);
        } else {
          // odd H or W: a second row outside the image is not stored; a second column outside the image is stored as ZERO -- it is
          // the first column of the plane's zero padding (the next layer's halo), so the row still leaves as one 8-byte piece
          typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
          const u32x2 r0 = {__float_as_uint(y00), __float_as_uint(col1 ? y01 : 0.f)}, r1 = {__float_as_uint(y10), __float_as_uint(col1 ? y11 : 0.f)};
          __builtin_amdgcn_raw_buffer_store_b64(r0, rsrc, r < kmax ? voff : OOB, r * oplane * 4, 0);
          __builtin_amdgcn_raw_buffer_store_b64(r1, rsrc, r < kmax ? voff10 : OOB, r * oplane * 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
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
    unsigned long long *o = a.stamps + 8 * (blockIdx.x * 8 + wave);
    o[0] = st_dma; o[1] = st_bar; o[2] = st_mfma; o[3] = st_epi; o[4] = st_items;
    o[5] = __builtin_amdgcn_s_memtime() - st_t0; o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
}

}  // namespace spvo
