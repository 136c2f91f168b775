// conv_wino64.hip.h -- K2u: the Winograd F(2x2, 3x3) convolution of conv_wino.hip.h for layers with 64 INPUT channels per
// (round 4: moved out of the library into tools/ -- measured no faster than the 8-wave form; kept for tools/wino_bench.hip -DWINO64)
// output-channel tile (VGG SuperPoint: conv1b, conv2a, conv2b 64 -> 64, conv3a 64 -> 128 = 540 of the 880 us of the conv
// stack), with the transformed filters RESIDENT IN REGISTERS.
//
// Same algorithm, same numerics (fp32 operands, products and accumulation; transforms with the coefficients 0, +-1, +-1/2),
// same reference (the TensorRT engine enqueued at feature_detection_neural_network.cpp:169).  What changes is where the
// filters live.  The other two forms stream the transformed filters U of every (tile, 8-channel chunk) item from L2 into LDS:
// 32 KB per item, 32 of the 45 LDS-DMA pieces a workgroup issues per item, and -- measured -- 14 % of the kernel's time in
// DMA issue alone plus the LDS reads of the A operands.  With cin = 64 the filters of one 64-channel output tile are
// 16 positions x 64 x 64 floats = 256 KB: more than LDS, exactly half of a CU's register file.  So:
//
//   * workgroup = 4 waves (one per SIMD, 512 registers each); wave r owns ROW r of the 4 x 4 transform domain -- positions
//     (r, 0..3) -- for all 64 output channels and the workgroup's 32 Winograd tiles (4 rows x 32 columns of output);
//   * its U[(r, c)][co 64][ci 64] = 256 registers per lane are loaded ONCE per workgroup (persistent kernel) and feed the A
//     operand of v_mfma_f32_32x32x2_f32 directly: no filter traffic, no A reads in the loop at all;
//   * accumulators M[(r, c)][co 64][tile 32] = 128 registers; B operands (transformed input V) from LDS, one ds_read_b128 per
//     four matrix instructions;
//   * items = (tile, 16-channel chunk): 64 matrix instructions per wave and item; what is left of the staging is the raw halo
//     tile -- 15 KB, 4 LDS-DMA pieces per wave and item (13 before) -- and the input transform of the next item (two patches per
//     thread) in the slots between the matrix instructions; one barrier per item;
//   * epilogue: Y = A^T M A.  The row pass (over c) is local to a wave; the column pass (over r) crosses waves: each wave
//     leaves its two partial rows Z[r][j] in LDS (64 KB), and after one barrier wave w finishes a quarter of the tile
//     (32 output channels x 16 register indices / 2): bias, ReLU, (2x2 max-pool = the tile's own four outputs), stores.
//
// LDS: 2 raw tiles (30 KB) + 2 transformed tiles (64 KB) + Z (64 KB) = 158 KB: one workgroup per CU, as before.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <vector>
#include "../superpoint-stereo-visual-odometry_amd/csrc/conv_mfma.hip.h"
#include "../superpoint-stereo-visual-odometry_amd/csrc/conv_wino2.hip.h"   // WinoTile

namespace spvo {

struct Wino64Tile {
  static constexpr int CIN = 64, CK = 16, NCH = CIN / CK, TH = 4, TW = 32, LW = TW + 8, LH = TH + 2;
  static constexpr int IN_FLOATS = CK * LH * LW;          // 3840: raw halo tile of one chunk, row = x0-4 .. x0+35
  static constexpr int V_FLOATS = 16 * CK * 32;           // 8192: transformed input of one chunk [xi 16][h 2][lane 64][jj 4]
  static constexpr int Z_FLOATS = 4 * 2 * 2 * 4 * 64 * 4; // 16384: [r 4][j 2][cb 2][quad 4][lane 64][4]
  static constexpr int RAW_OFF = 0, V_OFF = 2 * IN_FLOATS, Z_OFF = V_OFF + 2 * V_FLOATS;
  static constexpr int LDS_BYTES = (Z_OFF + Z_FLOATS) * 4 + 16;   // 161 808 (+ the slot through which a tile's successor is published)
  static constexpr int U_FLOATS = 16 * 64 * 64;           // transformed filters of one 64-channel output tile
};

// OIHW weights + bias -> [co_tile][r 4][c 4][cb 2][s4 8][lane 64][4] of U = G g G^T (computed in double), then [co_tiles * 64]
// biases.  Lane l of wave r holds, for position (r, c), channel block cb and k-step s = 4 s4 + e (input channels 2 s, 2 s + 1),
// the A operand of v_mfma_f32_32x32x2_f32: output channel 32 cb + (l & 31), input channel 2 s + (l >> 5).
inline std::vector<float> pack_conv_weights_wino64(const float *w, const float *bias, int cout, int cin) {
  const int co_tiles = (cout + CO_TILE - 1) / CO_TILE;
  std::vector<float> out((size_t)co_tiles * Wino64Tile::U_FLOATS + (size_t)co_tiles * CO_TILE, 0.f);
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int o = 0; o < CO_TILE; ++o) {
      const int co = ct * CO_TILE + o;
      if (co >= cout) continue;
      out[(size_t)co_tiles * Wino64Tile::U_FLOATS + co] = bias[co];
      for (int ci = 0; ci < cin; ++ci) {
        const float *g = w + ((size_t)co * cin + ci) * 9;
        double t[4][3];
        for (int a = 0; a < 4; ++a)
          for (int k = 0; k < 3; ++k) t[a][k] = G[a][0] * g[0 * 3 + k] + G[a][1] * g[1 * 3 + k] + G[a][2] * g[2 * 3 + k];
        const int cb = o >> 5, lane = 32 * (ci & 1) + (o & 31), s = ci >> 1;
        for (int a = 0; a < 4; ++a)
          for (int b = 0; b < 4; ++b)
            out[((((((size_t)ct * 4 + a) * 4 + b) * 2 + cb) * 8 + (s >> 2)) * 64 + lane) * 4 + (s & 3)] =
                (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
      }
    }
  return out;
}

#ifndef WINO64_ABL
#define WINO64_ABL 0   // timing experiments only (results are wrong when != 0): 1 no transform steps, 2 no LDS-DMA in the loop, 4 no B reads in the loop, 8 no epilogue, 16 no barrier per item
#endif

#ifndef WINO64_FLAGSYNC
#define WINO64_FLAGSYNC 1   // 1: the hand-over between items is an arrival counter in LDS (arrive early, wait late); 0: s_barrier at the item boundary
#endif

template <bool POOL, bool RELU, int TAG = 0>
__global__ __launch_bounds__(256, 1) void conv_wino64_kernel(const ConvArgs a) {
  using T = Wino64Tile;
  constexpr int CK = T::CK, LW = T::LW, LH = T::LH, LW4 = LW / 4;
  constexpr int IN_V4 = T::IN_FLOATS / 4;            // 960 16-byte pieces per raw tile
  constexpr int NIT_R = (IN_V4 + 255) / 256;         // 4 LDS-DMA instructions per thread and item
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  typedef float f32x4v __attribute__((ext_vector_type(4)));

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // = row r of the transform domain
  const size_t in_plane = (size_t)a.in_hp * a.in_wp;
  const size_t out_plane = (size_t)a.out_hp * a.out_wp;
  const int n_tiles = a.tiles_x * a.tiles_y * a.co_tiles * a.batch;

  struct TileRef { const float *in_base; int x0, y0, ct, img; };
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
    return t;
  };

  // staging plan: this thread's 16-byte pieces of a raw tile (wave-uniform 64-bit base + 32-bit lane offset)
  unsigned roff[NIT_R];
#pragma unroll
  for (int it = 0; it < NIT_R; ++it) {
    const int idx = min(it * 256 + tid, IN_V4 - 1);
    const int ci = idx / (LH * LW4);
    const int rem = idx - ci * (LH * LW4);
    const int r = rem / LW4;
    const int q = rem - r * LW4;
    roff[it] = 4u * (unsigned)(ci * (int)in_plane + r * a.in_wp + q * 4);
  }
  auto issue_raw_piece = [&](const TileRef &t, int chunk, float *buf, int it) {
    const char *inb = reinterpret_cast<const char *>(t.in_base + (size_t)chunk * CK * in_plane);
    if (it < NIT_R - 1 || it * 256 + tid < IN_V4) glds16(reinterpret_cast<const float *>(inb + roff[it]), buf + (it * 256 + wave * 64) * 4);
  };
  auto issue_raw = [&](const TileRef &t, int chunk, float *buf) {
#pragma unroll
    for (int it = 0; it < NIT_R; ++it) issue_raw_piece(t, chunk, buf, it);
  };

  // Input transform: two patches per thread and item -- tile (trow, tcol), input channels ci = 4 wave + 2 cib1 + i, i = 0, 1.
  // Lane bits: tcol = tid & 15, cib1 = bit 4, trow = bit 5.  The two 16-lane halves of a 32-lane group then read raw rows that
  // lie 2 channels = 480 floats = 32 banks (mod 64) apart: the 8-byte reads of a group cover all 64 banks once.
  const int t_tcol = tid & 15, t_cib1 = (tid >> 4) & 1, t_trow = (tid >> 5) & 1;
  const int raw_off0 = (4 * wave + 2 * t_cib1) * (LH * LW) + (2 * t_trow) * LW + 3 + 2 * t_tcol;   // patch i: + i * LH * LW
  // V[xi][h][lane = 32 (ci & 1) + tile][jj], k-step s = ci >> 1 = 4 h + jj = 2 wave + cib1:  + xi * 512 floats, patch i: + i * 128
  const int v_off0 = (wave >> 1) * 256 + (16 * t_trow + t_tcol) * 4 + 2 * (wave & 1) + t_cib1;
  auto transform = [&](const float *raw, float *vb) {   // item 0 of a workgroup: nothing to hide behind
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float *d = raw + raw_off0 + i * (LH * LW);
      float t[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float d0 = d[0 * LW + q], d1 = d[1 * LW + q], d2 = d[2 * LW + q], d3 = d[3 * LW + q];
        t[0][q] = d0 - d2; t[1][q] = d1 + d2; t[2][q] = d2 - d1; t[3][q] = d1 - d3;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float *v = vb + v_off0 + i * 128 + (r * 4) * 512;
        v[0 * 512] = t[r][0] - t[r][2];
        v[1 * 512] = t[r][1] + t[r][2];
        v[2 * 512] = t[r][2] - t[r][1];
        v[3 * 512] = t[r][1] - t[r][3];
      }
    }
  };
  // the same in 40 micro-steps (20 per patch) for the matrix stream: 0..11 read patch row u / 3 as three aligned 8-byte pairs
  // (columns c0-1 .. c0+4, of which c0 .. c0+3 are the patch); 12..15 column q of B^T d; 16..19 row r of (B^T d) B, stored
  f32x2 dp[12];
  float tt[16];
  auto xf_step = [&](const float *raw, float *vb, int st) {
    const int i = st / 20, u = st % 20;
    if (u < 12) {
      dp[u] = *reinterpret_cast<const f32x2 *>(raw + raw_off0 + i * (LH * LW) - 1 + (u / 3) * LW + 2 * (u % 3));
    } else if (u < 16) {
      const int q = u - 12;
      auto d = [&](int r) { return q == 0 ? dp[3 * r][1] : q == 1 ? dp[3 * r + 1][0] : q == 2 ? dp[3 * r + 1][1] : dp[3 * r + 2][0]; };
      tt[0 * 4 + q] = d(0) - d(2);
      tt[1 * 4 + q] = d(1) + d(2);
      tt[2 * 4 + q] = d(2) - d(1);
      tt[3 * 4 + q] = d(1) - d(3);
    } else {
      const int r = u - 16;
      float *v = vb + v_off0 + i * 128 + (r * 4) * 512;
      v[0 * 512] = tt[r * 4 + 0] - tt[r * 4 + 2];
      v[1 * 512] = tt[r * 4 + 1] + tt[r * 4 + 2];
      v[2 * 512] = tt[r * 4 + 2] - tt[r * 4 + 1];
      v[3 * 512] = tt[r * 4 + 1] - tt[r * 4 + 3];
    }
  };

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
  // Hand-over between items.  Item k + 1 needs from EVERY wave: its part of V(k+1) (written in slots 10..49 of item k), its LDS-DMA
  // pieces of raw(k+2)'s predecessor landed, and its reads of the buffers item k + 1 overwrites done.  A wave is through with all
  // of that at slot 50 of item k; it needs the others' only at the start of item k + 1, 14 matrix instructions later.  An s_barrier
  // at the boundary makes the four matrix pipes wait for the slowest wave at every item (measured: 13 % of the kernel).  Instead a
  // wave ARRIVES at slot 50 -- one LDS atomic after its own waits -- and at the boundary only checks that the count has reached
  // 4 x (items so far): the slack absorbs the skew.
  int *const sync_ctr = sched_slot + 1;
  int sync_gen = 0;   // arrivals per wave so far
  auto sync_arrive = [&]() {
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_fetch_add(sync_ctr, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    ++sync_gen;
  };
  auto sync_wait = [&]() {
    const int target = 4 * sync_gen;
    while (__builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int *>(sync_ctr)) < target) __builtin_amdgcn_s_sleep(1);
  };
  if (tid == 0) *sync_ctr = 0;
  int tile_id = blockIdx.x;
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

  // ---- the wave's filters: 256 registers, loaded when the output-channel tile changes (once per workgroup unless cout > 64)
  float ureg[4][2][32];   // [c][cb][s]
  float bias_v[8];        // the 8 output channels this lane finishes: 32 cbf + e + 8 (2 qh + qq) + 4 half
  int ct_loaded = -1;
  const int fin_cb = wave >> 1, fin_qh = wave & 1;
  auto load_filters = [&](int ct) {
    const f32x4v *up = reinterpret_cast<const f32x4v *>(a.wpack + ((size_t)ct * 4 + wave) * (T::U_FLOATS / 4)) + lane;
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int s4 = 0; s4 < 8; ++s4) {
          const f32x4v v = up[((c * 2 + cb) * 8 + s4) * 64];
          ureg[c][cb][4 * s4 + 0] = v[0]; ureg[c][cb][4 * s4 + 1] = v[1]; ureg[c][cb][4 * s4 + 2] = v[2]; ureg[c][cb][4 * s4 + 3] = v[3];
        }
    const float *bp = a.wpack + (size_t)a.co_tiles * T::U_FLOATS + ct * CO_TILE + 32 * fin_cb + 4 * half;
#pragma unroll
    for (int qq = 0; qq < 2; ++qq)
#pragma unroll
      for (int e = 0; e < 4; ++e) bias_v[4 * qq + e] = bp[e + 8 * (2 * fin_qh + qq)];
    ct_loaded = ct;
  };
  load_filters(cur.ct);

  // prefetch cursor over the item sequence: raw tiles two items ahead
  int nxt_id = tile_id + gridDim.x;
  int dyn_fetch = 0;
  struct Cursor { TileRef t; int chunk, id; };
  auto advance = [&](Cursor &q) {
    if (++q.chunk == T::NCH) {
      q.chunk = 0;
      q.id = a.sched ? nxt_id : q.id + (int)gridDim.x;
      if (q.id < n_tiles) q.t = decode(q.id);
    }
  };
  Cursor cr{cur, 0, tile_id};
  issue_raw(cr.t, 0, smem + T::RAW_OFF);
  advance(cr);                                   // item 1 (same tile: NCH = 4)
  issue_raw(cr.t, cr.chunk, smem + T::RAW_OFF + T::IN_FLOATS);
  advance(cr);                                   // item 2
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  transform(smem + T::RAW_OFF, smem + T::V_OFF);
  if (WINO64_FLAGSYNC) sync_arrive();

  bool drained = true;          // the LDS-DMA this item needs has been waited for already
  constexpr unsigned OOB = 0xFFFFFFFFu;
#ifdef WINO_STAMPS   // diagnostic build: shader-clock cycles per wave spent at the hand-over, in the matrix stream, in epilogues
  unsigned long long st_dma = 0, st_bar = 0, st_mfma = 0, st_epi = 0, st_items = 0;
  const unsigned long long st_t0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime();
#endif

  while (tile_id < n_tiles) {
    if (a.sched) {
      if (tid == 0) dyn_fetch = band_lo(band) + band_wgs(band) + atomicAdd(a.sched + band, 1);
    } else {
      nxt_id = tile_id + gridDim.x;
    }
    if (cur.ct != ct_loaded) load_filters(cur.ct);   // (uniform)

    // acc[c][cb]: position (wave, c), output channels 32 cb .. 32 cb + 31; register i = channel (i & 3) + 8 (i >> 2) + 4 half, lane j = tile
    f32x16 acc[4][2];
    // One item = chunk CH of the tile (input channels 16 CH .. 16 CH + 15 = k-steps 8 CH .. 8 CH + 7).  The four items of a tile
    // are four instantiations: the filter registers are indexed by compile-time constants only.  Buffers alternate with CH.
    auto item = [&](auto ch_tag, bool publish) {
      constexpr int CH = decltype(ch_tag)::value;
      constexpr bool FIRST = CH == 0;   // the first matrix instruction of every accumulator takes C = 0: nothing is ever zeroed
#ifdef WINO_STAMPS
      const unsigned long long ts0 = __builtin_amdgcn_s_memtime();
#endif
      if (WINO64_FLAGSYNC) {
        sync_wait();   // every wave has arrived for this item (the successor tile's id was written before thread 0's arrival of the tile's first item)
      } else {
        if (!drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        drained = false;
        if (publish && tid == 0) *sched_slot = dyn_fetch < band_hi(band) ? dyn_fetch : n_tiles;   // (the wait above covered the atomic's return)
        if (WINO64_ABL & 16) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        else asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      }
      if (publish) nxt_id = __builtin_amdgcn_readfirstlane(*reinterpret_cast<volatile int *>(sched_slot));
#ifdef WINO_STAMPS
      const unsigned long long ts2 = __builtin_amdgcn_s_memtime();
      st_bar += ts2 - ts0;
#endif
      const float *vb = smem + T::V_OFF + (CH & 1) * T::V_FLOATS;
      float *raw_next2 = smem + T::RAW_OFF + (CH & 1) * T::IN_FLOATS;            // raw(k+2) replaces raw(k), transformed during item k-1
      const float *raw_next = smem + T::RAW_OFF + ((CH + 1) & 1) * T::IN_FLOATS;
      float *v_next = smem + T::V_OFF + ((CH + 1) & 1) * T::V_FLOATS;
      const f32x4v *vb4 = reinterpret_cast<const f32x4v *>(vb) + (wave * 4) * 128 + lane;   // position (wave, c), half g: + (2 c + g) * 64
      f32x4v bv[4];   // [c]: the four k-steps of the current half; position c of the next half is read right after its last use
#pragma unroll
      for (int c = 0; c < 4; ++c) bv[c] = vb4[(2 * c + 0) * 64];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
#pragma unroll
        for (int q = 0; q < 32; ++q) {
          const int jj = q >> 3, c = (q >> 1) & 3, cb = q & 1, s = 8 * CH + 4 * g + jj, slot = 32 * g + q;
          if (FIRST && g == 0 && jj == 0) acc[c][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ureg[c][cb][s], bv[c][jj], f32x16{}, 0, 0, 0);
          else acc[c][cb] = __builtin_amdgcn_mfma_f32_32x32x2f32(ureg[c][cb][s], bv[c][jj], acc[c][cb], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (!(WINO64_ABL & 2) && slot >= 1 && slot < 1 + 2 * NIT_R && (slot & 1) && cr.id < n_tiles) issue_raw_piece(cr.t, cr.chunk, raw_next2, (slot - 1) >> 1);
          if (!(WINO64_ABL & 4) && g == 0 && q >= 25 && (q & 1)) bv[(q - 25) >> 1] = vb4[(2 * ((q - 25) >> 1) + 1) * 64];
          if (!(WINO64_ABL & 1) && slot >= 10 && slot < 50) xf_step(raw_next, v_next, slot - 10);   // input transform of the next item
          if (WINO64_FLAGSYNC && slot == 50) {
            if (CH == 0 && a.sched && tid == 0) *sched_slot = dyn_fetch < band_hi(band) ? dyn_fetch : n_tiles;   // (sync_arrive waits for the atomic's return)
            sync_arrive();
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      advance(cr);
#ifdef WINO_STAMPS
      st_mfma += __builtin_amdgcn_s_memtime() - ts2;
      ++st_items;
#endif
    };
    item(std::integral_constant<int, 0>{}, false);
    item(std::integral_constant<int, 1>{}, a.sched != nullptr);
    item(std::integral_constant<int, 2>{}, false);
    item(std::integral_constant<int, 3>{}, false);

    // everything in flight for the next item has landed before this tile's stores queue up behind it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drained = true;

#ifdef WINO_STAMPS
    const unsigned long long te0 = __builtin_amdgcn_s_memtime();
#endif
    // ---------------------------------------------------------------- epilogue: Y = A^T M A, bias, ReLU, (pool), store
    // row pass (local): Z[r][0] = (M[r][0] + M[r][1]) + M[r][2], Z[r][1] = (M[r][1] - M[r][2]) - M[r][3]  ->  LDS
    if (WINO64_ABL & 8) {   // (all eight accumulators stay live: nothing of the matrix stream may be optimised away)
#pragma unroll
      for (int c = 0; c < 4; ++c) { asm volatile("" :: "v"(acc[c][0])); asm volatile("" :: "v"(acc[c][1])); }
    } else {
    {
      f32x4v *zb = reinterpret_cast<f32x4v *>(smem + T::Z_OFF) + (wave * 16) * 64 + lane;   // [j][cb][quad]: + ((2 j + cb) * 4 + quad) * 64
#pragma unroll
      for (int cb = 0; cb < 2; ++cb)
#pragma unroll
        for (int quad = 0; quad < 4; ++quad) {
          f32x4v z0, z1;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const int i = 4 * quad + e;
            z0[e] = (acc[0][cb][i] + acc[1][cb][i]) + acc[2][cb][i];
            z1[e] = (acc[1][cb][i] - acc[2][cb][i]) - acc[3][cb][i];
          }
          zb[((0 + cb) * 4 + quad) * 64] = z0;
          zb[((2 + cb) * 4 + quad) * 64] = z1;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    // column pass: wave w finishes channel block fin_cb, register quads 2 fin_qh, 2 fin_qh + 1 of all 32 tiles
    {
      const f32x4v *zr = reinterpret_cast<const f32x4v *>(smem + T::Z_OFF) + lane;
      float *co_base = a.out + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)cur.ct * CO_TILE + fin_cb * 32) * out_plane;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
      const int oplane = (int)out_plane;
      const int kmax = a.cout - (cur.ct * CO_TILE + fin_cb * 32 + 4 * half);   // channels k < kmax of this lane's exist
      const int trow = j >> 4, tcol = j & 15;
      auto relu = [](float v) { return RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v; };
      unsigned voff;
      if constexpr (POOL) {
        const int y = (cur.y0 >> 1) + trow, x = (cur.x0 >> 1) + tcol;
        voff = (y < (a.H >> 1) && x < (a.W >> 1)) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
      } else {
        const int y = cur.y0 + 2 * trow, x = cur.x0 + 2 * tcol;   // H and W are even (host): a tile's four outputs are all inside or all outside
        voff = (y < a.H && x < a.W) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
      }
#pragma unroll
      for (int qq = 0; qq < 2; ++qq) {
        const int quad = 2 * fin_qh + qq;
        f32x4v z[4][2];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int jx = 0; jx < 2; ++jx) z[r][jx] = zr[(r * 16 + (2 * jx + fin_cb) * 4 + quad) * 64];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int k = e + 8 * quad;
          const float b = bias_v[4 * qq + e];
          const float y00 = relu(((z[0][0][e] + z[1][0][e]) + z[2][0][e]) + b), y01 = relu(((z[0][1][e] + z[1][1][e]) + z[2][1][e]) + b);
          const float y10 = relu(((z[1][0][e] - z[2][0][e]) - z[3][0][e]) + b), y11 = relu(((z[1][1][e] - z[2][1][e]) - z[3][1][e]) + b);
          const unsigned vo = k < kmax ? voff : OOB;
          if constexpr (POOL) {
            __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11))), rsrc, vo, k * oplane * 4, 0);
          } else {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 r0 = {__float_as_uint(y00), __float_as_uint(y01)}, r1 = {__float_as_uint(y10), __float_as_uint(y11)};
            __builtin_amdgcn_raw_buffer_store_b64(r0, rsrc, vo, k * oplane * 4, 0);
            __builtin_amdgcn_raw_buffer_store_b64(r1, rsrc, vo == OOB ? OOB : vo + 4u * (unsigned)a.out_wp, k * oplane * 4, 0);
          }
        }
      }
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
    unsigned long long *o = a.stamps + 8 * (blockIdx.x * 4 + wave);
    o[0] = st_dma; o[1] = st_bar; o[2] = st_mfma; o[3] = st_epi; o[4] = st_items;
    o[5] = __builtin_amdgcn_s_memtime() - st_t0; o[6] = __builtin_amdgcn_s_memrealtime() - st_r0;
  }
#endif
}

}  // namespace spvo
