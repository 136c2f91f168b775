// conv_wino.hip.h -- K2w: 3x3 convolution by the Winograd minimal-filtering algorithm F(2x2, 3x3) on the gfx950 fp32
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
// Workgroup = 4 waves = 64 output channels x 64 tiles (8 rows x 32 columns of output, the direct kernel's 8x32 tile):
// wave (cb, tb) owns channels [32 cb, 32 cb + 32) and the 32 tiles of tile rows {2 tb, 2 tb + 1} (lane j: tile row j >> 4,
// tile column j & 15), one 32x32 accumulator block per xi = 16 blocks = 256 registers.  The work of a workgroup is a linear
// sequence of items k = (tile, chunk of 8 input channels), software-pipelined three deep:
//   * LDS-DMA (global_load_lds): the host-transformed filters U[16][half 2][co 64][kk 4] of item k+1 and the raw halo tile
//     (8 x 10 x 40 floats, the direct kernel's layout) of item k+2 go out from inside the matrix stream of item k;
//   * input transform: raw tile of item k+1 -> V[16][half 2][tile 64][kk 4] (each thread 2 patches = 2 channels of one tile:
//     24 ds_read_b64, 64 adds, 16 ds_write_b64), in 40 micro-steps spread between the matrix instructions of item k;
//   * 64 matrix instructions of item k per wave (16 positions x 4 channel pairs ci = 2 kk + half), operands by conflict-free
//     ds_read_b128 (a lane's four channel pairs of one position are one 16-byte piece), read 3-4 instructions ahead.
// One barrier per item; the matrix pipe idles only there and in the epilogue.  The kernel must not spill: a scratch reload is a
// vector-memory load, its s_waitcnt vmcnt(0) also waits for the LDS-DMA in flight, and such waits inside the matrix stream
// cost 35 % (conv1b 440 us with 21 spilled registers, 296 us without).
// Epilogue: the inverse transform runs in registers (the 16 values of an output channel x tile sit in the same register of
// the 16 accumulator blocks), the bias enters through position (1,1), whose inverse-transform weight is +1 for all four
// outputs, and a 2x2 max-pool is the maximum of the tile's own four outputs: no cross-lane traffic at all.
//
// Numerics: the transforms use coefficients 0, +-1, +-1/2 only; against a float64 evaluation the layer's error is of the
// same order as the direct kernel's accumulated rounding, in fact smaller (tests/test_gpu_network.py: 1e-4 bar on every tensor
// against the oracle, and test_winograd_layers_stay_at_fp32_rounding_level against float64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>
#include <vector>
#include "conv_mfma.hip.h"

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

// OIHW weights + bias -> slabs [co_tile][chunk][xi 16][half 2][co 64][kk 4] of U = G g G^T (computed in double; input
// channel ci = 2 kk + half: a lane's four channel pairs sit in one 16-byte piece) + a bias row.
inline std::vector<float> pack_conv_weights_wino(const float *w, const float *bias, int cout, int cin) {
  constexpr int CK = WinoTile::CK;
  const int co_tiles = (cout + CO_TILE - 1) / CO_TILE, nch = cin / CK;
  std::vector<float> out((size_t)co_tiles * nch * WinoTile::W_FLOATS, 0.f);
  static const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int ch = 0; ch < nch; ++ch) {
      float *slab = out.data() + ((size_t)ct * nch + ch) * WinoTile::W_FLOATS;
      for (int o = 0; o < CO_TILE; ++o) {
        const int co = ct * CO_TILE + o;
        if (co >= cout) continue;
        for (int c = 0; c < CK; ++c) {
          const float *g = w + ((size_t)co * cin + ch * CK + c) * 9;
          double t[4][3];
          for (int a = 0; a < 4; ++a)
            for (int k = 0; k < 3; ++k) t[a][k] = G[a][0] * g[0 * 3 + k] + G[a][1] * g[1 * 3 + k] + G[a][2] * g[2 * 3 + k];
          for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b)
              slab[(((a * 4 + b) * 2 + (c & 1)) * CO_TILE + o) * 4 + (c >> 1)] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
        }
        if (ch == 0) slab[WinoTile::U_FLOATS + o] = bias[co];
      }
    }
  return out;
}

template <bool POOL, bool RELU, int TAG = 0, bool ODD = false>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a) {
  using T = WinoTile;
  constexpr int CK = T::CK, LW = T::LW, LH = T::LH, LW4 = LW / 4;
  constexpr int IN_V4 = T::IN_FLOATS / 4, W_V4 = T::W_FLOATS / 4;
  constexpr int NIT_R = (IN_V4 + 255) / 256, NIT_U = (W_V4 + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cb = wave & 1, tb = wave >> 1;
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
    t.w_base = a.wpack + (size_t)t.ct * a.n_chunks * T::W_FLOATS;
    return t;
  };

  // staging plans: this thread's 16-byte pieces of a raw tile / of a filter slab
  // Addresses are a wave-uniform 64-bit base (SGPRs) + an unsigned 32-bit byte offset per lane: the per-lane state of the
  // DMA is NIT_R + 1 registers.  (64-bit per-lane pointers spilled, and a scratch reload's vmcnt(0) between two LDS-DMA
  // instructions waits for the DMA itself.)
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
  auto issue_raw = [&](const TileRef &t, int chunk, float *buf) {
    const char *inb = reinterpret_cast<const char *>(t.in_base + (size_t)chunk * CK * in_plane);
#pragma unroll
    for (int it = 0; it < NIT_R; ++it)
      if (it < NIT_R - 1 || it * 256 + tid < IN_V4) glds16(reinterpret_cast<const float *>(inb + roff[it]), buf + (it * 256 + wave * 64) * 4);
  };
  const unsigned uoff = 16u * (unsigned)tid;
  auto issue_u = [&](const TileRef &t, int chunk, float *buf) {
    const char *wb = reinterpret_cast<const char *>(t.w_base + (size_t)chunk * T::W_FLOATS);
#pragma unroll
    for (int it = 0; it < NIT_U; ++it)
      if (it < NIT_U - 1 || it * 256 + tid < W_V4) glds16(reinterpret_cast<const float *>(wb + (uoff + 4096u * it)), buf + (it * 256 + wave * 64) * 4);
  };

  // input transform: this thread's two patches of a chunk = one tile, channels ci = half_t + 2 kk for kk = 2 p_t, 2 p_t + 1
  // (the pair shares one 8-byte piece of V[xi][half][tile][kk 4])
  const int t_tile = tid & 63, t_half = (tid >> 6) & 1, t_p = tid >> 7;
  int raw_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int ci = t_half + 2 * (2 * t_p + i);
    const int trow = t_tile >> 4, tcol = t_tile & 15;              // tile row 0..3 (= 2 tb + (j >> 4)), tile column 0..15
    raw_off[i] = ci * (LH * LW) + (2 * trow) * LW + 3 + 2 * tcol;  // LDS row 0 = output row y0 - 1, LDS column 4 = output column x0
  }
  const int v_off = (t_half * 64 + t_tile) * 4 + 2 * t_p;          // + xi * 512
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  auto transform = [&](const float *raw, float *vb) {
    float v0[16];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const float *d = raw + raw_off[i];
      float t[4][4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float d0 = d[0 * LW + q], d1 = d[1 * LW + q], d2 = d[2 * LW + q], d3 = d[3 * LW + q];
        t[0][q] = d0 - d2; t[1][q] = d1 + d2; t[2][q] = d2 - d1; t[3][q] = d1 - d3;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float w0 = t[r][0] - t[r][2], w1 = t[r][1] + t[r][2], w2 = t[r][2] - t[r][1], w3 = t[r][1] - t[r][3];
        if (i == 0) {
          v0[r * 4 + 0] = w0; v0[r * 4 + 1] = w1; v0[r * 4 + 2] = w2; v0[r * 4 + 3] = w3;
        } else {
          f32x2 *v = reinterpret_cast<f32x2 *>(vb + v_off);
          v[(r * 4 + 0) * 256] = f32x2{v0[r * 4 + 0], w0};
          v[(r * 4 + 1) * 256] = f32x2{v0[r * 4 + 1], w1};
          v[(r * 4 + 2) * 256] = f32x2{v0[r * 4 + 2], w2};
          v[(r * 4 + 3) * 256] = f32x2{v0[r * 4 + 3], w3};
        }
      }
    }
  };

  // The same transform in 40 micro-steps (20 per patch) for the matrix stream: 0..11 read the patch row u / 3 as three aligned
  // 8-byte pairs (columns c0-1 .. c0+4 of which c0 .. c0+3 are the patch: consecutive lanes read consecutive pairs, no bank
  // conflicts; 4-byte reads at the odd column c0 use only the 16 odd banks); 12..15 column q of B^T d; 16..19 row r of
  // (B^T d) B -> kept (patch 0) or written together with patch 0's (patch 1).
  f32x2 dp[12];
  float tt[16], v0s[16];
  auto xf_step = [&](const float *raw, float *vb, int st) {
    const int i = st / 20, u = st % 20;
    if (u < 12) {
      dp[u] = *reinterpret_cast<const f32x2 *>(raw + raw_off[i] - 1 + (u / 3) * LW + 2 * (u % 3));
    } else if (u < 16) {
      const int q = u - 12;
      auto d = [&](int r) { return q == 0 ? dp[3 * r][1] : q == 1 ? dp[3 * r + 1][0] : q == 2 ? dp[3 * r + 1][1] : dp[3 * r + 2][0]; };
      tt[0 * 4 + q] = d(0) - d(2);
      tt[1 * 4 + q] = d(1) + d(2);
      tt[2 * 4 + q] = d(2) - d(1);
      tt[3 * 4 + q] = d(1) - d(3);
    } else {
      const int r = u - 16;
      const float w0 = tt[r * 4 + 0] - tt[r * 4 + 2], w1 = tt[r * 4 + 1] + tt[r * 4 + 2], w2 = tt[r * 4 + 2] - tt[r * 4 + 1], w3 = tt[r * 4 + 1] - tt[r * 4 + 3];
      if (i == 0) {
        v0s[r * 4 + 0] = w0; v0s[r * 4 + 1] = w1; v0s[r * 4 + 2] = w2; v0s[r * 4 + 3] = w3;
      } else {
        f32x2 *v = reinterpret_cast<f32x2 *>(vb + v_off);
        v[(r * 4 + 0) * 256] = f32x2{v0s[r * 4 + 0], w0};
        v[(r * 4 + 1) * 256] = f32x2{v0s[r * 4 + 1], w1};
        v[(r * 4 + 2) * 256] = f32x2{v0s[r * 4 + 2], w2};
        v[(r * 4 + 3) * 256] = f32x2{v0s[r * 4 + 3], w3};
      }
    }
  };
  const int a_lane = half * CO_TILE + cb * 32 + j;   // 16-byte pieces in a filter slab: + xi * 128
  const int b_lane = half * 64 + tb * 32 + j;        // 16-byte pieces in a V buffer:    + xi * 128

#ifdef WINO_STAMPS   // diagnostic build (tools/wino_bench.hip -DWINO_STAMPS): 100 MHz wall-clock stamps per workgroup
  const unsigned long long stamp0 = __builtin_amdgcn_s_memrealtime();
  unsigned long long stamp_epi = 0;
#endif
  int tile_id = blockIdx.x;
  if (tile_id >= n_tiles) return;
  TileRef cur = decode(tile_id);

  // prefetch cursors over the item sequence: filters one item ahead, raw tiles two items ahead
  struct Cursor { TileRef t; int chunk, id; };
  auto advance = [&](Cursor &q) {
    if (++q.chunk == a.n_chunks) {
      q.chunk = 0;
      q.id += gridDim.x;
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
  // item 0's transform has nothing to hide behind
  transform(smem + T::RAW_OFF, smem + T::V_OFF);

  int k = 0;                    // items done: selects the buffers
  bool drained = true;          // the LDS-DMA this item needs has been waited for already
  constexpr unsigned OOB = 0xFFFFFFFFu;

  for (; tile_id < n_tiles; tile_id += gridDim.x) {
    const int next_id = tile_id + gridDim.x;
    TileRef nxt = cur;
    if (next_id < n_tiles) nxt = decode(next_id);

    // One item (tile, chunk).  FIRST = the tile's first chunk: the first matrix instruction of every accumulator takes C = 0, so
    // the 256 accumulator registers are never zeroed (v_mov / v_accvgpr_write went through 256 live VGPRs and made the kernel
    // spill; 16 zeroing matrix instructions cost 2.4 %).  The first chunk is peeled from the loop, not branched inside it, so
    // that no accumulator is conditionally defined.
    f32x16 acc[16];
    auto item = [&](auto first_tag) {
      constexpr bool FIRST = decltype(first_tag)::value;
      // Item k may start: its filters and the raw tile of item k+1 have landed (issued one item ago), every wave has
      // written its part of V(k) and finished the matrix work of item k-1.
      if (!drained) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      drained = false;
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const float *ub = smem + T::U_OFF + (k & 1) * T::W_FLOATS;
      const float *vb = smem + T::V_OFF + (k & 1) * T::V_FLOATS;
      float *u_next = smem + T::U_OFF + ((k + 1) & 1) * T::W_FLOATS;
      float *raw_next2 = smem + T::RAW_OFF + (k & 1) * T::IN_FLOATS;          // raw(k+2) replaces raw(k), transformed during item k-1
      const float *raw_next = smem + T::RAW_OFF + ((k + 1) & 1) * T::IN_FLOATS;
      float *v_next = smem + T::V_OFF + ((k + 1) & 1) * T::V_FLOATS;

      // 64 matrix instructions in 4 groups of 4 positions.  A group's operands are 8 ds_read_b128 (a lane's four channel
      // pairs of one position sit in one 16-byte piece of U and of V), read while the previous group multiplies; inside a
      // group the order is channel pair outer, position inner, so an accumulator is revisited after 3 other instructions.
      // The LDS-DMA of the following items and the 40 micro-steps of item k+1's input transform (harmless garbage in ->
      // garbage out when there is no item k+1) sit between the matrix instructions.  sched_barrier pins the order.
      typedef float f32x4v __attribute__((ext_vector_type(4)));
      const f32x4v *ub4 = reinterpret_cast<const f32x4v *>(ub), *vb4 = reinterpret_cast<const f32x4v *>(vb);
      {
        f32x4v av[4], bv[4];   // one register set: position x of the next group is read right after its last use in this one
        auto ld = [&](int g, int x) {
          av[x] = ub4[a_lane + (4 * g + x) * 128];
          bv[x] = vb4[b_lane + (4 * g + x) * 128];
        };
#pragma unroll
        for (int x = 0; x < 4; ++x) ld(0, x);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
#pragma unroll
          for (int q = 0; q < 16; ++q) {
            const int kk = q >> 2, x = q & 3;
            if (FIRST && kk == 0) acc[4 * g + x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][kk], bv[x][kk], f32x16{}, 0, 0, 0);
            else acc[4 * g + x] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[x][kk], bv[x][kk], acc[4 * g + x], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (g == 0 && q == 1 && cu.id < n_tiles) issue_u(cu.t, cu.chunk, u_next);
            if (g == 0 && q == 3 && cr.id < n_tiles) issue_raw(cr.t, cr.chunk, raw_next2);
            if (g < 3 && q >= 12) ld(g + 1, q - 12);   // 3 matrix instructions (192 cycles) before its first use
            if (16 * g + q >= 8 && 16 * g + q < 48) xf_step(raw_next, v_next, 16 * g + q - 8);   // input transform of item k+1
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
      if (FIRST) {   // bias through position (1,1), whose inverse-transform weight is +1 for all four outputs: A = (bias, 0), B = (1, 1)
        const float bias_a = half ? 0.f : ub[T::U_FLOATS + cb * 32 + j];
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, 1.0f, acc[5], 0, 0, 0);
      }
      advance(cu);
      advance(cr);
      ++k;
    };
    item(std::true_type{});
    for (int c = 1; c < a.n_chunks; ++c) item(std::false_type{});

    // everything in flight for the next item has landed before this tile's stores queue up behind it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    drained = true;

#ifdef WINO_STAMPS
    stamp_epi = __builtin_amdgcn_s_memrealtime();
#endif
    // ---------------------------------------------------------------- epilogue: Y = A^T M A, ReLU, (pool), store
    float *co_base = a.out + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)cur.ct * CO_TILE + cb * 32) * out_plane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
    const int oplane = (int)out_plane;
    const int kmax = a.cout - (cur.ct * CO_TILE + cb * 32 + 4 * half);   // channels k < kmax of this wave's 32 exist for this lane
    const int trow = 2 * tb + (j >> 4), tcol = j & 15;
    auto relu = [](float v) { return RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v; };
    unsigned voff, voff01 = OOB, voff10 = OOB, voff11 = OOB;
    if constexpr (POOL) {
      const int y = (cur.y0 >> 1) + trow, x = (cur.x0 >> 1) + tcol;
      voff = (y < (a.H >> 1) && x < (a.W >> 1)) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
    } else {
      const int y = cur.y0 + 2 * trow, x = cur.x0 + 2 * tcol;
      voff = (y < a.H && x < a.W) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
      if constexpr (ODD) {
        voff01 = (voff != OOB && x + 1 < a.W) ? voff + 4u : OOB;
        voff10 = (voff != OOB && y + 1 < a.H) ? voff + 4u * (unsigned)a.out_wp : OOB;
        voff11 = (voff01 != OOB && voff10 != OOB) ? voff10 + 4u : OOB;
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2);
      // two-wide (v_pk_add_f32): rows of M in pairs for the column pass, (y00, y01) / (y10, y11) for the row pass
      f32x2 sa[2], sb[2];   // sa[h] = (s0[2h], s0[2h+1]), sb[h] = (s1[2h], s1[2h+1])
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 m0 = {acc[8 * h + 0][r], acc[8 * h + 4][r]}, m1 = {acc[8 * h + 1][r], acc[8 * h + 5][r]};
        const f32x2 m2 = {acc[8 * h + 2][r], acc[8 * h + 6][r]}, m3 = {acc[8 * h + 3][r], acc[8 * h + 7][r]};
        sa[h] = (m0 + m1) + m2;
        sb[h] = (m1 - m2) - m3;
      }
      const f32x2 q0 = {sa[0][0], sb[0][0]}, q1 = {sa[0][1], sb[0][1]}, q2 = {sa[1][0], sb[1][0]}, q3 = {sa[1][1], sb[1][1]};
      const f32x2 ya = (q0 + q1) + q2, yb = (q1 - q2) - q3;   // ya = (y00, y01), yb = (y10, y11)
      const float y00 = relu(ya[0]), y01 = relu(ya[1]), y10 = relu(yb[0]), y11 = relu(yb[1]);
      const unsigned vo = k < kmax ? voff : OOB;
      if constexpr (POOL) {
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11))), rsrc, vo, k * oplane * 4, WINO_STORE_AUX);
      } else if constexpr (!ODD) {
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        const u32x2 r0 = {__float_as_uint(y00), __float_as_uint(y01)}, r1 = {__float_as_uint(y10), __float_as_uint(y11)};
        __builtin_amdgcn_raw_buffer_store_b64(r0, rsrc, vo, k * oplane * 4, WINO_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b64(r1, rsrc, vo == OOB ? OOB : vo + 4u * (unsigned)a.out_wp, k * oplane * 4, WINO_STORE_AUX);
      } else {
        // odd H or W: the second row / column of the last tiles is outside the image and must stay zero (it is the next
        // layer's halo), so the four outputs leave one by one
        const unsigned v00 = k < kmax ? voff : OOB, v01 = k < kmax ? voff01 : OOB, v10 = k < kmax ? voff10 : OOB, v11 = k < kmax ? voff11 : OOB;
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y00), rsrc, v00, k * oplane * 4, WINO_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y01), rsrc, v01, k * oplane * 4, WINO_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y10), rsrc, v10, k * oplane * 4, WINO_STORE_AUX);
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(y11), rsrc, v11, k * oplane * 4, WINO_STORE_AUX);
      }
      __builtin_amdgcn_sched_barrier(0);   // one register index at a time: 16 accumulator reads live, not 256
    }
    cur = nxt;
  }
#ifdef WINO_STAMPS
  if (tid == 0 && a.stamps) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    a.stamps[3 * blockIdx.x] = stamp0;
    a.stamps[3 * blockIdx.x + 1] = stamp_epi;   // start of the last epilogue
    a.stamps[3 * blockIdx.x + 2] = __builtin_amdgcn_s_memrealtime();
  }
#endif
}

}  // namespace spvo
