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
// tile column j & 15), one 32x32 accumulator block per xi = 16 blocks = 256 registers.  Per chunk of 8 input channels:
//   1. the raw halo tile (8 x 10 x 40 floats, the direct kernel's LDS layout) and the host-transformed filters
//      U[16][8][64] arrive by global_load_lds into a 2-deep ring, prefetched one chunk (and one tile) ahead;
//   2. the 4 waves transform the raw tile to V[16][8][64] in LDS (each thread 2 patches: 16 reads, 32 adds, 16 writes);
//   3. 64 matrix instructions per wave (4 channel pairs x 16 positions), operands by conflict-free ds_read_b32.
// Epilogue: the inverse transform runs in registers (the 16 values of an output channel x tile sit in the same register of
// the 16 accumulator blocks), the bias enters through position (1,1), whose inverse-transform weight is +1 for all four
// outputs, and a 2x2 max-pool is the maximum of the tile's own four outputs: no cross-lane traffic at all.
//
// Numerics: the transforms use coefficients 0, +-1, +-1/2 only; against a float64 evaluation the layer's error is of the
// same order as the direct kernel's accumulated rounding (tests/test_gpu_network.py, 1e-4 bar on every tensor).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "conv_mfma.hip.h"

namespace spvo {

struct WinoTile {
  static constexpr int CK = 8, TH = 8, TW = 32, LW = TW + 8, LH = TH + 2;
  static constexpr int IN_FLOATS = CK * LH * LW;            // 3200: raw halo tile, row = x0-4 .. x0+35
  static constexpr int U_FLOATS = 16 * CK * CO_TILE;        // 8192
  static constexpr int W_FLOATS = U_FLOATS + CO_TILE;       // + the bias row (chunk 0's slab)
  static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;   // one ring buffer (LDS-DMA target)
  static constexpr int V_FLOATS = 16 * CK * 64;             // transformed input, single buffer
  static constexpr int LDS_BYTES = (2 * BUF_FLOATS + V_FLOATS) * 4;   // 124 416
};

// OIHW weights + bias -> slabs [co_tile][chunk][xi 16][ci 8][co 64] of U = G g G^T (computed in double) + a bias row.
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
              slab[((a * 4 + b) * CK + c) * CO_TILE + o] = (float)(t[a][0] * G[b][0] + t[a][1] * G[b][1] + t[a][2] * G[b][2]);
        }
        if (ch == 0) slab[WinoTile::U_FLOATS + o] = bias[co];
      }
    }
  return out;
}

template <bool POOL, bool RELU, int TAG = 0>
__global__ __launch_bounds__(256, 1) void conv_wino_kernel(const ConvArgs a) {
  using T = WinoTile;
  constexpr int CK = T::CK, LW = T::LW, LH = T::LH, LW4 = LW / 4;
  constexpr int IN_V4 = T::IN_FLOATS / 4, W_V4 = T::W_FLOATS / 4, TOT_V4 = IN_V4 + W_V4;
  constexpr int NIT = (TOT_V4 + 255) / 256;

  extern __shared__ __attribute__((aligned(16))) float smem[];
  float *const vbuf = smem + 2 * T::BUF_FLOATS;

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

  int piece_off[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * 256 + tid;
    if (idx < IN_V4) {
      const int ci = idx / (LH * LW4);
      const int rem = idx - ci * (LH * LW4);
      const int r = rem / LW4;
      const int q = rem - r * LW4;
      piece_off[it] = ci * (int)in_plane + r * a.in_wp + q * 4;
    } else {
      piece_off[it] = (min(idx, TOT_V4 - 1) - IN_V4) * 4;
    }
  }
  auto issue = [&](const TileRef &t, int chunk, float *buf) {
    const float *inb = t.in_base + (size_t)chunk * CK * in_plane;
    const float *wb = t.w_base + (size_t)chunk * T::W_FLOATS;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 256 + tid;
      const float *src = ((idx < IN_V4) ? inb : wb) + piece_off[it];
      if (it < NIT - 1 || idx < TOT_V4) glds16(src, buf + (it * 256 + wave * 64) * 4);
    }
  };

  // input transform: this thread's two (tile, channel) patches of a chunk
  int raw_off[2], v_off[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int p = tid + 256 * i;
    const int tile = p & 63, ci = p >> 6;
    const int trow = tile >> 4, tcol = tile & 15;                 // tile row 0..3 (= 2 tb + (j >> 4)), tile column 0..15
    raw_off[i] = ci * (LH * LW) + (2 * trow) * LW + 3 + 2 * tcol;  // LDS row 0 = output row y0 - 1, LDS column 4 = output column x0
    v_off[i] = ci * 64 + tile;
  }
  const int a_lane = T::IN_FLOATS + half * CO_TILE + cb * 32 + j;   // + (xi * CK + 2 kk) * 64
  const int b_lane = half * 64 + tb * 32 + j;                        // + (xi * CK + 2 kk) * 64

  int tile_id = blockIdx.x;
  if (tile_id >= n_tiles) return;
  TileRef cur = decode(tile_id);
  issue(cur, 0, smem);
  int ring = 0;
  bool first_landed = false;
  constexpr unsigned OOB = 0xFFFFFFFFu;

  for (; tile_id < n_tiles; tile_id += gridDim.x) {
    const int next_id = tile_id + gridDim.x;
    TileRef nxt = cur;
    if (next_id < n_tiles) nxt = decode(next_id);

    f32x16 acc[16];
#pragma unroll
    for (int x = 0; x < 16; ++x)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[x][r] = 0.f;

    for (int c = 0; c < a.n_chunks; ++c, ++ring) {
      // chunk `ring` has landed and every wave is done with the previous chunk's matrix work (V and the other ring buffer
      // are free).  At a tile's first chunk the DMA wait already happened in front of the previous tile's epilogue.
      if (c > 0 || !first_landed) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_barrier" ::: "memory");
      float *nbuf = smem + ((ring + 1) & 1) * T::BUF_FLOATS;
      const float *buf = smem + (ring & 1) * T::BUF_FLOATS;

      if (c == 0) {   // bias through position (1,1): A = (bias, 0), B = (1, 1), C = 0
        const float bias_a = half ? 0.f : buf[T::IN_FLOATS + T::U_FLOATS + cb * 32 + j];
        acc[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a, 1.0f, acc[5], 0, 0, 0);
      }

      // ---- input transform V = B^T d B
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const float *d = buf + raw_off[i];
        float t[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float d0 = d[0 * LW + q], d1 = d[1 * LW + q], d2 = d[2 * LW + q], d3 = d[3 * LW + q];
          t[0][q] = d0 - d2; t[1][q] = d1 + d2; t[2][q] = d2 - d1; t[3][q] = d1 - d3;
        }
        float *v = vbuf + v_off[i];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          v[((r * 4 + 0) * CK) * 64] = t[r][0] - t[r][2];
          v[((r * 4 + 1) * CK) * 64] = t[r][1] + t[r][2];
          v[((r * 4 + 2) * CK) * 64] = t[r][2] - t[r][1];
          v[((r * 4 + 3) * CK) * 64] = t[r][1] - t[r][3];
        }
      }
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");   // V is complete; not __syncthreads(): its fence would drain the LDS-DMA

      // ---- 16 GEMMs, 4 channel pairs each.  Operands are read PF instructions ahead (an LDS read takes longer than one
      // matrix instruction runs); sched_barrier pins the order: one matrix instruction, then the two reads for PF later.
      constexpr int PF = 4;
      float av[PF], bv[PF];
      auto ld = [&](int s) {
        const int kk = s >> 4, xi = s & 15;
        av[s % PF] = buf[a_lane + (xi * CK + 2 * kk) * 64];
        bv[s % PF] = vbuf[b_lane + (xi * CK + 2 * kk) * 64];
      };
#pragma unroll
      for (int s = 0; s < PF - 1; ++s) ld(s);
#pragma unroll
      for (int s = 0; s < 64; ++s) {
        if (s + PF - 1 < 64) ld(s + PF - 1);
        if (s == 1) {   // the next chunk's LDS-DMA goes out from inside the matrix stream (issued in front of the transform
                        // it stalls the LDS instructions behind it: 572 -> 450 us on conv1b)
          if (c + 1 < a.n_chunks) issue(cur, c + 1, nbuf);
          else if (next_id < n_tiles) issue(nxt, 0, nbuf);      // first chunk of the NEXT tile
        }
        __builtin_amdgcn_sched_barrier(0);
        acc[s & 15] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[s % PF], bv[s % PF], acc[s & 15], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // the next tile's first chunk has landed before this tile's stores queue up behind it
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first_landed = true;

    // ---------------------------------------------------------------- epilogue: Y = A^T M A, ReLU, (pool), store
    float *co_base = a.out + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)cur.ct * CO_TILE + cb * 32) * out_plane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
    const int oplane = (int)out_plane;
    const int kmax = a.cout - (cur.ct * CO_TILE + cb * 32 + 4 * half);   // channels k < kmax of this wave's 32 exist for this lane
    const int trow = 2 * tb + (j >> 4), tcol = j & 15;
    auto relu = [](float v) { return RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v; };
    unsigned voff;
    if constexpr (POOL) {
      const int y = (cur.y0 >> 1) + trow, x = (cur.x0 >> 1) + tcol;
      voff = (y < (a.H >> 1) && x < (a.W >> 1)) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
    } else {
      const int y = cur.y0 + 2 * trow, x = cur.x0 + 2 * tcol;
      voff = (y < a.H && x < a.W) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = (r & 3) + 8 * (r >> 2);
      float s0[4], s1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float m0 = acc[4 * q + 0][r], m1 = acc[4 * q + 1][r], m2 = acc[4 * q + 2][r], m3 = acc[4 * q + 3][r];
        s0[q] = (m0 + m1) + m2;
        s1[q] = (m1 - m2) - m3;
      }
      const float y00 = relu((s0[0] + s0[1]) + s0[2]), y01 = relu((s1[0] + s1[1]) + s1[2]);
      const float y10 = relu((s0[1] - s0[2]) - s0[3]), y11 = relu((s1[1] - s1[2]) - s1[3]);
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
    cur = nxt;
  }
}

}  // namespace spvo
