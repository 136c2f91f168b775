// conv_f16.hip.h -- the FP16 engines (BASELINE config 3): convolutions with fp16 storage and
// fp32 accumulation on v_mfma_f32_32x32x16_f16.
//
// Replaces what the reference gets from a TensorRT engine built with `--fp16`
// (scripts/engine_generation.py:13-56, loaded by feature_detection_neural_network.cpp:44-49,
// enqueued at :169): network input and the two outputs stay fp32 (`binding_size *= sizeof(float)`,
// nn.cpp:117), everything in between is half precision.
//
// Activation layout "C8": act[img][C/8][Hp][Wp][8] of _Float16 -- the 8 channels of a group sit next to
// each other (16 bytes per pixel), groups are padded planes with the same zero border as the fp32
// planes (pixel (y, x) at [y + PADY][x + PADX]).  That is exactly the operand shape of the matrix
// instruction: lane l of the B operand holds 8 consecutive k = 8 channels of ONE pixel, lane l of the
// A operand 8 consecutive input channels of ONE output channel, so every operand is one
// ds_read_b128 and consecutive lanes read consecutive 16-byte pieces (no bank conflicts).
//
// GEMM mapping (as in conv_mfma.hip.h): D[co][x], A = weights, B = activations; a k-step is one
// filter tap x 16 input channels (lane half h supplies channel group 2s + h).  A workgroup is 4 waves =
// 64 output channels x (4*WR rows) x (32*WC columns); the reduction runs over chunks of CKG channel
// groups whose halo tile and weight slab arrive by global_load_lds_dwordx4 in a 2-deep LDS ring.
// The kernel is persistent with cross-tile prefetch like the fp32 one.
//
// At 16x the fp32 matrix rate these layers are bound by staging bandwidth (L2 -> LDS) and HBM, not
// by the matrix pipe: DESIGN.md section 7.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "conv_mfma.hip.h"

namespace spvo {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct ConvArgs16 {
  const _Float16 *in;     // C8 tensor, image 0, group 0
  void *out;              // C8 tensor (_Float16) or fp32 padded planes (OUT_F32)
  const _Float16 *wpack;  // [co_tiles][n_chunks] slabs: pack_conv_weights_f16()
  int H, W;               // conv resolution (input == pre-pool output)
  int in_hp, in_wp, in_gtot, in_goff;     // input geometry in pixels; channel groups of the tensor / first group read
  int out_hp, out_wp, out_ctot, out_coff; // output geometry; channels of the tensor / first channel written
  int cout;               // real output channels of this op (a multiple of 8 unless OUT_F32)
  int n_chunks;           // cin / (8 * CKG)
  int tiles_x, tiles_y, co_tiles;
  int batch;
  // MobileNet epilogues (EPI template parameter), applied per element in fp32 BEFORE pooling and the fp16 rounding
  const float *bn_scale = nullptr, *bn_shift = nullptr;  // EPI 1: v = relu(v * scale[co] + shift[co]), [co_tiles*64]
  const _Float16 *residual = nullptr;                    // EPI 2: v = relu(v + residual[co][y][x]); C8 tensor of the input's
                                                         //        geometry (in_hp x in_wp) with cout channels, image 0
};

template <int KS, int CKG_, int WR, int WC>
struct ConvTile16 {
  static constexpr int CKG = CKG_;                      // channel groups (of 8) per chunk: 2 for 3x3; 4 (or 2, cin % 32 != 0) for 1x1
  static constexpr int TH = 4 * WR, TW = 32 * WC, HALO = KS / 2;
  static constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
  static constexpr int IN_P = CKG * LH * LW;            // 16-byte pieces: one pixel of one group
  static constexpr int W_P = KS * KS * CKG * CO_TILE;   // one (tap, group, co) row of 8 halfs
  static constexpr int BIAS_P = CO_TILE / 4;            // 64 fp32 biases behind the weights (chunk 0's slab)
  static constexpr int BUF_P = IN_P + W_P + BIAS_P;
  static constexpr int LDS_BYTES = 2 * BUF_P * 16;
  static constexpr int NSTEP = KS * KS * (CKG / 2);
};

// Host side: OIHW fp32 weights + bias -> slabs [co_tile][chunk][tap][group][co 64][8] of fp16 (round to nearest even,
// what numpy's astype(float16) does in the oracle) followed by 64 fp32 biases (chunk 0; zeros elsewhere).
inline std::vector<_Float16> pack_conv_weights_f16(const float *w, const float *bias, int cout, int cin, int ks, int ckg) {
  const int co_tiles = (cout + CO_TILE - 1) / CO_TILE, ck = 8 * ckg, nch = cin / ck, taps = ks * ks;
  const size_t slab = ((size_t)taps * ckg * CO_TILE + CO_TILE / 4) * 8;   // in halfs
  std::vector<_Float16> out((size_t)co_tiles * nch * slab, (_Float16)0.f);
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int ch = 0; ch < nch; ++ch) {
      _Float16 *s = out.data() + ((size_t)ct * nch + ch) * slab;
      for (int o = 0; o < CO_TILE; ++o) {
        const int co = ct * CO_TILE + o;
        if (co >= cout) continue;
        for (int t = 0; t < taps; ++t)
          for (int g = 0; g < ckg; ++g)
            for (int e = 0; e < 8; ++e)
              s[(((size_t)t * ckg + g) * CO_TILE + o) * 8 + e] = (_Float16)w[((size_t)co * cin + ch * ck + g * 8 + e) * taps + t];
        if (ch == 0) reinterpret_cast<float *>(s + (size_t)taps * ckg * CO_TILE * 8)[o] = bias[co];
      }
    }
  return out;
}

template <int KS, int CKG_, int WR, int WC, bool POOL, bool RELU, bool OUT_F32, int EPI = 0>
__global__ __launch_bounds__(256) void conv_f16_kernel(const ConvArgs16 a) {
  using T = ConvTile16<KS, CKG_, WR, WC>;
  constexpr int NT = WR * WC, CKG = T::CKG, LW = T::LW, LH = T::LH, NSTEP = T::NSTEP;
  constexpr int TOT_P = T::BUF_P;
  constexpr int NIT = (TOT_P + 255) / 256;
  static_assert(!POOL || WR == 2, "fused pooling needs both rows of a 2x2 window in one wave");
  static_assert(!(POOL && OUT_F32), "the fp32-output layers are the unpooled heads");
  static_assert(EPI == 0 || !OUT_F32, "BatchNorm / residual epilogues write C8");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem16[];   // 2 buffers of BUF_P 16-byte pieces

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t in_plane = (size_t)a.in_hp * a.in_wp;      // pixels per group plane
  const size_t out_plane = (size_t)a.out_hp * a.out_wp;
  const int n_tiles = a.tiles_x * a.tiles_y * a.co_tiles * a.batch;

  struct TileRef { const _Float16 *in_base, *w_base; int x0, y0, ct, img; };
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
    t.in_base = a.in + (((size_t)t.img * a.in_gtot + a.in_goff) * in_plane + (size_t)(t.y0 + PADY - T::HALO) * a.in_wp + (t.x0 + PADX - T::HALO)) * 8;
    t.w_base = a.wpack + (size_t)t.ct * a.n_chunks * (T::W_P + T::BIAS_P) * 8;
    return t;
  };

  // staging plan: this thread's 16-byte pieces of a chunk (input halo tile first, then the weight slab)
  int piece_off[NIT];   // in halfs, relative to the chunk's input / weight base
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * 256 + tid;
    if (idx < T::IN_P) {
      const int g = idx / (LH * LW);
      const int rem = idx - g * (LH * LW);
      const int r = rem / LW;
      const int q = rem - r * LW;
      piece_off[it] = (g * (int)in_plane + r * a.in_wp + q) * 8;
    } else {
      piece_off[it] = (min(idx, TOT_P - 1) - T::IN_P) * 8;
    }
  }
  auto issue = [&](const TileRef &t, int chunk, unsigned char *buf) {
    const _Float16 *inb = t.in_base + (size_t)chunk * CKG * in_plane * 8;
    const _Float16 *wb = t.w_base + (size_t)chunk * (T::W_P + T::BIAS_P) * 8;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 256 + tid;
      const _Float16 *src = ((idx < T::IN_P) ? inb : wb) + piece_off[it];
      if (it < NIT - 1 || idx < TOT_P)
        glds16(reinterpret_cast<const float *>(src), reinterpret_cast<float *>(buf + (size_t)(it * 256 + wave * 64) * 16));
    }
  };

  // per-lane LDS piece indices of the operands at (tap 0, group pair 0)
  const int b_lane = (half * LH + wave * WR) * LW + j;                 // + (2s*LH + rr + ky) * LW + cc*32 + kx
  const int a_lane = T::IN_P + half * CO_TILE + j;                     // + ((tap*CKG + 2s) * 64 + 32m)

  int tile_id = blockIdx.x;
  if (tile_id >= n_tiles) return;
  TileRef cur = decode(tile_id);
  issue(cur, 0, smem16);
  int ring = 0;
  bool first_landed = false;
  constexpr unsigned OOB = 0xFFFFFFFFu;

  for (; tile_id < n_tiles; tile_id += gridDim.x) {
    const int next_id = tile_id + gridDim.x;
    TileRef nxt = cur;
    if (next_id < n_tiles) nxt = decode(next_id);

    if (!first_landed) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_barrier" ::: "memory");

    // accumulators start from the fp32 bias row that sits behind chunk 0's weights
    f32x16 acc[2][NT];
    {
      const f32x4 *bp = reinterpret_cast<const f32x4 *>(smem16 + (size_t)(ring & 1) * TOT_P * 16 + (size_t)(T::IN_P + T::W_P) * 16);
#pragma unroll
      for (int m = 0; m < 2; ++m) {
        f32x16 bv;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 q = bp[(32 * m + 8 * g + 4 * half) / 4];
          bv[4 * g + 0] = q[0]; bv[4 * g + 1] = q[1]; bv[4 * g + 2] = q[2]; bv[4 * g + 3] = q[3];
        }
#pragma unroll
        for (int n = 0; n < NT; ++n) acc[m][n] = bv;
      }
    }

    for (int c = 0; c < a.n_chunks; ++c, ++ring) {
      if (c > 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      unsigned char *nbuf = smem16 + (size_t)((ring + 1) & 1) * TOT_P * 16;
      const half8 *buf = reinterpret_cast<const half8 *>(smem16 + (size_t)(ring & 1) * TOT_P * 16);
      half8 av[2][2], bv[2][NT];
      auto load_step = [&](int st, int slot) {
        const int t = st / (CKG / 2), s = st % (CKG / 2);
        const int ky = t / KS, kx = t % KS;
#pragma unroll
        for (int m = 0; m < 2; ++m) av[slot][m] = buf[a_lane + (t * CKG + 2 * s) * CO_TILE + 32 * m];
#pragma unroll
        for (int rr = 0; rr < WR; ++rr)
#pragma unroll
          for (int cc = 0; cc < WC; ++cc)
            bv[slot][rr * WC + cc] = buf[b_lane + (2 * s * LH + rr + ky) * LW + cc * 32 + kx];
      };
      load_step(0, 0);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int cs = st & 1;
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[cs][0], bv[cs][0], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st == 0) {
          if (c + 1 < a.n_chunks) issue(cur, c + 1, nbuf);
          else if (next_id < n_tiles) issue(nxt, 0, nbuf);
        }
        if (st + 1 < NSTEP) load_step(st + 1, cs ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if (m + n > 0) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_f16(av[cs][m], bv[cs][n], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next tile's first chunk has landed before the stores queue up
    first_landed = true;

    // ------------------------------------------------------------------ epilogue
    auto relu = [](float v) { return RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v; };
    if constexpr (OUT_F32) {
      // fp32 padded planes (the two network outputs): as in conv_mfma.hip.h
      float *co_base = reinterpret_cast<float *>(a.out) + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)cur.ct * CO_TILE) * out_plane;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
      const int oplane = (int)out_plane;
      const int kmax = a.cout - (cur.ct * CO_TILE + 4 * half);
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int rr = 0; rr < WR; ++rr)
#pragma unroll
          for (int cc = 0; cc < WC; ++cc) {
            const int y = cur.y0 + wave * WR + rr, x = cur.x0 + cc * 32 + j;
            const unsigned voff = ((y < a.H) && (x < a.W)) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              const int k = 32 * m + (r & 3) + 8 * (r >> 2);
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(relu(acc[m][rr * WC + cc][r])), rsrc, k < kmax ? voff : OOB, k * oplane * 4, 0);
            }
          }
    } else {
      // C8 fp16: registers 4g .. 4g+3 of a lane are 4 consecutive channels of group 4m + g -> one 8-byte store;
      // the two lane halves fill the two halves of the pixel's 16 bytes
      _Float16 *g_base = reinterpret_cast<_Float16 *>(a.out) + ((size_t)cur.img * (a.out_ctot / 8) + a.out_coff / 8 + (size_t)cur.ct * (CO_TILE / 8)) * out_plane * 8;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(g_base, 0, 0x7FFFFFFF, 0x00020000);
      const int groups_valid = (a.cout - cur.ct * CO_TILE + 7) / 8;   // groups of this co tile that exist
      // element-wise tail in graph order: ReLU, then (EPI 1) BatchNorm + ReLU or (EPI 2) residual + ReLU; register r of
      // accumulator row block m is channel ct*64 + 32m + (r&3) + 8(r>>2) + 4*half of pixel (y, x) of the INPUT resolution
      const size_t res_plane = (size_t)a.in_hp * a.in_wp;
      auto tail = [&](float v, int m, int r, int y, int x) -> float {
        v = relu(v);
        if constexpr (EPI != 0) {
          const int co = cur.ct * CO_TILE + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
          if constexpr (EPI == 1) v = __builtin_amdgcn_fmed3f(fmaf(v, a.bn_scale[co], a.bn_shift[co]), 0.f, __builtin_inff());
          if constexpr (EPI == 2) {
            // the padded plane covers the whole tile (zeros outside the image); channels past cout do not exist
            const float rv = co < a.cout ? (float)a.residual[((((size_t)cur.img * (a.cout / 8) + co / 8) * res_plane + (size_t)(y + PADY) * a.in_wp + (x + PADX)) * 8) + (co & 7)] : 0.f;
            v = __builtin_amdgcn_fmed3f(v + rv, 0.f, __builtin_inff());
          }
        }
        return v;
      };
      auto store_tile = [&](const float (&v)[16], int m, unsigned voff) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (4 * m + g < groups_valid) {
            half4 hv;
            hv[0] = (_Float16)v[4 * g + 0]; hv[1] = (_Float16)v[4 * g + 1]; hv[2] = (_Float16)v[4 * g + 2]; hv[3] = (_Float16)v[4 * g + 3];
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, hv), rsrc, voff, (4 * m + g) * (int)out_plane * 16, 0);
          }
        }
      };
      if constexpr (!POOL) {
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int rr = 0; rr < WR; ++rr)
#pragma unroll
            for (int cc = 0; cc < WC; ++cc) {
              const int y = cur.y0 + wave * WR + rr, x = cur.x0 + cc * 32 + j;
              const unsigned voff = ((y < a.H) && (x < a.W)) ? (unsigned)(((y + PADY) * a.out_wp + (x + PADX)) * 16 + 8 * half) : OOB;
              float v[16];
#pragma unroll
              for (int r = 0; r < 16; ++r) v[r] = tail(acc[m][rr * WC + cc][r], m, r, y, x);
              store_tile(v, m, voff);
            }
      } else {
        const int OH = a.H >> 1, OW = a.W >> 1;
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int cc = 0; cc < WC; ++cc) {
            const int y = (cur.y0 >> 1) + wave, x = (cur.x0 + cc * 32 + j) >> 1;
            const unsigned voff = ((y < OH) && (x < OW) && !(j & 1)) ? (unsigned)(((y + PADY) * a.out_wp + (x + PADX)) * 16 + 8 * half) : OOB;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float q;
              if constexpr (EPI == 0) {
                q = fmaxf(acc[m][0 * WC + cc][r], acc[m][1 * WC + cc][r]);   // ReLU commutes with max
                q = relu(fmaxf(q, __shfl_xor(q, 1)));
              } else {
                const int yi = cur.y0 + wave * 2, xi = cur.x0 + cc * 32 + j;
                q = fmaxf(tail(acc[m][0 * WC + cc][r], m, r, yi, xi), tail(acc[m][1 * WC + cc][r], m, r, yi + 1, xi));
                q = fmaxf(q, __shfl_xor(q, 1));
              }
              v[r] = q;
            }
            store_tile(v, m, voff);
          }
      }
    }
    cur = nxt;
  }
}

// Layers of an FP16 engine with Cin = 1 (fp32 plane in: the network input, nn.cpp:117, or mbv's one-channel
// stem) and a C8 fp16 output: fp32 arithmetic on fp16-rounded weights; one thread = one pixel, 8 channels = one
// 16-byte store.  bn_scale != nullptr: BatchNorm + ReLU after the activation (mbv1's second layer).
template <int KS, bool RELU>
__global__ __launch_bounds__(256) void conv_first_f16_kernel(const float *__restrict__ in, _Float16 *__restrict__ out,
                                                              const float *__restrict__ w,     // [cout][KS*KS], already fp16-rounded values
                                                              const float *__restrict__ bias, const float *__restrict__ bn_scale,
                                                              const float *__restrict__ bn_shift, int H, int W, int hp, int wp, int out_gtot,
                                                              int out_goff, int cout) {
  constexpr int TAPS = KS * KS, HALO = KS / 2;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int img = blockIdx.z;
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)hp * wp;
  const float *ip = in + (size_t)img * plane + (size_t)(y + PADY - HALO) * wp + (x + PADX - HALO);
  float v[TAPS];
#pragma unroll
  for (int ky = 0; ky < KS; ++ky)
#pragma unroll
    for (int kx = 0; kx < KS; ++kx) v[ky * KS + kx] = ip[ky * wp + kx];
  half8 *op = reinterpret_cast<half8 *>(out) + ((size_t)img * out_gtot + out_goff) * plane + (size_t)(y + PADY) * wp + (x + PADX);
  for (int g = 0; g < cout / 8; ++g) {
    half8 hv;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co = g * 8 + e;
      float s = bias[co];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) s = fmaf(w[co * TAPS + t], v[t], s);
      if (RELU) s = fmaxf(s, 0.f);
      if (bn_scale) s = fmaxf(fmaf(s, bn_scale[co], bn_shift[co]), 0.f);
      hv[e] = (_Float16)s;
    }
    op[(size_t)g * plane] = hv;
  }
}

// Depthwise 3x3 (+ bias, ReLU) on C8: one thread = one pixel x 8 channels, nine 16-byte loads, fp32 accumulation on
// fp16-rounded weights, one 16-byte store.  The group is blockIdx.z, so its 72 weights and 8 biases are scalar loads.
template <bool RELU>
__global__ __launch_bounds__(256) void dwconv3x3_f16_kernel(const _Float16 *__restrict__ in, _Float16 *__restrict__ out,
                                                             const float *__restrict__ w,  // [C][9], fp16-rounded values
                                                             const float *__restrict__ bias, int G, int H, int W, int hp, int wp) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int pg = blockIdx.z;   // image * G + group
  const int g = pg % G;
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)hp * wp;
  const half8 *ip = reinterpret_cast<const half8 *>(in) + (size_t)pg * plane + (size_t)(y + PADY - 1) * wp + (x + PADX - 1);
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = bias[g * 8 + e];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const half8 v = ip[(size_t)ky * wp + kx];
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = fmaf(w[(g * 8 + e) * 9 + ky * 3 + kx], (float)v[e], s[e]);
    }
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (_Float16)(RELU ? fmaxf(s[e], 0.f) : s[e]);
  reinterpret_cast<half8 *>(out)[(size_t)pg * plane + (size_t)(y + PADY) * wp + (x + PADX)] = r;
}

// Stand-alone 2x2/2 max-pool on C8 (squeeze graph: pool after a Concat).  One thread = one output pixel of one group.
template <int UNUSED = 0>   // (a template so that every translation unit may include this header)
__global__ __launch_bounds__(256) void maxpool2_f16_kernel(const _Float16 *__restrict__ in, _Float16 *__restrict__ out, int OH, int OW,
                                                            int in_hp, int in_wp, int out_hp, int out_wp) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int g = blockIdx.z;   // image * groups + group
  if (x >= OW || y >= OH) return;
  const half8 *ip = reinterpret_cast<const half8 *>(in) + (size_t)g * in_hp * in_wp + (size_t)(2 * y + PADY) * in_wp + (2 * x + PADX);
  const half8 a0 = ip[0], a1 = ip[1], b0 = ip[in_wp], b1 = ip[in_wp + 1];
  half8 r;
#pragma unroll
  for (int e = 0; e < 8; ++e) r[e] = (_Float16)fmaxf(fmaxf((float)a0[e], (float)a1[e]), fmaxf((float)b0[e], (float)b1[e]));
  reinterpret_cast<half8 *>(out)[(size_t)g * out_hp * out_wp + (size_t)(y + PADY) * out_wp + (x + PADX)] = r;
}

// C8 fp16 -> dense NCHW fp32 (spvo_debug_tensor)
template <int UNUSED = 0>   // (a template so that every translation unit may include this header)
__global__ void unpad_c8_kernel(const _Float16 *__restrict__ in, float *__restrict__ out, int C, int H, int W, int hp, int wp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int c = blockIdx.z;   // image * C + channel
  if (x >= W) return;
  const int img = c / C, ch = c % C;
  out[((size_t)c * H + y) * W + x] = (float)in[((((size_t)img * (C / 8) + ch / 8) * hp + (y + PADY)) * wp + (x + PADX)) * 8 + (ch & 7)];
}

}  // namespace spvo
