// conv_i8.hip.h -- INT8 engines (BASELINE config 5: MobileNet-backbone SuperPoint in int8):
// convolutions on v_mfma_i32_32x32x32_i8 with exact int32 accumulation.
//
// The reference knows FP32 and FP16 engines only (feature_detection.hpp:124-126); config 5 is a build-side
// extension, so the arithmetic is DEFINED by oracle/net_int8.py and reproduced here bit for bit: integer
// accumulation is exact and every float operation of the requantisation is a separately rounded fp32 multiply or
// add in the oracle's order.  Round 6: every affine of the requantisation is ONE fused multiply-add (__builtin_fmaf = v_fma_f32, correctly
// rounded, as oracle/net_int8.py's fma32), and the loader folds 1 / s_out into the chain's last affine where the oracle does (inv_s_out = 1 then).
//
// Activation layout "C16": act[img][C/16][Hp][Wp][16] of int8 -- 16 channels = the 16 bytes of one pixel, padded
// group planes with the usual zero border (q = 0 is the real value 0: the quantisation is symmetric).  As in
// conv_f16.hip.h that is the operand shape of the matrix instruction (lane = one pixel x 16 consecutive k), so an
// operand is one ds_read_b128.  Tiling, LDS ring, persistence and the k-step order are those of conv_f16.hip.h;
// a k-step is one filter tap x 32 input channels.
//
// Epilogue (per output value, in this order):  r = f32(acc) * m[co];  r = r + bias[co];  [ReLU]
// [r = r * bn_scale[co]; r = r + bn_shift[co]; ReLU]  [r = r + f32(res_q) * s_res; ReLU]  [2x2 max]  then either
// q = clip(rint(r * inv_s_out), -127, 127) -> C16, or r itself -> fp32 planes (the fp32 bindings).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>
#include "conv_mfma.hip.h"

namespace spvo {

typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef int i32x16 __attribute__((ext_vector_type(16)));

struct ConvArgs8 {
  const int8_t *in;       // C16 tensor, image 0, group 0
  void *out;              // C16 tensor (int8) or fp32 padded planes (OUT_F32)
  const int8_t *wpack;    // [co_tiles][n_chunks][tap][group][co 64][16]: pack_conv_weights_i8()
  const float *qm;        // [co_tiles*64]  weight scale * input scale
  const float *bias;      // [co_tiles*64]
  const float *bn_scale = nullptr, *bn_shift = nullptr;   // EPI 1
  const int8_t *residual = nullptr;                       // EPI 2: C16 tensor of the input's geometry, cout channels, image 0
  float s_res = 0.f;      // EPI 2: scale of the residual tensor
  float inv_s_out = 0.f;  // 1 / scale of the output tensor (unused with OUT_F32)
  int H, W;
  int in_hp, in_wp, in_gtot, in_goff;     // groups of 16 channels
  int out_hp, out_wp, out_ctot, out_coff; // channels
  int cout, n_chunks, tiles_x, tiles_y, co_tiles, batch;
};

template <int KS, int CKG_, int WR, int WC>
struct ConvTile8 {
  static constexpr int CKG = CKG_;                      // channel groups (of 16) per chunk: 2 for 3x3; 4 (or 2) for 1x1
  static constexpr int TH = 4 * WR, TW = 32 * WC, HALO = KS / 2;
  static constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
  static constexpr int IN_P = CKG * LH * LW;            // 16-byte pieces: one pixel of one group
  static constexpr int W_P = KS * KS * CKG * CO_TILE;   // one (tap, group, co) row of 16 input channels
  static constexpr int BUF_P = IN_P + W_P;
  static constexpr int LDS_BYTES = 2 * BUF_P * 16;
  static constexpr int NSTEP = KS * KS * (CKG / 2);
};

// Host side.  Per-output-channel symmetric quantisation exactly as oracle/net_int8.py: ws = max|w[co]| / 127 (fp32
// division; 1 for an all-zero channel), w_q = clip(rint(w / ws), -127, 127) with round-half-even.
inline void quantize_conv_weights(const float *w, int cout, int per_co, std::vector<int8_t> &wq, std::vector<float> &ws) {
  wq.assign((size_t)cout * per_co, 0);
  ws.assign(cout, 1.f);
  for (int co = 0; co < cout; ++co) {
    float amax = 0.f;
    for (int i = 0; i < per_co; ++i) amax = std::max(amax, std::fabs(w[(size_t)co * per_co + i]));
    const float s = amax == 0.f ? 1.f : amax / 127.f;
    ws[co] = s;
    for (int i = 0; i < per_co; ++i) {
      const float q = std::nearbyintf(w[(size_t)co * per_co + i] / s);
      wq[(size_t)co * per_co + i] = (int8_t)std::min(127.f, std::max(-127.f, q));
    }
  }
}

// quantised OIHW weights -> slabs [co_tile][chunk][tap][group][co 64][16]
inline std::vector<int8_t> pack_conv_weights_i8(const int8_t *wq, int cout, int cin, int ks, int ckg) {
  const int co_tiles = (cout + CO_TILE - 1) / CO_TILE, ck = 16 * ckg, nch = cin / ck, taps = ks * ks;
  const size_t slab = (size_t)taps * ckg * CO_TILE * 16;
  std::vector<int8_t> out((size_t)co_tiles * nch * slab, 0);
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int ch = 0; ch < nch; ++ch) {
      int8_t *s = out.data() + ((size_t)ct * nch + ch) * slab;
      for (int o = 0; o < CO_TILE; ++o) {
        const int co = ct * CO_TILE + o;
        if (co >= cout) continue;
        for (int t = 0; t < taps; ++t)
          for (int g = 0; g < ckg; ++g)
            for (int e = 0; e < 16; ++e)
              s[(((size_t)t * ckg + g) * CO_TILE + o) * 16 + e] = wq[((size_t)co * cin + ch * ck + g * 16 + e) * taps + t];
      }
    }
  return out;
}

__device__ __forceinline__ int quantize_i8(float r, float inv_s) {
  return (int)fminf(127.f, fmaxf(-127.f, rintf(mul_rn(r, inv_s))));
}

template <int KS, int CKG_, int WR, int WC, bool POOL, bool RELU, bool OUT_F32, int EPI = 0>
__global__ __launch_bounds__(256) void conv_i8_kernel(const ConvArgs8 a) {
  using T = ConvTile8<KS, CKG_, WR, WC>;
  constexpr int NT = WR * WC, CKG = T::CKG, LW = T::LW, LH = T::LH, NSTEP = T::NSTEP;
  constexpr int TOT_P = T::BUF_P;
  constexpr int NIT = (TOT_P + 255) / 256;
  static_assert(!POOL || WR == 2, "fused pooling needs both rows of a 2x2 window in one wave");
  static_assert(!(POOL && OUT_F32), "the fp32-output layers are the unpooled heads");
  static_assert(EPI == 0 || !OUT_F32, "BatchNorm / residual epilogues write C16");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem8[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t in_plane = (size_t)a.in_hp * a.in_wp;
  const size_t out_plane = (size_t)a.out_hp * a.out_wp;
  const int n_tiles = a.tiles_x * a.tiles_y * a.co_tiles * a.batch;

  struct TileRef { const int8_t *in_base, *w_base; int x0, y0, ct, img; };
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
    t.in_base = a.in + (((size_t)t.img * a.in_gtot + a.in_goff) * in_plane + (size_t)(t.y0 + PADY - T::HALO) * a.in_wp + (t.x0 + PADX - T::HALO)) * 16;
    t.w_base = a.wpack + (size_t)t.ct * a.n_chunks * T::W_P * 16;
    return t;
  };

  int piece_off[NIT];   // in bytes, relative to the chunk's input / weight base
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * 256 + tid;
    if (idx < T::IN_P) {
      const int g = idx / (LH * LW);
      const int rem = idx - g * (LH * LW);
      const int r = rem / LW;
      const int q = rem - r * LW;
      piece_off[it] = (g * (int)in_plane + r * a.in_wp + q) * 16;
    } else {
      piece_off[it] = (min(idx, TOT_P - 1) - T::IN_P) * 16;
    }
  }
  auto issue = [&](const TileRef &t, int chunk, unsigned char *buf) {
    const int8_t *inb = t.in_base + (size_t)chunk * CKG * in_plane * 16;
    const int8_t *wb = t.w_base + (size_t)chunk * T::W_P * 16;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 256 + tid;
      const int8_t *src = ((idx < T::IN_P) ? inb : wb) + piece_off[it];
      if (it < NIT - 1 || idx < TOT_P)
        glds16(reinterpret_cast<const float *>(src), reinterpret_cast<float *>(buf + (size_t)(it * 256 + wave * 64) * 16));
    }
  };

  const int b_lane = (half * LH + wave * WR) * LW + j;
  const int a_lane = T::IN_P + half * CO_TILE + j;

  int tile_id = blockIdx.x;
  if (tile_id >= n_tiles) return;
  TileRef cur = decode(tile_id);
  issue(cur, 0, smem8);
  int ring = 0;
  bool first_landed = false;
  constexpr unsigned OOB = 0xFFFFFFFFu;

  for (; tile_id < n_tiles; tile_id += gridDim.x) {
    const int next_id = tile_id + gridDim.x;
    TileRef nxt = cur;
    if (next_id < n_tiles) nxt = decode(next_id);

    i32x16 acc[2][NT];
#pragma unroll
    for (int m = 0; m < 2; ++m)
#pragma unroll
      for (int n = 0; n < NT; ++n)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[m][n][r] = 0;

    for (int c = 0; c < a.n_chunks; ++c, ++ring) {
      // the wait for the first chunk of a tile happened in front of the previous tile's epilogue (see conv_mfma.hip.h)
      if (c > 0 || !first_landed) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      else asm volatile("s_barrier" ::: "memory");
      unsigned char *nbuf = smem8 + (size_t)((ring + 1) & 1) * TOT_P * 16;
      const i32x4 *buf = reinterpret_cast<const i32x4 *>(smem8 + (size_t)(ring & 1) * TOT_P * 16);
      i32x4 av[2][2], bv[2][NT];
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
        acc[0][0] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[cs][0], bv[cs][0], acc[0][0], 0, 0, 0);
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
            if (m + n > 0) acc[m][n] = __builtin_amdgcn_mfma_i32_32x32x32_i8(av[cs][m], bv[cs][n], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first_landed = true;

    // ------------------------------------------------------------------ epilogue
    // register r of accumulator row block m = channel 32m + (r&3) + 8(r>>2) + 4*half of the co tile
    const int co_t = cur.ct * CO_TILE;
    const size_t res_plane = (size_t)a.in_hp * a.in_wp;
    auto tail = [&](int accv, int m, int r, int y, int x) -> float {   // requantisation chain up to (not including) pooling
      const int co = co_t + 32 * m + (r & 3) + 8 * (r >> 2) + 4 * half;
      float v = __builtin_fmaf((float)accv, a.qm[co], a.bias[co]);   // one fused multiply-add per affine (oracle/net_int8.py)
      if (RELU) v = fmaxf(v, 0.f);
      if constexpr (EPI == 1) v = fmaxf(__builtin_fmaf(v, a.bn_scale[co], a.bn_shift[co]), 0.f);
      if constexpr (EPI == 2) {
        const int rq = co < a.cout ? (int)a.residual[((((size_t)cur.img * (a.cout / 16) + co / 16) * res_plane + (size_t)(y + PADY) * a.in_wp + (x + PADX)) * 16) + (co & 15)] : 0;
        v = fmaxf(__builtin_fmaf((float)rq, a.s_res, v), 0.f);
      }
      return v;
    };
    if constexpr (OUT_F32) {
      float *co_base = reinterpret_cast<float *>(a.out) + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)co_t) * out_plane;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
      const int oplane = (int)out_plane;
      const int kmax = a.cout - (co_t + 4 * half);
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
              __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(tail(acc[m][rr * WC + cc][r], m, r, y, x)), rsrc, k < kmax ? voff : OOB, k * oplane * 4, 0);
            }
          }
    } else {
      // C16: registers 4g .. 4g+3 of a lane are 4 consecutive channels -> one 4-byte store at byte 8(g&1) + 4*half of
      // the pixel's 16 bytes in group 2m + (g>>1) of the co tile
      int8_t *g_base = reinterpret_cast<int8_t *>(a.out) + ((size_t)cur.img * (a.out_ctot / 16) + a.out_coff / 16 + (size_t)cur.ct * (CO_TILE / 16)) * out_plane * 16;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(g_base, 0, 0x7FFFFFFF, 0x00020000);
      const int groups_valid = (a.cout - co_t + 15) / 16;
      auto store_tile = [&](const float (&v)[16], int m, unsigned voff) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (2 * m + (g >> 1) < groups_valid) {
            unsigned pk = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) pk |= ((unsigned)quantize_i8(v[4 * g + e], a.inv_s_out) & 0xFFu) << (8 * e);
            __builtin_amdgcn_raw_buffer_store_b32(pk, rsrc, voff == OOB ? OOB : voff + 8u * (g & 1), (2 * m + (g >> 1)) * (int)out_plane * 16, 0);
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
              const unsigned voff = ((y < a.H) && (x < a.W)) ? (unsigned)(((y + PADY) * a.out_wp + (x + PADX)) * 16 + 4 * half) : OOB;
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
            const int yi = cur.y0 + wave * 2, xi = cur.x0 + cc * 32 + j;
            const int y = (cur.y0 >> 1) + wave, x = xi >> 1;
            const unsigned voff = ((y < OH) && (x < OW) && !(j & 1)) ? (unsigned)(((y + PADY) * a.out_wp + (x + PADX)) * 16 + 4 * half) : OOB;
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
              float q = fmaxf(tail(acc[m][0 * WC + cc][r], m, r, yi, xi), tail(acc[m][1 * WC + cc][r], m, r, yi + 1, xi));
              v[r] = fmaxf(q, __shfl_xor(q, 1));
            }
            store_tile(v, m, voff);
          }
      }
    }
    cur = nxt;
  }
}

// fp32 stem of an INT8 engine (Cin = 1): bias first, then one separately rounded multiply and add per tap in
// row-major order -- the order oracle/net_int8.py uses -- optional BatchNorm + ReLU; the result is stored as an
// fp32 plane (OUT_Q = false: a stem tensor with fewer than 16 channels) or quantised to C16 (OUT_Q = true).
template <int KS, bool RELU, bool OUT_Q>
__global__ __launch_bounds__(256) void conv_first_i8_kernel(const float *__restrict__ in, void *__restrict__ out,
                                                             const float *__restrict__ w,  // [cout][KS*KS] fp32
                                                             const float *__restrict__ bias, const float *__restrict__ bn_scale,
                                                             const float *__restrict__ bn_shift, float inv_s_out, int H, int W, int hp, int wp,
                                                             int out_ctot, int out_coff, int cout) {
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
  auto eval = [&](int co) {
    float s = bias[co];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s = __builtin_fmaf(w[co * TAPS + t], v[t], s);
    if (RELU) s = fmaxf(s, 0.f);
    if (bn_scale) s = fmaxf(__builtin_fmaf(s, bn_scale[co], bn_shift[co]), 0.f);
    return s;
  };
  if constexpr (OUT_Q) {
    i32x4 *op = reinterpret_cast<i32x4 *>(out) + ((size_t)img * (out_ctot / 16) + out_coff / 16) * plane + (size_t)(y + PADY) * wp + (x + PADX);
    for (int g = 0; g < cout / 16; ++g) {
      i32x4 pk;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        unsigned u = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) u |= ((unsigned)quantize_i8(eval(g * 16 + 4 * d + e), inv_s_out) & 0xFFu) << (8 * e);
        pk[d] = (int)u;
      }
      op[(size_t)g * plane] = pk;
    }
  } else {
    float *op = reinterpret_cast<float *>(out) + ((size_t)img * out_ctot + out_coff) * plane + (size_t)(y + PADY) * wp + (x + PADX);
    for (int co = 0; co < cout; ++co) op[(size_t)co * plane] = eval(co);
  }
}

// Depthwise 3x3 on C16: one thread = one pixel x 16 channels, nine 16-byte loads, exact int32 accumulation,
// r = f32(acc) * m[c] + bias[c], ReLU, requantise, one 16-byte store.  The group is blockIdx.z: its weights are uniform.
template <bool RELU>
__global__ __launch_bounds__(256) void dwconv3x3_i8_kernel(const int8_t *__restrict__ in, int8_t *__restrict__ out,
                                                            const int *__restrict__ wq,   // [C][9] quantised weights as int32
                                                            const float *__restrict__ qm, const float *__restrict__ bias, float inv_s_out,
                                                            int G, int H, int W, int hp, int wp) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int pg = blockIdx.z;   // image * G + group
  const int g = pg % G;
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)hp * wp;
  const i32x4 *ip = reinterpret_cast<const i32x4 *>(in) + (size_t)pg * plane + (size_t)(y + PADY - 1) * wp + (x + PADX - 1);
  int acc[16];
#pragma unroll
  for (int e = 0; e < 16; ++e) acc[e] = 0;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const i32x4 v = ip[(size_t)ky * wp + kx];
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int xv = (int)(int8_t)((unsigned)v[e >> 2] >> (8 * (e & 3)));
        acc[e] += wq[(g * 16 + e) * 9 + ky * 3 + kx] * xv;
      }
    }
  i32x4 pk;
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    unsigned u = 0;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = g * 16 + 4 * d + e;
      float r = __builtin_fmaf((float)acc[4 * d + e], qm[c], bias[c]);
      if (RELU) r = fmaxf(r, 0.f);
      u |= ((unsigned)quantize_i8(r, inv_s_out) & 0xFFu) << (8 * e);
    }
    pk[d] = (int)u;
  }
  reinterpret_cast<i32x4 *>(out)[(size_t)pg * plane + (size_t)(y + PADY) * wp + (x + PADX)] = pk;
}

// C16 int8 -> dense NCHW fp32 holding the integer values q (spvo_debug_tensor)
template <int UNUSED = 0>   // (a template so that every translation unit may include this header)
__global__ void unpad_c16_kernel(const int8_t *__restrict__ in, float *__restrict__ out, int C, int H, int W, int hp, int wp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int c = blockIdx.z;   // image * C + channel
  if (x >= W) return;
  const int img = c / C, ch = c % C;
  out[((size_t)c * H + y) * W + x] = (float)in[((((size_t)img * (C / 16) + ch / 16) * hp + (y + PADY)) * wp + (x + PADX)) * 16 + (ch & 15)];
}

}  // namespace spvo
