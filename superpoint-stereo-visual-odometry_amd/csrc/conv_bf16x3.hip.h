// conv_bf16x3.hip.h -- FP32 engines in "split" mode (opt-in, spvo_set_fp32_split): fp32-equivalent
// convolutions on the bf16 matrix pipe.
//
// Every fp32 value v is stored as three bf16 pieces v = p0 + p1 + p2 (p0 = bf16(v), p1 = bf16(v - p0),
// p2 = v - p0 - p1: 3 x 8 significand bits = the 24 of an fp32, the sum is exact), and a product a*b is
// evaluated as the six partial products a0b0 + a0b1 + a1b0 + a1b1 + a0b2 + a2b0.  Each partial product
// of two bf16 numbers is exact in fp32 and accumulates in fp32 inside the matrix instruction; the
// three dropped terms (a1b2, a2b1, a2b2) are below 2^-24 |a||b|, the size of an fp32 rounding error.
// v_mfma_f32_32x32x16_bf16 runs at 16x the rate of v_mfma_f32_32x32x2_f32, so six of them per
// product are 2.7x faster than the native fp32 matrix instruction.  The results agree with the native
// fp32 engine to fp32 rounding level (tests/test_gpu_network.py::test_fp32_split_*): this is a different
// evaluation order of the same fp32 convolution, not a reduced-precision engine.
//
// The reference's counterpart is still the FP32 TensorRT engine (feature_detection_neural_network.cpp:44-49,
// enqueue at :169) -- which on the Ampere-class GPUs it was written for multiplies in TF32 (10 mantissa
// bits) unless that is switched off.
//
// Layout "C8x3": act[img][C/8][3][Hp][Wp][8] of bf16 -- the C8 layout of conv_f16.hip.h with the three
// pieces of a channel group as three consecutive planes.  A k-step of the matrix instruction (K = 16)
// covers ONE group of 8 channels: lane half 0 supplies one piece and lane half 1 another, so that
//     [a0 a0] x [b0 b1] = a0b0 + a0b1,   [a1 a1] x [b0 b1] = a1b0 + a1b1,   [a0 a2] x [b2 b0] = a0b2 + a2b0
// are the six partial products in three instructions.  Everything else (persistent tiles, LDS ring fed by
// global_load_lds, D[co][x] mapping, fused ReLU / 2x2 max-pool, bias from the slab) is as in conv_f16.hip.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string.h>
#include <vector>
#include "conv_f16.hip.h"

namespace spvo {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned short u16x4 __attribute__((ext_vector_type(4)));
typedef unsigned short u16x8 __attribute__((ext_vector_type(8)));

// bf16 bits of the nearest-even rounding of v (finite values)
__host__ __device__ inline unsigned short bf16_bits_rne(float v) {
  unsigned u;
  memcpy(&u, &v, 4);
  u += 0x7FFFu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
__host__ __device__ inline float bf16_bits_to_float(unsigned short h) {
  const unsigned u = (unsigned)h << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}
// v = p[0] + p[1] + p[2] exactly (|v| well inside the fp32 range)
__host__ __device__ inline void split3(float v, unsigned short p[3]) {
  p[0] = bf16_bits_rne(v);
  const float r1 = v - bf16_bits_to_float(p[0]);
  p[1] = bf16_bits_rne(r1);
  const float r2 = r1 - bf16_bits_to_float(p[1]);
  p[2] = bf16_bits_rne(r2);
}

// Device side: two values at a time on v_cvt_pk_bf16_f32 (round to nearest even, as bf16_bits_rne) and v_pk_add_f32:
// P[q] = piece q of x in the low half, of y in the high half.
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void split3_pair(float x, float y, unsigned P[3]) {
  f32x2 v = {x, y};
#pragma unroll
  for (int q = 0; q < 3; ++q) {
    const unsigned u = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
    P[q] = u;
    if (q < 2) {
      const f32x2 h = {__uint_as_float(u << 16), __uint_as_float(u & 0xFFFF0000u)};
      v = v - h;   // exact
    }
  }
}

struct ConvArgsS3 {
  const unsigned short *in;     // C8x3 tensor, image 0, group 0, piece 0
  void *out;                    // C8x3 tensor or fp32 padded planes (OUT_F32)
  const unsigned short *wpack;  // pack_conv_weights_s3()
  int H, W;
  int in_hp, in_wp, in_gtot, in_goff;      // input geometry; channel groups of the tensor / first group read
  int out_hp, out_wp, out_ctot, out_coff;  // output geometry; channels of the tensor / first channel written
  int cout, n_chunks, tiles_x, tiles_y, co_tiles, batch;
};

template <int KS, int CKG_, int WR, int WC>
struct ConvTileS3 {
  static constexpr int CKG = CKG_;                          // channel groups (of 8) per chunk: 1 for 3x3, 2 for 1x1
  static constexpr int TH = 4 * WR, TW = 32 * WC, HALO = KS / 2;
  static constexpr int LW = TW + 2 * HALO, LH = TH + 2 * HALO;
  static constexpr int IN_P = CKG * 3 * LH * LW;            // 16-byte pieces: one pixel of one plane
  static constexpr int W_P = KS * KS * CKG * 3 * CO_TILE;   // one (tap, group, piece, co) row of 8 bf16
  static constexpr int BIAS_P = CO_TILE / 4;
  static constexpr int BUF_P = IN_P + W_P + BIAS_P;
  static constexpr int BUF_PAD = (BUF_P + 255) / 256 * 256;   // every thread stages the same number of pieces (the tail lands in padding)
  // Ring depth: two chunks in flight wherever three buffers fit into the 160 KiB of LDS.  Measured: neutral for the 3x3
  // layers (a chunk of 9 taps is long enough to cover the LDS-DMA latency), 25 % faster for the 1x1 heads (2 steps per chunk).
  static constexpr int NBUF = 3 * BUF_PAD * 16 <= 160 * 1024 ? 3 : 2;
  static constexpr int LDS_BYTES = NBUF * BUF_PAD * 16;
  static constexpr int NSTEP = KS * KS * CKG;
};

// OIHW fp32 weights + bias -> slabs [co_tile][chunk][tap][group][piece][co 64][8] of bf16 bits + 64 fp32 biases (chunk 0)
inline std::vector<unsigned short> pack_conv_weights_s3(const float *w, const float *bias, int cout, int cin, int ks, int ckg) {
  const int co_tiles = (cout + CO_TILE - 1) / CO_TILE, ck = 8 * ckg, nch = cin / ck, taps = ks * ks;
  const size_t slab = ((size_t)taps * ckg * 3 * CO_TILE + CO_TILE / 4) * 8;   // in 16-bit units
  std::vector<unsigned short> out((size_t)co_tiles * nch * slab, 0);
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int ch = 0; ch < nch; ++ch) {
      unsigned short *s = out.data() + ((size_t)ct * nch + ch) * slab;
      for (int o = 0; o < CO_TILE; ++o) {
        const int co = ct * CO_TILE + o;
        if (co >= cout) continue;
        for (int t = 0; t < taps; ++t)
          for (int g = 0; g < ckg; ++g)
            for (int e = 0; e < 8; ++e) {
              unsigned short p[3];
              split3(w[((size_t)co * cin + ch * ck + g * 8 + e) * taps + t], p);
              for (int q = 0; q < 3; ++q) s[((((size_t)t * ckg + g) * 3 + q) * CO_TILE + o) * 8 + e] = p[q];
            }
        if (ch == 0) reinterpret_cast<float *>(s + (size_t)taps * ckg * 3 * CO_TILE * 8)[o] = bias[co];
      }
    }
  return out;
}

template <int KS, int CKG_, int WR, int WC, bool POOL, bool RELU, bool OUT_F32>
__global__ __launch_bounds__(256) void conv_s3_kernel(const ConvArgsS3 a) {
  using T = ConvTileS3<KS, CKG_, WR, WC>;
  constexpr int NT = WR * WC, CKG = T::CKG, LW = T::LW, LH = T::LH, NSTEP = T::NSTEP;
  constexpr int TOT_P = T::BUF_P, NBUF = T::NBUF, BUF_BYTES = T::BUF_PAD * 16;
  constexpr int NIT = T::BUF_PAD / 256;
  static_assert(!POOL || WR == 2, "fused pooling needs both rows of a 2x2 window in one wave");
  static_assert(!(POOL && OUT_F32), "the fp32-output layers are the unpooled heads");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem_s3[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t in_plane = (size_t)a.in_hp * a.in_wp;
  const size_t out_plane = (size_t)a.out_hp * a.out_wp;
  const int n_tiles = a.tiles_x * a.tiles_y * a.co_tiles * a.batch;

  struct TileRef { const unsigned short *in_base, *w_base; int x0, y0, ct, img; };
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
    t.in_base = a.in + (((size_t)t.img * a.in_gtot + a.in_goff) * 3 * in_plane + (size_t)(t.y0 + PADY - T::HALO) * a.in_wp + (t.x0 + PADX - T::HALO)) * 8;
    t.w_base = a.wpack + (size_t)t.ct * a.n_chunks * (T::W_P + T::BIAS_P) * 8;
    return t;
  };

  int piece_off[NIT];   // in 16-bit units, relative to the chunk's input / weight base
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const int idx = it * 256 + tid;
    if (idx < T::IN_P) {
      const int pl = idx / (LH * LW);         // plane of the chunk: group * 3 + piece
      const int rem = idx - pl * (LH * LW);
      const int r = rem / LW;
      const int q = rem - r * LW;
      piece_off[it] = (pl * (int)in_plane + r * a.in_wp + q) * 8;
    } else {
      piece_off[it] = (min(idx, TOT_P - 1) - T::IN_P) * 8;
    }
  }
  auto issue = [&](const TileRef &t, int chunk, unsigned char *buf) {
    const unsigned short *inb = t.in_base + (size_t)chunk * CKG * 3 * in_plane * 8;
    const unsigned short *wb = t.w_base + (size_t)chunk * (T::W_P + T::BIAS_P) * 8;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const int idx = it * 256 + tid;
      const unsigned short *src = ((idx < T::IN_P) ? inb : wb) + piece_off[it];
      glds16(reinterpret_cast<const float *>(src), reinterpret_cast<float *>(buf + (size_t)(it * 256 + wave * 64) * 16));
    }
  };

  // Per-lane LDS piece indices at (tap 0, group 0).  Operand "01": lane half 0 reads piece 0, half 1 piece 1;
  // operand "20": half 0 piece 2, half 1 piece 0.  A operands: "00", "11", "02".
  const int b_row = (wave * WR) * LW + j;
  const int b01 = (half ? 1 : 0) * LH * LW + b_row;
  const int b20 = (half ? 0 : 2) * LH * LW + b_row;
  const int a00 = T::IN_P + j;
  const int a11 = T::IN_P + CO_TILE + j;
  const int a02 = T::IN_P + (half ? 2 : 0) * CO_TILE + j;

  int tile_id = blockIdx.x;
  if (tile_id >= n_tiles) return;
  TileRef cur = decode(tile_id);
  // The staging sequence is linear over (tile, chunk) of this workgroup; `pf` is the next item to fetch, NBUF - 1 items
  // ahead of the one being multiplied.  n_chunks >= NBUF - 1 (host side).
  TileRef pf = cur;
  int pf_chunk = 0, pf_id = tile_id, pf_buf = 0;
  auto fetch_next = [&]() {
    if (pf_id < n_tiles) {
      issue(pf, pf_chunk, smem_s3 + (size_t)pf_buf * BUF_BYTES);
      pf_buf = pf_buf + 1 == NBUF ? 0 : pf_buf + 1;
      if (++pf_chunk == a.n_chunks) {
        pf_chunk = 0;
        pf_id += gridDim.x;
        if (pf_id < n_tiles) pf = decode(pf_id);
      }
    }
  };
#pragma unroll
  for (int i = 0; i < NBUF - 1; ++i) fetch_next();
  int rbuf = 0;                 // LDS buffer of the chunk being multiplied
  bool first_landed = false;
  constexpr unsigned OOB = 0xFFFFFFFFu;
  // wait until the oldest chunk in flight has landed: everything but the youngest (NBUF - 2) * NIT loads of this wave
  auto wait_oldest = [&](bool younger_in_flight) {
    if (NBUF == 3 && younger_in_flight) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NBUF == 3 ? NIT : 0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };

  for (; tile_id < n_tiles; tile_id += gridDim.x) {
    const int next_id = tile_id + gridDim.x;
    TileRef nxt = cur;
    if (next_id < n_tiles) nxt = decode(next_id);

    if (!first_landed) wait_oldest(a.n_chunks > 1 || next_id < n_tiles);
    asm volatile("s_barrier" ::: "memory");

    f32x16 acc[2][NT];
    {
      const f32x4 *bp = reinterpret_cast<const f32x4 *>(smem_s3 + (size_t)rbuf * BUF_BYTES + (size_t)(T::IN_P + T::W_P) * 16);
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

    for (int c = 0; c < a.n_chunks; ++c) {
      if (c > 0) {
        wait_oldest(c + 1 < a.n_chunks || next_id < n_tiles);
        asm volatile("s_barrier" ::: "memory");
      }
      const bf16x8 *buf = reinterpret_cast<const bf16x8 *>(smem_s3 + (size_t)rbuf * BUF_BYTES);
      rbuf = rbuf + 1 == NBUF ? 0 : rbuf + 1;
      bf16x8 av[2][3][2], bv[2][2][NT];   // [register set][operand kind][m or n]
      // Operand reads of a step, in the order the matrix instructions first use them:
      //   A[0][0], B[0][0..NT-1], A[0][1], A[1][0], A[1][1], A[2][0], B[1][0..NT-1], A[2][1]
      constexpr int NLOAD = 6 + 2 * NT;
      auto load_one = [&](int st, int slot, int k) {
        const int t = st / CKG, s = st % CKG;
        const int ky = t / KS, kx = t % KS;
        const int wrow = (t * CKG + s) * 3 * CO_TILE;
        auto ld_a = [&](int kind, int m) { av[slot][kind][m] = buf[(kind == 0 ? a00 : kind == 1 ? a11 : a02) + wrow + 32 * m]; };
        auto ld_b = [&](int kind, int n) {
          const int rr = n / WC, cc = n % WC;
          bv[slot][kind][n] = buf[(kind == 0 ? b01 : b20) + (s * 3 * LH + rr + ky) * LW + cc * 32 + kx];
        };
        if (k == 0) ld_a(0, 0);
        else if (k <= NT) ld_b(0, k - 1);
        else if (k == NT + 1) ld_a(0, 1);
        else if (k == NT + 2) ld_a(1, 0);
        else if (k == NT + 3) ld_a(1, 1);
        else if (k == NT + 4) ld_a(2, 0);
        else if (k <= 2 * NT + 4) ld_b(1, k - NT - 5);
        else ld_a(2, 1);
      };
#pragma unroll
      for (int k = 0; k < NLOAD; ++k) load_one(0, 0, k);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int cs = st & 1;
        // 6*NT matrix instructions: [a0 a0] x [b0 b1], [a1 a1] x [b0 b1], [a0 a2] x [b2 b0]; an accumulator is revisited
        // after 2*NT others.  The reads of the next step are spread one per matrix instruction: a ds_read_b128 of 4 waves
        // keeps the LDS busy for about as long as one matrix instruction runs, so issued in one burst they fill the LDS
        // queue and stall the matrix instructions behind them.
#pragma unroll
        for (int i = 0; i < 6 * NT; ++i) {
          const int kind = i / (2 * NT), m = (i / NT) % 2, n = i % NT;
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[cs][kind][m], bv[cs][kind == 2 ? 1 : 0][n], acc[m][n], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
          if (i == 0 && st == 0) fetch_next();   // into the buffer every wave finished reading before the barrier above
          if (st + 1 < NSTEP) {
            if (i < NLOAD) load_one(st + 1, cs ^ 1, i);
            if (i == 6 * NT - 1)   // NT = 1: more reads than matrix instructions
              for (int k = 6 * NT; k < NLOAD; ++k) load_one(st + 1, cs ^ 1, k);
          }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    // the next tile's first chunk has landed before this tile's stores queue up behind it (same in-order counter)
    wait_oldest(next_id < n_tiles && a.n_chunks > 1);
    first_landed = true;

    // ------------------------------------------------------------------ epilogue
    auto relu = [](float v) { return RELU ? __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()) : v; };
    if constexpr (OUT_F32) {
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
      // C8x3: registers 4g .. 4g+3 of a lane are 4 consecutive channels of group 4m + g -> one 8-byte store per piece
      unsigned short *g_base = reinterpret_cast<unsigned short *>(a.out) + ((size_t)cur.img * (a.out_ctot / 8) + a.out_coff / 8 + (size_t)cur.ct * (CO_TILE / 8)) * 3 * out_plane * 8;
      const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(g_base, 0, 0x7FFFFFFF, 0x00020000);
      const int groups_valid = (a.cout - cur.ct * CO_TILE + 7) / 8;
      auto store_tile = [&](const float (&v)[16], int m, unsigned voff) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          if (4 * m + g < groups_valid) {
            unsigned lo[3], hi[3];
            split3_pair(v[4 * g + 0], v[4 * g + 1], lo);
            split3_pair(v[4 * g + 2], v[4 * g + 3], hi);
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
            for (int q = 0; q < 3; ++q) {
              const u32x2 pk = {lo[q], hi[q]};
              __builtin_amdgcn_raw_buffer_store_b64(pk, rsrc, voff, ((4 * m + g) * 3 + q) * (int)out_plane * 16, 0);
            }
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
              for (int r = 0; r < 16; ++r) v[r] = relu(acc[m][rr * WC + cc][r]);
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
              float q = fmaxf(acc[m][0 * WC + cc][r], acc[m][1 * WC + cc][r]);   // ReLU commutes with max
              v[r] = relu(fmaxf(q, __shfl_xor(q, 1)));
            }
            store_tile(v, m, voff);
          }
      }
    }
    cur = nxt;
  }
}

// Cin = 1 layers (the network input is an fp32 plane): fp32 vector arithmetic exactly as conv_first_kernel, C8x3 out.
template <int KS, bool RELU>
__global__ __launch_bounds__(256) void conv_first_s3_kernel(const float *__restrict__ in, unsigned short *__restrict__ out,
                                                             const float *__restrict__ w, const float *__restrict__ bias, int H, int W, int hp,
                                                             int wp, int out_gtot, int out_goff, int cout) {
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
  u16x8 *op = reinterpret_cast<u16x8 *>(out) + ((size_t)img * out_gtot + out_goff) * 3 * plane + (size_t)(y + PADY) * wp + (x + PADX);
  for (int g = 0; g < cout / 8; ++g) {
    float sv[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const int co = g * 8 + e;
      float s = bias[co];
#pragma unroll
      for (int t = 0; t < TAPS; ++t) s = fmaf(w[co * TAPS + t], v[t], s);
      sv[e] = RELU ? fmaxf(s, 0.f) : s;
    }
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    u32x4 pc[3];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsigned P[3];
      split3_pair(sv[2 * e], sv[2 * e + 1], P);
      pc[0][e] = P[0]; pc[1][e] = P[1]; pc[2][e] = P[2];
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) op[((size_t)g * 3 + q) * plane] = __builtin_bit_cast(u16x8, pc[q]);
  }
}

// C8x3 -> dense NCHW fp32 (spvo_debug_tensor): the exact sum of the three pieces
template <int UNUSED = 0>   // (a template so that every translation unit may include this header)
__global__ void unpad_s3_kernel(const unsigned short *__restrict__ in, float *__restrict__ out, int C, int H, int W, int hp, int wp) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  const int y = blockIdx.y;
  const int c = blockIdx.z;   // image * C + channel
  if (x >= W) return;
  const int img = c / C, ch = c % C;
  const size_t plane = (size_t)hp * wp;
  const unsigned short *p = in + ((((size_t)img * (C / 8) + ch / 8) * 3 * plane) + (size_t)(y + PADY) * wp + (x + PADX)) * 8 + (ch & 7);
  out[((size_t)c * H + y) * W + x] = (bf16_bits_to_float(p[0]) + bf16_bits_to_float(p[plane * 8])) + bf16_bits_to_float(p[2 * plane * 8]);
}

}  // namespace spvo
