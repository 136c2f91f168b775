// conv_mfma.hip.h -- K2/K3: 3x3 and 1x1 convolution as an implicit GEMM on the
// gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32), bias + ReLU (+ 2x2 max-pool)
// fused in the epilogue.
//
// Replaces the TensorRT engine the reference enqueues at
// feature_detection_neural_network.cpp:169 for every Conv/Relu/MaxPool node of
// the SuperPoint graphs.
//
// Data layout (DESIGN.md "HBM layout"): activations are padded planes
//   act[b][c][Hp][Wp],  pixel (y, x) at [y + PADY][x + PADX],  PADY = 1, PADX = 4,
//   Hp = roundup(H, 8) + 2,  Wp = roundup(W, 64) + 8,
// and everything outside the H x W interior is zero and never written, so the
// halo loads need no bounds checks and every 16-byte load is aligned.
//
// GEMM mapping: D[co][pixel] = sum_k Wt[co][k] * X[k][pixel], k = (tap, ci).
//   A operand (32 rows = output channels): lane l holds Wt[co = l&31][k = l>>5]
//   B operand (32 cols = 32 consecutive x): lane l holds X[k = l>>5][x = l&31]
//   D: lane l holds column x = l&31, rows (reg&3) + 8*(reg>>2) + 4*(l>>5)
// so each accumulator register stores to 32 consecutive pixels of one channel
// (128-byte segments) and both LDS operand reads are bank-conflict-free
// ds_read_b32 (consecutive lanes -> consecutive dwords).
//
// A workgroup is 4 waves; it owns 64 output channels x (4*WR rows) x (32*WC cols).
// Wave w owns rows [w*WR, w*WR+WR) and all WC column tiles: 2 x WR*WC
// accumulator tiles.  The reduction runs over chunks of CK input channels; each
// chunk's input halo tile and weight slab are fetched straight into LDS
// (global_load_lds_dwordx4), double-buffered, one barrier per chunk.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <vector>

namespace spvo {

#ifndef SPVO_PADX
#define SPVO_PADX 4
#endif
constexpr int PADX = SPVO_PADX;   // multiple of 4 (16-byte aligned rows)
constexpr int PADY = 1;
constexpr int CO_TILE = 64;

__host__ __device__ inline int padded_h(int h) { return ((h + 7) / 8) * 8 + 2; }
__host__ __device__ inline int padded_w(int w) { return ((w + 63) / 64) * 64 + 2 * PADX; }

typedef float f32x16 __attribute__((ext_vector_type(16)));

// Separately rounded fp32 operations.  hipcc compiles device code with -ffp-contract=fast and HIP's __fmul_rn /
// __fadd_rn are plain `x * y` / `x + y`, so a multiply feeding an add may become one fused multiply-add -- a
// different rounding.  Results that must be bit-identical to a CPU restatement (match distances, softmax sums, the
// INT8 requantisation) go through these: instructions emitted under `contract(off)` carry no contract flag and are
// never fused, also after inlining.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

struct ConvArgs {
  const float *in;     // padded planes of the input tensor (batch 0, channel 0)
  float *out;          // padded planes of the output tensor
  const float *wpack;  // [co_tiles][n_chunks][KS*KS*CK + 1][64]: pack_conv_weights() below
  const float *bias;   // [cout]: read by the Cin = 1 / depthwise kernels only (the MFMA kernel finds its bias in wpack)
  int H, W;            // conv resolution (input == pre-pool output)
  int in_hp, in_wp, in_ctot, in_coff;
  int out_hp, out_wp, out_ctot, out_coff;
  int cout;            // real output channels of this op
  int n_chunks;        // cin / CK
  int tiles_x, tiles_y, co_tiles;
  int batch;           // images in this launch
  // MobileNet epilogues (EPI template parameter of the kernel), applied per element BEFORE pooling
  const float *bn_scale = nullptr, *bn_shift = nullptr;  // EPI 1: v = relu(v * scale[co] + shift[co]), [co_tiles*64]
  const float *residual = nullptr;                       // EPI 2: v = relu(v + residual[co][y][x]); planes of the
                                                         //        input's geometry (in_hp x in_wp), cout channels
  int *sched = nullptr;                                  // conv_wino2_kernel: {8 tile counters (one per XCD band), workgroups done} of this layer,
                                                         //        all 0 between launches: tiles are handed out dynamically (nullptr: blockIdx.x + k gridDim.x)
  unsigned long long *stamps = nullptr;                  // diagnostic build (ABL 4) only: per workgroup
                                                         //        {shader cycles, 100 MHz ticks} around the tile loop
};

__device__ __forceinline__ void glds16(const float *src, float *lds_wave_base) {
  __builtin_amdgcn_global_load_lds(
      (const __attribute__((address_space(1))) void *)src,
      (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}

template <int KS, int CK, int WR, int WC>
struct ConvTile {
  static constexpr int TH = 4 * WR;
  static constexpr int TW = 32 * WC;
  static constexpr int HALO = KS / 2;
  static constexpr int LW = TW + (KS == 3 ? 8 : 0);  // LDS row: x0-4 .. x0+TW+3
  static constexpr int LH = TH + 2 * HALO;
  static constexpr int IN_FLOATS = CK * LH * LW;
  static constexpr int W_ROWS = KS * KS * CK;            // (tap, ci) rows of 64 output channels ...
  static constexpr int W_FLOATS = (W_ROWS + 1) * CO_TILE;  // ... plus one row that holds the bias in chunk 0's slab
  static constexpr int BUF_FLOATS = IN_FLOATS + W_FLOATS;
  static constexpr int LDS_BYTES = 2 * BUF_FLOATS * 4;
};

// Host side: OIHW weights + bias -> the slabs the kernel stages, [co_tile][chunk][(tap, ci) rows + 1][64].
// The extra row carries the bias of the co tile in chunk 0 (zeros elsewhere): the kernel reads it from
// LDS together with the weights instead of through a separate global load.
inline std::vector<float> pack_conv_weights(const float *w, const float *bias, int cout, int cin, int ks, int ck) {
  const int co_tiles = (cout + CO_TILE - 1) / CO_TILE, nch = cin / ck, taps = ks * ks, rows = taps * ck + 1;
  std::vector<float> out((size_t)co_tiles * nch * rows * CO_TILE, 0.f);
  for (int ct = 0; ct < co_tiles; ++ct)
    for (int ch = 0; ch < nch; ++ch) {
      float *slab = out.data() + ((size_t)ct * nch + ch) * rows * CO_TILE;
      for (int o = 0; o < CO_TILE; ++o) {
        const int co = ct * CO_TILE + o;
        if (co >= cout) continue;
        for (int t = 0; t < taps; ++t)
          for (int c = 0; c < ck; ++c) slab[(t * ck + c) * CO_TILE + o] = w[((size_t)co * cin + ch * ck + c) * taps + t];
        if (ch == 0) slab[(rows - 1) * CO_TILE + o] = bias[co];
      }
    }
  return out;
}

// The kernel is PERSISTENT: the grid is sized to what the chip holds at once (host side) and
// every workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ...  The LDS ring keeps
// running across tile boundaries -- the first chunk of the next tile is fetched while the last
// chunk of the current one is multiplied -- so the fill latency, the bias loads and the output
// stores of a tile hide behind matrix work instead of bracketing it.
// ABL (timing experiments only, results are wrong when != 0): 1 = stage only the first chunk,
// 2 = additionally keep the MFMA operands in registers (no LDS reads in the loop);
// 4 = correct results plus clock stamps around the tile loop (in-kernel clock = cycles / ticks * 100 MHz).
// TAG changes nothing but the kernel's name: the layer with the most FLOPs of a plan (conv1b) runs the TAG = 1
// instance, so that profilers list the dominant kernel on a line of its own instead of averaged with the other
// layers that share its tile variant.
template <int KS, int CK, int WR, int WC, bool POOL, bool RELU, int MINW = 1, int ABL = 0, int EPI = 0, int TAG = 0>
__global__ __launch_bounds__(256, MINW) void conv_mfma_kernel(const ConvArgs a) {
  using T = ConvTile<KS, CK, WR, WC>;
  constexpr int NT = WR * WC;
  constexpr int LW = T::LW, LH = T::LH, LW4 = LW / 4;
  constexpr int IN_V4 = T::IN_FLOATS / 4, W_V4 = T::W_FLOATS / 4, TOT_V4 = IN_V4 + W_V4;
  constexpr int NIT = (TOT_V4 + 255) / 256;
  constexpr int XO = (KS == 3) ? 3 : 0;  // LDS column of output column 0, tap kx = 0
  static_assert(!POOL || WR == 2, "fused pooling needs both rows of a 2x2 window in one wave");

  extern __shared__ __attribute__((aligned(16))) float smem[];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int half = lane >> 5;
  const int j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const size_t in_plane = (size_t)a.in_hp * a.in_wp;
  const size_t out_plane = (size_t)a.out_hp * a.out_wp;
  const int n_tiles = a.tiles_x * a.tiles_y * a.co_tiles * a.batch;

  // tile id -> (x fastest, then y, then co tile, then image)
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
    t.in_base = a.in + ((size_t)t.img * a.in_ctot + a.in_coff) * in_plane + (size_t)(t.y0 + PADY - T::HALO) * a.in_wp +
                (t.x0 + PADX - (KS == 3 ? 4 : 0));
    t.w_base = a.wpack + (size_t)t.ct * a.n_chunks * T::W_FLOATS;
    return t;
  };

  // Per-thread staging plan, computed once: element offset of each of this thread's 16-byte
  // pieces relative to the chunk's input base (pieces < IN_V4) or weight base (the rest).
  // Per chunk the issue loop is then branch-free except for the ragged last piece.
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

  const int b_lane = half * (LH * LW) + (wave * WR) * LW + j + XO;
  const int a_lane = T::IN_FLOATS + half * CO_TILE + j;

  int tile_id = blockIdx.x;
  if (tile_id >= n_tiles) return;
  unsigned long long stamp_c = 0, stamp_r = 0;
  if constexpr (ABL == 4) { stamp_c = __builtin_amdgcn_s_memtime(); stamp_r = __builtin_amdgcn_s_memrealtime(); }
  TileRef cur = decode(tile_id);
  issue(cur, 0, smem);
  int ring = 0;                 // chunks consumed so far: selects the LDS buffer
  bool first_landed = false;

  constexpr int NSTEP = KS * KS * (CK / 2);
  constexpr unsigned OOB = 0xFFFFFFFFu;   // buffer-store offset of a lane that must not write

  for (; tile_id < n_tiles; tile_id += gridDim.x) {
    const int next_id = tile_id + gridDim.x;
    TileRef nxt = cur;
    if (next_id < n_tiles) nxt = decode(next_id);

    // Chunk `ring` has landed for every wave once each wave has drained its own LDS-DMA and passed the
    // barrier; after the barrier buffer (ring+1)&1 is no longer read by anyone.  The wait is explicit
    // because hipcc does not count global_load_lds in the waits it emits.  At the first chunk of a tile
    // (but the very first) the wait already happened in front of the previous tile's epilogue, so that
    // tile's output stores -- younger than the LDS-DMA and counted by the same in-order counter -- stay
    // in flight under this tile's matrix work instead of being drained here.  A bare s_barrier, not
    // __syncthreads(): its release fence would drain them too.
    if (!first_landed) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
    else asm volatile("s_barrier" ::: "memory");
    // The accumulators start as one extra k-step, A = (bias, 0), B = (1, 1), C = 0: 0 + bias * 1 + 0 * 1
    // is exactly the bias, so the chain equals one that starts from the bias.  The bias row sits behind
    // the weights of chunk 0 in LDS.
    f32x16 acc[2][NT];
    {
      const float *buf0 = smem + (ring & 1) * T::BUF_FLOATS;
      float bias_a[2];
#pragma unroll
      for (int m = 0; m < 2; ++m) bias_a[m] = half ? 0.f : buf0[T::IN_FLOATS + T::W_ROWS * CO_TILE + 32 * m + j];
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int n = 0; n < NT; ++n) {
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[m][n][r] = 0.f;
          acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(bias_a[m], 1.0f, acc[m][n], 0, 0, 0);
        }
    }

    for (int c = 0; c < a.n_chunks; ++c, ++ring) {
      if (c > 0) asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
      float *nbuf = smem + ((ring + 1) & 1) * T::BUF_FLOATS;
      const float *buf = smem + (ring & 1) * T::BUF_FLOATS;
      // k-steps of this chunk: step = (tap, channel pair).  Operands of step s+1 are read from LDS
      // into a second register set BEFORE the MFMAs of step s issue, so the LDS latency hides
      // under 2*NT matrix instructions instead of stalling the head of every step.
      float av[2][2], bv[2][NT];
      auto load_step = [&](int st, int slot) {
        const int t = st / (CK / 2), p = st % (CK / 2);
        const int ky = t / KS, kx = t % KS;
#pragma unroll
        for (int m = 0; m < 2; ++m) av[slot][m] = buf[a_lane + (t * CK + 2 * p) * CO_TILE + 32 * m];
#pragma unroll
        for (int rr = 0; rr < WR; ++rr)
#pragma unroll
          for (int cc = 0; cc < WC; ++cc)
            bv[slot][rr * WC + cc] = buf[b_lane + 2 * p * (LH * LW) + (rr + ky) * LW + cc * 32 + kx];
      };
      load_step(0, 0);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        const int cs = st & 1;
        // One wave per SIMD issues in order and -- measured, tools/conv_bench "abl" -- every vector-ALU,
        // LDS or memory instruction placed between two matrix instructions delays the next one by about
        // its own issue time, so the loop holds nothing but the operand reads.  First matrix instruction
        // of the step, then ALL LDS reads of the next step, then the other 2*NT-1 matrix instructions:
        // when hipcc's lgkmcnt(0) in front of the next step is reached the reads are >= (2*NT-1)*64
        // cycles old.  sched_barrier(0) pins exactly this order.
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cs][0], bv[cs][0], acc[0][0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (st == 0 && (ABL == 0 || ABL == 4 || ring == 0)) {
          // the next chunk's LDS-DMA is issued behind the first matrix instruction, not in front of it
          if (c + 1 < a.n_chunks) issue(cur, c + 1, nbuf);
          else if (next_id < n_tiles) issue(nxt, 0, nbuf);      // first chunk of the NEXT tile
        }
        if (st + 1 < NSTEP && (ABL != 2 || st == 0)) load_step(st + 1, cs ^ 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < 2; ++m)
#pragma unroll
          for (int n = 0; n < NT; ++n)
            if (m + n > 0) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[cs][m], bv[cs][n], acc[m][n], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }

    // the next tile's first chunk (issued one chunk ago) has landed before this tile's stores queue up
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    first_landed = true;

    // ----------------------------------------------------------- epilogue of this tile
    // 2*NT*16 values per lane leave through raw buffer stores: the descriptor (co tile of this image)
    // and the channel offset are wave-uniform scalars, the pixel is ONE per-lane byte offset per
    // (row, column tile), and a lane that must not write carries the offset 0xFFFFFFFF, which the
    // hardware range check drops -- no 64-bit address arithmetic and no exec-mask branch per store.
    // Values are converted 16 at a time into distinct registers before their stores issue, so neither
    // a dependent chain nor a store still reading its data register stalls the next conversion.
    float *co_base = a.out + (((size_t)cur.img * a.out_ctot + a.out_coff) + (size_t)cur.ct * CO_TILE) * out_plane;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(co_base, 0, 0x7FFFFFFF, 0x00020000);
    const int oplane = (int)out_plane;
    const bool co_full = (cur.ct + 1) * CO_TILE <= a.cout;   // wave-uniform: every channel of the tile exists
    const int kmax = a.cout - (cur.ct * CO_TILE + 4 * half); // channels k < kmax of the co tile exist for this lane
    // element-wise tail in graph order: ReLU, then (EPI 1) BatchNorm + ReLU or (EPI 2) residual + ReLU
    const size_t res_plane = (size_t)a.in_hp * a.in_wp;
    const float *res_img = EPI == 2 ? a.residual + (size_t)cur.img * a.cout * res_plane : nullptr;
    auto relu = [](float v) { return __builtin_amdgcn_fmed3f(v, 0.f, __builtin_inff()); };   // one instruction
    auto tail = [&](float v, int k, int y, int x) -> float {
      if (RELU) v = relu(v);
      if constexpr (EPI != 0) {
        const int co = cur.ct * CO_TILE + 4 * half + k;
        if constexpr (EPI == 1) v = relu(fmaf(v, a.bn_scale[co], a.bn_shift[co]));
        if constexpr (EPI == 2) {
          // the padded plane covers the whole tile (zeros outside the image); channels past cout do not exist
          const float rv = co < a.cout ? res_img[(size_t)co * res_plane + (size_t)(y + PADY) * a.in_wp + (x + PADX)] : 0.f;
          v = relu(v + rv);
        }
      }
      return v;
    };
    auto store16 = [&](const float (&pv)[16], int m, unsigned voff) {
      if (co_full) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = 32 * m + (r & 3) + 8 * (r >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv[r]), rsrc, voff, k * oplane * 4, 0);
        }
      } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int k = 32 * m + (r & 3) + 8 * (r >> 2);
          __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(pv[r]), rsrc, k < kmax ? voff : OOB, k * oplane * 4, 0);
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
            const int y = cur.y0 + wave * WR + rr;
            const int x = cur.x0 + cc * 32 + j;
            const unsigned voff = ((y < a.H) && (x < a.W)) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
            float pv[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) pv[r] = tail(acc[m][rr * WC + cc][r], 32 * m + (r & 3) + 8 * (r >> 2), y, x);
            store16(pv, m, voff);
          }
    } else {
      const int OH = a.H >> 1, OW = a.W >> 1;
#pragma unroll
      for (int m = 0; m < 2; ++m)
#pragma unroll
        for (int cc = 0; cc < WC; ++cc) {
          const int yi = cur.y0 + wave * 2, xi = cur.x0 + cc * 32 + j;
          const int y = (cur.y0 >> 1) + wave;
          const int x = xi >> 1;
          const unsigned voff = ((y < OH) && (x < OW) && !(j & 1)) ? 4u * (unsigned)(4 * half * oplane + (y + PADY) * a.out_wp + (x + PADX)) : OOB;
          float pv[16];
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int k = 32 * m + (r & 3) + 8 * (r >> 2);
            float v;
            if constexpr (EPI == 0) {   // ReLU commutes with max: one clamp instead of four
              v = fmaxf(acc[m][0 * WC + cc][r], acc[m][1 * WC + cc][r]);
              v = fmaxf(v, __shfl_xor(v, 1));
              if (RELU) v = relu(v);
            } else {
              v = fmaxf(tail(acc[m][0 * WC + cc][r], k, yi, xi), tail(acc[m][1 * WC + cc][r], k, yi + 1, xi));
              v = fmaxf(v, __shfl_xor(v, 1));
            }
            pv[r] = v;
          }
          store16(pv, m, voff);
        }
    }
    cur = nxt;
  }
  if constexpr (ABL == 4) {
    if (tid == 0) {
      a.stamps[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - stamp_c;
      a.stamps[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - stamp_r;
    }
  }
}

// ---------------------------------------------------------------------------
// K1: layers with Cin = 1 (K = 9 or 1): direct convolution on the vector ALU; such a
// layer is bound by writing its output planes, not by arithmetic.
// Weights (cout x KS*KS) and bias are wave-uniform -> scalar loads.
// One thread = one pixel, all output channels; consecutive lanes = consecutive x.
// bn_scale != nullptr: BatchNorm + ReLU after the activation (mbv1's second layer).
// ---------------------------------------------------------------------------
template <int KS, bool RELU>
__global__ __launch_bounds__(256) void conv_first_kernel(const float *__restrict__ in,
                                                         float *__restrict__ out,
                                                         const float *__restrict__ w,  // [cout][KS*KS]
                                                         const float *__restrict__ bias,
                                                         const float *__restrict__ bn_scale,
                                                         const float *__restrict__ bn_shift, int H,
                                                         int W, int hp, int wp, int out_ctot,
                                                         int out_coff, int cout, int round16 = 0) {
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
  float *op = out + ((size_t)img * out_ctot + out_coff) * plane + (size_t)(y + PADY) * wp + (x + PADX);
  for (int co = 0; co < cout; ++co) {
    float s = bias[co];
#pragma unroll
    for (int t = 0; t < TAPS; ++t) s = fmaf(w[co * TAPS + t], v[t], s);
    if (RELU) s = fmaxf(s, 0.f);
    if (bn_scale) s = fmaxf(fmaf(s, bn_scale[co], bn_shift[co]), 0.f);
    if (round16) s = (float)(_Float16)s;   // an fp32-stored tensor inside an FP16 engine holds fp16 values
    op[(size_t)co * plane] = s;
  }
}

// K1 for 3x3 without BatchNorm (conv1a of every graph): one thread = 4 consecutive pixels, so each output channel
// leaves as one aligned 16-byte store per lane (1 KiB per wave instruction) instead of four 4-byte ones; the layer
// writes cout planes and is bound by that.
template <bool RELU>
__global__ __launch_bounds__(256) void conv_first4_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                          const float *__restrict__ w,  // [cout][9]
                                                          const float *__restrict__ bias, int H, int W, int hp, int wp,
                                                          int out_ctot, int out_coff, int cout) {
  const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int img = blockIdx.z;
  if (x >= W || y >= H) return;   // W is a multiple of 8: a thread's 4 pixels are all inside or all outside
  const size_t plane = (size_t)hp * wp;
  const float *ip = in + (size_t)img * plane + (size_t)(y + PADY - 1) * wp + (x + PADX);
  float v[3][6];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const float *rp = ip + (size_t)ky * wp;
    const float4 m = *reinterpret_cast<const float4 *>(rp);
    v[ky][0] = rp[-1]; v[ky][1] = m.x; v[ky][2] = m.y; v[ky][3] = m.z; v[ky][4] = m.w; v[ky][5] = rp[4];
  }
  float *op = out + ((size_t)img * out_ctot + out_coff) * plane + (size_t)(y + PADY) * wp + (x + PADX);
  for (int co = 0; co < cout; ++co) {
    float s[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      s[i] = bias[co];
#pragma unroll
      for (int t = 0; t < 9; ++t) s[i] = fmaf(w[co * 9 + t], v[t / 3][i + t % 3], s[i]);   // same tap order as the 1-pixel kernel
      if (RELU) s[i] = fmaxf(s[i], 0.f);
    }
    *reinterpret_cast<float4 *>(op + (size_t)co * plane) = make_float4(s[0], s[1], s[2], s[3]);
  }
}

// ---------------------------------------------------------------------------
// K1b: depthwise 3x3 convolution (+ bias, ReLU) of the MobileNet graphs: 9 MACs per output, so the
// layer is HBM-bound (read a plane, write a plane).  One thread = 4 consecutive pixels of one row of
// one channel: three aligned 16-byte loads plus the two neighbours per row; the channel is
// blockIdx.z, so its 9 weights and bias are scalar loads.
// ---------------------------------------------------------------------------
template <bool RELU>
__global__ __launch_bounds__(256) void dwconv3x3_kernel(const float *__restrict__ in,
                                                        float *__restrict__ out,
                                                        const float *__restrict__ w,  // [C][9]
                                                        const float *__restrict__ bias, int C, int H,
                                                        int W, int hp, int wp) {
  const int x = (blockIdx.x * 64 + (threadIdx.x & 63)) * 4;
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int pc = blockIdx.z;  // image * C + channel
  const int c = pc % C;
  if (x >= W || y >= H) return;
  const size_t plane = (size_t)hp * wp;
  const float *ip = in + (size_t)pc * plane + (size_t)(y + PADY - 1) * wp + (x + PADX);
  float wk[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) wk[t] = w[c * 9 + t];
  const float b = bias[c];
  float s[4] = {b, b, b, b};
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const float *rp = ip + (size_t)ky * wp;
    const float4 m = *reinterpret_cast<const float4 *>(rp);
    const float r[6] = {rp[-1], m.x, m.y, m.z, m.w, rp[4]};
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) s[i] = fmaf(wk[ky * 3 + kx], r[i + kx], s[i]);
  }
  float *op = out + (size_t)pc * plane + (size_t)(y + PADY) * wp + (x + PADX);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (x + i >= W) break;   // the border columns must stay zero
    op[i] = RELU ? fmaxf(s[i], 0.f) : s[i];
  }
}

// Stand-alone 2x2/2 max-pool (squeeze graph: pool after a Concat).
template <int UNUSED = 0>   // (a template so that every translation unit may include this header)
__global__ __launch_bounds__(256) void maxpool2_kernel(const float *__restrict__ in,
                                                       float *__restrict__ out, int C, int OH,
                                                       int OW, int in_hp, int in_wp, int out_hp,
                                                       int out_wp) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z;  // image * C + channel
  if (x >= OW || y >= OH) return;
  const float *ip = in + (size_t)c * in_hp * in_wp + (size_t)(2 * y + PADY) * in_wp + (2 * x + PADX);
  const float v = fmaxf(fmaxf(ip[0], ip[1]), fmaxf(ip[in_wp], ip[in_wp + 1]));
  out[(size_t)c * out_hp * out_wp + (size_t)(y + PADY) * out_wp + (x + PADX)] = v;
}

// K6: descriptor tail -- channel-wise L2 normalisation (ONNX ReduceL2 + Div, no
// epsilon) fused with the NCHW -> NHWC transpose the reference does on the CPU
// (neural_network.cpp:339-342).  A block handles 32
// consecutive pixels of one row; each thread first walks channels for "its"
// pixel (coalesced along x), the tile goes through LDS and leaves as
// [pixel][256] rows (coalesced along channels).
template <int C, int PX = 32>
__global__ __launch_bounds__(256) void l2norm_nhwc_kernel(const float *__restrict__ in,
                                                          float *__restrict__ out, int H, int W,
                                                          int hp, int wp) {
  constexpr int CG = 256 / PX;
  __shared__ float tile[PX][C + 1];
  __shared__ float part[CG][PX];
  __shared__ float nrm[PX];
  const int x0 = blockIdx.x * PX, y = blockIdx.y, img = blockIdx.z;
  const int tid = threadIdx.x;
  const int px = tid & (PX - 1), cg = tid / PX;
  const size_t plane = (size_t)hp * wp;
  const float *ip = in + (size_t)img * C * plane + (size_t)(y + PADY) * wp + (x0 + PADX);
  float ss = 0.f;
  for (int c = cg; c < C; c += CG) {
    const float v = (x0 + px < W) ? ip[(size_t)c * plane + px] : 0.f;
    tile[px][c] = v;
    ss = fmaf(v, v, ss);
  }
  part[cg][px] = ss;
  __syncthreads();
  if (tid < PX) {
    float s = 0.f;
#pragma unroll
    for (int g = 0; g < CG; ++g) s += part[g][tid];  // fixed order
    nrm[tid] = sqrtf(s);
  }
  __syncthreads();
  float *op = out + (((size_t)img * H + y) * W + x0) * C;
  const int npx = min(PX, W - x0);
  for (int i = tid; i < npx * C; i += 256) {
    const int p = i / C, c = i - p * C;
    op[i] = tile[p][c] / nrm[p];
  }
}

}  // namespace spvo
