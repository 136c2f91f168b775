// post.hip.h -- K0 (preprocess), K7 (softmax + depth-to-space), K8-K10 (threshold,
// exact parallel greedy NMS, ranked emission) and K11 (descriptor sampling).
//
// Replaces the CPU post-processing of the reference:
//   nn.cpp = src/odml_visual_odometry/src/feature_detection_neural_network.cpp
//   base.cpp = src/odml_visual_odometry/src/feature_detection_base.cpp
// All of these kernels are HBM/latency-bound integer and byte work: no MFMA.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_mfma.hip.h"
#include "spvo_types.hip.h"

namespace spvo {

// ---------------------------------------------------------------------------
// K0: crop + cv::resize(INTER_LINEAR, 8-bit fixed point) + /255
// (base.cpp:68-121, nn.cpp:159-160).  The per-row / per-column source index and
// the two 11-bit coefficients are tabulated on the host in float32 exactly as
// OpenCV does (resize.cpp); the kernel applies the integer arithmetic:
//   hor = S[x0]*a0 + S[x1]*a1                         (int, scaled 2^11)
//   out = (((b0*(hor0>>4))>>16) + ((b1*(hor1>>4))>>16) + 2) >> 2
// Writes the resized u8 image (what nn.cpp:154 pushes to images_dq) and the
// network input plane (padded layout, f32 = u8 * (1/255)).
// ---------------------------------------------------------------------------
struct ResizeTab {  // device arrays
  const int *xi, *xa0, *xa1;  // [W]
  const int *yi, *yb0, *yb1;  // [H]
};

// blockIdx.z selects the image of a stereo pair (src0 / src1; output slot z of out_u8 and of the input tensor)
__global__ __launch_bounds__(256) void preprocess_kernel(const uint8_t *__restrict__ src0, const uint8_t *__restrict__ src1,
                                                         size_t stride, int src_rows, int src_cols,
                                                         int row_off, int col_off, int crop_rows,
                                                         int crop_cols, ResizeTab tab, int H, int W,
                                                         uint8_t *__restrict__ out_u8,
                                                         float *__restrict__ out_plane, size_t plane_per_image, int hp,
                                                         int wp, int identity) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= W || y >= H) return;
  const uint8_t *src = blockIdx.z ? src1 : src0;
  if (out_u8) out_u8 += (size_t)blockIdx.z * H * W;
  out_plane += (size_t)blockIdx.z * plane_per_image;
  int v;
  if (identity) {  // cv::resize copies when the sizes already match
    v = src[(size_t)(row_off + y) * stride + col_off + x];
  } else {
    const int x0 = tab.xi[x], x1 = min(x0 + 1, crop_cols - 1);
    const int y0 = tab.yi[y], y1 = min(y0 + 1, crop_rows - 1);
    const int a0 = tab.xa0[x], a1 = tab.xa1[x], b0 = tab.yb0[y], b1 = tab.yb1[y];
    const uint8_t *r0 = src + (size_t)(row_off + y0) * stride + col_off;
    const uint8_t *r1 = src + (size_t)(row_off + y1) * stride + col_off;
    const int h0 = (int)r0[x0] * a0 + (int)r0[x1] * a1;
    const int h1 = (int)r1[x0] * a0 + (int)r1[x1] * a1;
    v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
    v = min(max(v, 0), 255);
  }
  if (out_u8) out_u8[(size_t)y * W + x] = (uint8_t)v;
  out_plane[(size_t)(y + PADY) * wp + (x + PADX)] = mul_rn((float)v, 1.0f / 255.0f);
}

// device buffer -> pinned host mirror, 16 bytes per lane (posted PCIe writes; no SDMA engine: see spvo_detect.hip)
__global__ __launch_bounds__(256) void mirror_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
}

// the descriptors of both images of a pair, device slots -> pinned host mirrors: one wave per 1 KiB row, only the rows that exist
// (the counts are read on the device)
struct MirrorDescJob { const float *src[2]; const int *n[2]; float *dst[2]; };
__global__ __launch_bounds__(256) void mirror_desc_kernel(MirrorDescJob job) {
  const int img = blockIdx.y, lane = threadIdx.x & 63;
  const int n = *job.n[img];
  const uint4 *s = reinterpret_cast<const uint4 *>(job.src[img]);
  uint4 *d = reinterpret_cast<uint4 *>(job.dst[img]);
  for (int row = blockIdx.x * 4 + (threadIdx.x >> 6); row < n; row += gridDim.x * 4) d[(size_t)row * 64 + lane] = s[(size_t)row * 64 + lane];
}

// Dense f32 [B,1,H,W] -> padded input planes (spvo_forward's host-input path).
__global__ __launch_bounds__(256) void pad_input_kernel(const float *__restrict__ in,
                                                        float *__restrict__ out, int H, int W,
                                                        int hp, int wp) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int img = blockIdx.z;
  if (x >= W || y >= H) return;
  out[(size_t)img * hp * wp + (size_t)(y + PADY) * wp + (x + PADX)] = in[((size_t)img * H + y) * W + x];
}

// Padded planes -> dense NCHW (test hook / det output).
__global__ __launch_bounds__(256) void unpad_kernel(const float *__restrict__ in,
                                                    float *__restrict__ out, int C, int H, int W,
                                                    int hp, int wp) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int c = blockIdx.z;  // image*C + channel
  if (x >= W || y >= H) return;
  out[((size_t)c * H + y) * W + x] = in[(size_t)c * hp * wp + (size_t)(y + PADY) * wp + (x + PADX)];
}

// ---------------------------------------------------------------------------
// K7: softmax over 65 channels (exp without max-subtraction, +1e-5 in the
// denominator), drop the dustbin, depth-to-space 8x8  (nn.cpp:266-326):
//   heat[8i+u][8j+v] = exp(det[8u+v][i][j]) / (sum_c exp(det[c][i][j]) + 1e-5)
// One thread per coarse cell; reads are coalesced along j.  `det` is either the
// padded-plane tensor (PADDED) or a dense [65][Hc][Wc] array.
// ---------------------------------------------------------------------------
template <bool PADDED>
__global__ __launch_bounds__(256) void heatmap_kernel(const float *__restrict__ det,
                                                      float *__restrict__ heat, int Hc, int Wc,
                                                      int hp, int wp) {
  const int j = blockIdx.x * 64 + (threadIdx.x & 63);
  const int i = blockIdx.y * 4 + (threadIdx.x >> 6);
  const int img = blockIdx.z;
  if (i >= Hc || j >= Wc) return;
  const size_t plane = PADDED ? (size_t)hp * wp : (size_t)Hc * Wc;
  const float *p = det + (size_t)img * 65 * plane +
                   (PADDED ? (size_t)(i + PADY) * wp + (j + PADX) : (size_t)i * Wc + j);
  float e[65];
  float s = 0.f;
#pragma unroll
  for (int c = 0; c < 65; ++c) {
    e[c] = expf(p[(size_t)c * plane]);
    s = add_rn(s, e[c]);
  }
  s = add_rn(s, 0.00001f);
  const int W = Wc * 8;
  float *h = heat + (size_t)img * (Hc * 8) * W + (size_t)(i * 8) * W + j * 8;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    float4 lo, hi;
    lo.x = __fdiv_rn(e[u * 8 + 0], s);
    lo.y = __fdiv_rn(e[u * 8 + 1], s);
    lo.z = __fdiv_rn(e[u * 8 + 2], s);
    lo.w = __fdiv_rn(e[u * 8 + 3], s);
    hi.x = __fdiv_rn(e[u * 8 + 4], s);
    hi.y = __fdiv_rn(e[u * 8 + 5], s);
    hi.z = __fdiv_rn(e[u * 8 + 6], s);
    hi.w = __fdiv_rn(e[u * 8 + 7], s);
    *(float4 *)(h + (size_t)u * W) = lo;
    *(float4 *)(h + (size_t)u * W + 4) = hi;
  }
}

// ---------------------------------------------------------------------------
// K8-K10: processOneHeatmap (nn.cpp:188-262) as an exact parallel algorithm.
//
// Rank: the reference sorts candidates by confidence descending with an
// unstable std::sort after walking a column-major sparse matrix; the total
// order pinned here (and in the oracle) is
//     key = (0xFFFFFFFF - float_bits(conf)) << 32 | (x*H + y),  smaller = earlier.
// Greedy NMS visits candidates in rank order and keeps one iff no earlier KEPT
// candidate lies within Chebyshev distance `dist` (nn.cpp:229-254).  That is the
// lexicographically-first maximal independent set of the "within dist" graph,
// which is computed without sorting by monotone local decisions:
//     UNDECIDED -> SUPPRESSED  if a KEPT candidate is in its window
//     UNDECIDED -> KEPT        if no KEPT and no better-ranked UNDECIDED candidate
//                              is in its window
// Decisions are final and each one equals the greedy outcome, whatever order or
// staleness the neighbour states are observed with (a stale read can only show
// UNDECIDED, which defers the decision).  The best undecided candidate of the
// whole image is always decided, so the loop terminates.
// Border candidates suppress but are not emitted (nn.cpp:239-244); the cap keeps
// the first `max_kp` emitted in rank order (nn.cpp:256-257).
// ---------------------------------------------------------------------------
enum : uint8_t { ST_NONE = 0, ST_UNDECIDED = 1, ST_KEPT = 2, ST_SUPPRESSED = 4 };   // one bit each: word-wide tests

// (NMS_PAD, NMS_COUNTER_INTS, nms_state_pitch, NmsBuffers, NmsPair: spvo_types.hip.h)

__device__ __forceinline__ unsigned long long rank_key(float conf, int x, int y, int H) {
  return ((unsigned long long)(0xFFFFFFFFu - __float_as_uint(conf)) << 32) | (unsigned)(x * H + y);
}

// K8: threshold + stream compaction.  One atomic per workgroup (wave ballots + LDS prefix), so
// the candidate list comes out grouped by 64x4-pixel tiles: neighbours in the image are
// neighbours in the list, which keeps most NMS decisions inside one workgroup of K10.
__global__ __launch_bounds__(256) void nms_threshold_kernel(const float *__restrict__ heat, int H,
                                                            int W, float thresh, NmsPair np) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int x = blockIdx.x * 64 + lane;
  const int y = blockIdx.y * 4 + wave;
  const NmsBuffers nb = np.b[blockIdx.z];
  const bool valid = x < W && y < H;
  const int p = y * W + x;
  const bool c = valid && heat[(size_t)blockIdx.z * H * W + p] > thresh;  // strict, nn.cpp:203
  if (valid) nb.state[(y + NMS_PAD) * nms_state_pitch(W) + x + NMS_PAD] = c ? ST_UNDECIDED : ST_NONE;
  const unsigned long long m = __ballot(c);
  if (lane == 0) s_wave[wave] = __popcll(m);
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = tot ? atomicAdd(&nb.counters[0], tot) : 0;
  }
  __syncthreads();
  if (c) {
    int off = s_base + __popcll(m & ((1ull << lane) - 1ull));
    for (int w = 0; w < wave; ++w) off += s_wave[w];
    nb.cand[off] = p;
  }
}

// K7 + K8 fused (product path): one thread per coarse cell computes its 8x8 block of the heat
// map, writes it, writes the NMS state bytes and appends its candidates to the list.  ONE atomic
// per workgroup (256 cells = 16 384 pixels): 72 atomics per stereo pair instead of one per
// 64x4 tile, and no separate pass over the heat map.
__global__ __launch_bounds__(256) void heatmap_nms_kernel(const float *__restrict__ det,
                                                          float *__restrict__ heat, int Hc, int Wc,
                                                          int hp, int wp, float thresh, NmsPair np) {
  __shared__ int s_wave[4];
  __shared__ int s_base;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int j = blockIdx.x * 64 + lane;
  const int i = blockIdx.y * 4 + wave;
  const int img = blockIdx.z;
  const NmsBuffers nb = np.b[img];
  const bool valid = i < Hc && j < Wc;
  const int W = Wc * 8, H = Hc * 8;
  const int pitch = nms_state_pitch(W);
  unsigned long long cm = 0;   // bit 8u+v: pixel (8i+u, 8j+v) is a candidate
  if (valid) {
    const size_t plane = (size_t)hp * wp;
    const float *p = det + (size_t)img * 65 * plane + (size_t)(i + PADY) * wp + (j + PADX);
    float e[65];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < 65; ++c) {
      e[c] = expf(p[(size_t)c * plane]);
      s = add_rn(s, e[c]);
    }
    s = add_rn(s, 0.00001f);
    float *h = heat + (size_t)img * H * W + (size_t)(i * 8) * W + j * 8;
    uint8_t *st = nb.state + (size_t)(i * 8 + NMS_PAD) * pitch + j * 8 + NMS_PAD;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      float v[8];
      uint32_t lo = 0, hi = 0;
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        v[k] = __fdiv_rn(e[u * 8 + k], s);
        const bool c = v[k] > thresh;                       // strict, nn.cpp:203
        if (c) cm |= 1ull << (u * 8 + k);
        if (k < 4) lo |= (c ? (uint32_t)ST_UNDECIDED : 0u) << (8 * k);
        else hi |= (c ? (uint32_t)ST_UNDECIDED : 0u) << (8 * (k - 4));
      }
      *(float4 *)(h + (size_t)u * W) = make_float4(v[0], v[1], v[2], v[3]);
      *(float4 *)(h + (size_t)u * W + 4) = make_float4(v[4], v[5], v[6], v[7]);
      *(uint32_t *)(st + (size_t)u * pitch) = lo;
      *(uint32_t *)(st + (size_t)u * pitch + 4) = hi;
    }
  }
  // block-wide exclusive prefix of the per-thread candidate counts
  const int cnt = __popcll(cm);
  int incl = cnt;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) s_wave[wave] = incl;
  __syncthreads();
  if (threadIdx.x == 0) {
    const int tot = s_wave[0] + s_wave[1] + s_wave[2] + s_wave[3];
    s_base = tot ? atomicAdd(&nb.counters[0], tot) : 0;
  }
  __syncthreads();
  int off = s_base + incl - cnt;
  for (int w = 0; w < wave; ++w) off += s_wave[w];
  while (cm) {
    const int b = __ffsll((long long)cm) - 1;
    cm &= cm - 1;
    nb.cand[off++] = (i * 8 + (b >> 3)) * W + j * 8 + (b & 7);
  }
}

// decision for one undecided candidate: returns ST_KEPT / ST_SUPPRESSED / ST_UNDECIDED
template <int DIST>
__device__ __forceinline__ uint8_t nms_decide(const float *__restrict__ hm, uint8_t *state, int H,
                                              int W, int pitch, int dist_rt, int p, int x, int y) {
  const unsigned long long key = rank_key(hm[p], x, y, H);
  bool any_kept = false, any_better = false;
  if constexpr (DIST > 0) {
    // state bytes and confidences of the whole window are fetched up front with independent
    // aligned loads (27 words + 27 float4 for the 9x9 window); decisions are pure ALU after that
    constexpr int NW = (2 * DIST + 1 + 3 + 3) / 4;  // words that cover the window from an aligned start
    const int a = (x + NMS_PAD - DIST) & ~3;        // aligned first column (padded coordinates)
    const int ah = a - NMS_PAD;                     // same column in image coordinates (may be < 0: masked by the state)
    const float hp = hm[p];
    const int idp = x * H + y;
    uint32_t w[2 * DIST + 1][NW];
    float4 hv[2 * DIST + 1][NW];
#pragma unroll
    for (int dy = 0; dy <= 2 * DIST; ++dy) {
      const uint32_t *row = (const uint32_t *)(state + (y + NMS_PAD + dy - DIST) * pitch + a);
      const int yc = min(max(y + dy - DIST, 0), H - 1);   // rows outside the image are ST_NONE anyway
      const float4 *hrow = (const float4 *)(hm + (ptrdiff_t)yc * W + ah);
#pragma unroll
      for (int k = 0; k < NW; ++k) {
        w[dy][k] = __builtin_nontemporal_load(row + k);
        hv[dy][k] = hrow[k];
      }
    }
    // word-wide tests: the window is the 2*DIST+1 bytes that start `o` bytes into the first word
    const int o = (x + NMS_PAD - DIST) & 3;
    uint32_t wmask[NW];
#pragma unroll
    for (int k = 0; k < NW; ++k) {
      const int lo = 4 * k - o, hi = 4 * k + 3 - o;          // byte positions relative to the window start
      uint32_t m = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) m |= ((lo + b >= 0 && lo + b <= 2 * DIST) ? 0xFFu : 0u) << (8 * b);
      (void)hi;
      wmask[k] = m;
    }
    uint32_t kept = 0;
#pragma unroll
    for (int dy = 0; dy <= 2 * DIST; ++dy)
#pragma unroll
      for (int k = 0; k < NW; ++k) kept |= w[dy][k] & wmask[k];
    any_kept = (kept & (0x01010101u * ST_KEPT)) != 0;
    if (!any_kept) {
#pragma unroll
      for (int dy = 0; dy <= 2 * DIST; ++dy)
#pragma unroll
        for (int k = 0; k < NW; ++k) {
          uint32_t u = w[dy][k] & wmask[k] & (0x01010101u * ST_UNDECIDED);
          while (u) {                                    // rare: undecided neighbours only
            const int b = (__ffs((int)u) - 1) >> 3;
            u &= u - 1;
            const int xx = ah + 4 * k + b, yy = y + dy - DIST;
            const float4 h4 = hv[dy][k];
            const float hq = b == 0 ? h4.x : b == 1 ? h4.y : b == 2 ? h4.z : h4.w;
            // rank_key(q) < rank_key(p): higher confidence first, then smaller column-major index
            if ((yy != y || xx != x) && (hq > hp || (hq == hp && xx * H + yy < idp))) any_better = true;
          }
        }
    }
  } else {
    for (int yy = y - dist_rt; yy <= y + dist_rt; ++yy)
      for (int xx = x - dist_rt; xx <= x + dist_rt; ++xx) {
        const uint8_t s = ((volatile uint8_t *)state)[(yy + NMS_PAD) * pitch + xx + NMS_PAD];
        if (s == ST_KEPT) any_kept = true;
        else if (s == ST_UNDECIDED && (yy != y || xx != x) && rank_key(hm[yy * W + xx], xx, yy, H) < key) any_better = true;
      }
  }
  return any_kept ? ST_SUPPRESSED : (any_better ? ST_UNDECIDED : ST_KEPT);
}

// K10.  Grid-stride over the candidate list; the first candidate of every thread (all of them
// when n <= grid size, the usual case) is tracked in registers, so decided threads cost nothing
// in the following in-kernel rounds.  DIST > 0: compile-time radius, every window row is fetched
// with 3 independent aligned word loads (27 loads in flight for the 9x9 window);
// DIST == 0: run-time radius (<= NMS_PAD).
template <int INNER, int DIST>
// `collect`: a candidate that is decided KEPT is entered into the survivor list on the spot (decisions are final, so the list is
// complete when nothing is undecided) and no separate collect launch is needed; the host-driven continuation after an unsettled
// first batch clears the list and runs nms_collect_kernel over all candidates instead (collect = 0).
__global__ __launch_bounds__(256) void nms_round_kernel(const float *__restrict__ heat, int H, int W,
                                                        int dist_rt, NmsPair np, int launch, int border, int surv_cap, int collect) {
  const NmsBuffers nb = np.b[blockIdx.y];
  if (launch > 0 && nb.counters[8 + launch - 1] == 0) return;  // nothing left undecided
  const float *hm = heat + (size_t)blockIdx.y * H * W;
  const int n = nb.counters[0];
  const int pitch = nms_state_pitch(W);
  const int gid = blockIdx.x * 256 + threadIdx.x, stride = gridDim.x * 256;
  uint8_t *state = nb.state;
  int p0 = 0, x0 = 0, y0 = 0, sp0 = 0;
  bool live0 = false;
  if (gid < n) {
    p0 = nb.cand[gid];
    y0 = p0 / W;
    x0 = p0 - y0 * W;
    sp0 = (y0 + NMS_PAD) * pitch + x0 + NMS_PAD;
    live0 = ((volatile uint8_t *)state)[sp0] == ST_UNDECIDED;
  }
  const bool extra = gid + stride < n;
  auto survive = [&](int p, int x, int y) {   // nn.cpp:239-242: a kept candidate inside the border is emitted
    if (y >= border && y + border < H && x >= border && x + border < W) {
      const int s = atomicAdd(&nb.counters[1], 1);
      if (s < surv_cap) nb.surv_key[s] = rank_key(hm[p], x, y, H);
      else nb.counters[3] = 1;
    }
  };
  for (int it = 0; it < INNER; ++it) {
    if (live0) {
      const uint8_t d = nms_decide<DIST>(hm, state, H, W, pitch, dist_rt, p0, x0, y0);
      if (d != ST_UNDECIDED) {
        ((volatile uint8_t *)state)[sp0] = d;
        live0 = false;
        if (d == ST_KEPT && collect) survive(p0, x0, y0);
      }
    }
    if (extra)
      for (int i = gid + stride; i < n; i += stride) {
        const int p = nb.cand[i];
        const int y = p / W, x = p - y * W;
        const int sp = (y + NMS_PAD) * pitch + x + NMS_PAD;
        if (((volatile uint8_t *)state)[sp] != ST_UNDECIDED) continue;
        const uint8_t d = nms_decide<DIST>(hm, state, H, W, pitch, dist_rt, p, x, y);
        if (d != ST_UNDECIDED) ((volatile uint8_t *)state)[sp] = d;
        if (d == ST_KEPT && collect) survive(p, x, y);
      }
    if (!__syncthreads_or(live0 || extra)) break;
  }
  int rem = live0 ? 1 : 0;
  if (extra)
    for (int i = gid + stride; i < n; i += stride) {
      const int p = nb.cand[i];
      const int y = p / W, x = p - y * W;
      rem += (((volatile uint8_t *)state)[(y + NMS_PAD) * pitch + x + NMS_PAD] == ST_UNDECIDED) ? 1 : 0;
    }
  const unsigned long long m = __ballot(rem != 0);
  if (m) {
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) rem += __shfl_xor(rem, o);
    if ((threadIdx.x & 63) == 0) atomicAdd(&nb.counters[8 + launch], rem);
  }
}

// K10f: the stragglers.  After the first round launches a real heat map has a few hundred candidates left whose decision hangs on a
// chain of better neighbours; every further round launch decides one more link per chain (a workgroup sees what the others wrote at
// the next kernel boundary at the latest).  ONE workgroup per image takes them all: it lists the undecided candidates in LDS and
// iterates rounds over that list with workgroup barriers between them -- all writers and readers of the state bytes sit on one CU,
// so every round sees the round before it -- until nothing is undecided (or its budget of rounds / window evaluations is spent).  Decisions are final and unique whatever the order they
// are taken in (nms_decide), so the keypoints are the ones the round launches alone would give.  More undecided candidates than the
// list holds, or chains longer than the round budget (adversarial heat maps), are left to the host's continuation (the count of what
// is left goes into the launch's slot of the counter block, as after a round launch).
constexpr int NMS_FIN_THREADS = 512, NMS_FIN_CAP = 8192, NMS_FIN_ROUNDS = 1024, NMS_FIN_EVALS = 768;
template <int DIST>
__global__ __launch_bounds__(NMS_FIN_THREADS) void nms_finish_kernel(const float *__restrict__ heat, int H, int W, int dist_rt, NmsPair np, int launch, int border,
                                                                     int surv_cap) {
  __shared__ int s_list[NMS_FIN_CAP];
  __shared__ int s_n, s_rem;
  const NmsBuffers nb = np.b[blockIdx.x];
  if (nb.counters[8 + launch - 1] == 0) return;   // nothing left undecided
  const float *hm = heat + (size_t)blockIdx.x * H * W;
  const int n = nb.counters[0];
  const int pitch = nms_state_pitch(W);
  uint8_t *state = nb.state;
  if (threadIdx.x == 0) { s_n = 0; s_rem = 0; }
  __syncthreads();
  for (int i = threadIdx.x; i < n; i += NMS_FIN_THREADS) {
    const int p = nb.cand[i];
    const int y = p / W, x = p - y * W;
    if (((volatile uint8_t *)state)[(y + NMS_PAD) * pitch + x + NMS_PAD] == ST_UNDECIDED) {
      const int k = atomicAdd(&s_n, 1);
      if (k < NMS_FIN_CAP) s_list[k] = p;
    }
  }
  __syncthreads();
  const int total = s_n;
  if (total > NMS_FIN_CAP) {   // not a job for one workgroup
    if (threadIdx.x == 0) atomicAdd(&nb.counters[8 + launch], total);
    return;
  }
  // budget: NMS_FIN_ROUNDS rounds, NMS_FIN_EVALS window evaluations per thread (~2 us each: about a millisecond in the worst case)
  bool any = total > 0;
  int evals = 0;
  for (int round = 0; round < NMS_FIN_ROUNDS && any; ++round) {
    bool live = false;
    for (int k = threadIdx.x; k < total; k += NMS_FIN_THREADS) {
      const int p = s_list[k];
      if (p < 0) continue;
      const int y = p / W, x = p - y * W;
      ++evals;
      const uint8_t d = nms_decide<DIST>(hm, state, H, W, pitch, dist_rt, p, x, y);
      if (d == ST_UNDECIDED) { live = true; continue; }
      ((volatile uint8_t *)state)[(y + NMS_PAD) * pitch + x + NMS_PAD] = d;
      s_list[k] = -1;
      if (d == ST_KEPT && y >= border && y + border < H && x >= border && x + border < W) {   // nn.cpp:239-242, as in nms_round_kernel
        const int s = atomicAdd(&nb.counters[1], 1);
        if (s < surv_cap) nb.surv_key[s] = rank_key(hm[p], x, y, H);
        else nb.counters[3] = 1;
      }
    }
    any = __syncthreads_or(live) != 0;
    if (__syncthreads_or(evals >= NMS_FIN_EVALS)) break;
  }
  if (any) {
    int rem = 0;
    for (int k = threadIdx.x; k < total; k += NMS_FIN_THREADS) rem += s_list[k] >= 0 ? 1 : 0;
    if (rem) atomicAdd(&s_rem, rem);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&nb.counters[8 + launch], s_rem);
  }
}

__global__ __launch_bounds__(256) void nms_collect_kernel(const float *__restrict__ heat, int H,
                                                          int W, int border, int surv_cap,
                                                          NmsPair np) {
  const NmsBuffers nb = np.b[blockIdx.y];
  const float *hm = heat + (size_t)blockIdx.y * H * W;
  const int n = nb.counters[0];
  const int pitch = nms_state_pitch(W);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int p = nb.cand[i];
    const int y = p / W, x = p - y * W;
    if (nb.state[(y + NMS_PAD) * pitch + x + NMS_PAD] != ST_KEPT) continue;
    if (y >= border && y + border < H && x >= border && x + border < W) {  // nn.cpp:239-242
      const int s = atomicAdd(&nb.counters[1], 1);
      if (s < surv_cap) nb.surv_key[s] = rank_key(hm[p], x, y, H);
      else nb.counters[3] = 1;
    }
  }
}

// K9: rank by counting -- the output position of a survivor is the number of survivors with a
// smaller key.  2-D decomposition: block (bi, bj) counts 256 keys against a 1024-key LDS tile.
constexpr int RANK_TILE = 1024;
// The grid does not depend on the buffer's capacity (one survivor per 5x5 cell at most: 17 k; a few thousand in practice): a fixed
// number of workgroups per image walks the blocks the survivor count on the device calls for -- a capacity-sized grid is 2278
// workgroups per launch, most of which only start and exit, next to convolutions whose workgroups need whole CUs.
__global__ __launch_bounds__(256) void nms_rank_kernel(int surv_cap, NmsPair np) {
  __shared__ __attribute__((aligned(16))) unsigned long long tile[RANK_TILE];
  const NmsBuffers nb = np.b[blockIdx.y];
  const int n = min(nb.counters[1], surv_cap);
  const int nbi = (n + 255) / 256, nbj = (n + RANK_TILE - 1) / RANK_TILE;
  for (int b = blockIdx.x; b < nbi * nbj; b += gridDim.x) {
    const int bi = b % nbi, j0 = (b / nbi) * RANK_TILE;
    __syncthreads();
    for (int t = threadIdx.x; t < RANK_TILE; t += 256) tile[t] = (j0 + t < n) ? nb.surv_key[j0 + t] : ~0ull;
    __syncthreads();
    const int i = bi * 256 + threadIdx.x;
    if (i >= n) continue;
    const unsigned long long key = nb.surv_key[i];
    int cnt = 0;
    const ulonglong2 *t2 = (const ulonglong2 *)tile;
#pragma unroll 8
    for (int t = 0; t < RANK_TILE / 2; ++t) {
      const ulonglong2 v = t2[t];
      cnt += (v.x < key ? 1 : 0) + (v.y < key ? 1 : 0);
    }
    if (cnt) atomicAdd(&nb.rank[i], cnt);
  }
}

// host_counters: the set's pinned mirror of the counter blocks ([image][NMS_COUNTER_INTS]); this kernel is the last one that changes a
// block (entry 2, the keypoint count), so its first workgroup writes the mirror itself: no 48-byte copy behind the chain
__global__ __launch_bounds__(256) void nms_write_kernel(int H, int max_kp, int surv_cap, NmsPair np, int *zero_next, int *host_counters) {
  const NmsBuffers nb = np.b[blockIdx.y];
  // hand the next submission a clean counter block (it belongs to the other parity)
  if (zero_next && blockIdx.x == 0 && threadIdx.x < NMS_COUNTER_INTS) zero_next[blockIdx.y * NMS_COUNTER_INTS + threadIdx.x] = 0;
  const int n = min(nb.counters[1], surv_cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) nb.counters[2] = min(n, max_kp);
  if (host_counters && blockIdx.x == 0 && threadIdx.x < NMS_COUNTER_INTS)
    host_counters[blockIdx.y * NMS_COUNTER_INTS + threadIdx.x] = threadIdx.x == 2 ? min(n, max_kp) : nb.counters[threadIdx.x];
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int rank = nb.rank[i];
    nb.rank[i] = 0;
    if (rank < max_kp) {
      const unsigned cm = (unsigned)(nb.surv_key[i] & 0xFFFFFFFFull);
      nb.out_xy[2 * rank + 0] = (int)(cm / (unsigned)H);
      nb.out_xy[2 * rank + 1] = (int)(cm % (unsigned)H);
    }
  }
}

// ---------------------------------------------------------------------------
// K11: bilinearInterpolationDesc (nn.cpp:366-431): align_corners=True mapping,
// 4-tap blend, re-normalise.  One wave per keypoint, 4 channels per lane
// (16-byte loads from the NHWC map: each tap is one contiguous 1 KiB row).
// Blend order as written at nn.cpp:423-427 with no FMA contraction.
// The count comes from device memory so no host round trip is needed.
// ---------------------------------------------------------------------------
struct SampleJob {
  const float *desc_nhwc;   // this image's descriptor map
  const int *xy;            // keypoints [n][2] int
  const int *n_ptr;         // device count (or NULL -> n_fixed)
  int n_fixed;
  float *out;               // [n][256]
  float *out_sqn;           // [n] squared norm of the stored descriptor (for K12a) or NULL
  float *out_xy_f32;        // [n][2] or NULL
  int *out_xy_i32;          // [n][2] or NULL
  int *out_n;               // device copy of n or NULL
  float *out_host;          // [n][256] second copy of the descriptors, written straight into pinned HOST memory, or NULL
};
struct SampleJobs { SampleJob j[2]; };   // blockIdx.y selects the image

__global__ __launch_bounds__(256) void sample_desc_kernel(SampleJobs jobs, int H, int W, int Hc, int Wc) {
  const SampleJob jb = jobs.j[blockIdx.y];
  const float *__restrict__ desc_nhwc = jb.desc_nhwc;
  const int n = jb.n_ptr ? *jb.n_ptr : jb.n_fixed;
  const int k = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (jb.out_n && blockIdx.x == 0 && threadIdx.x == 0) *jb.out_n = n;
  if (k >= n) return;
  const int col = jb.xy[2 * k], row = jb.xy[2 * k + 1];
  if (jb.out_xy_f32 && lane == 0) {
    jb.out_xy_f32[2 * k] = (float)col;
    jb.out_xy_f32[2 * k + 1] = (float)row;
  }
  if (jb.out_xy_i32 && lane == 0) {
    jb.out_xy_i32[2 * k] = col;
    jb.out_xy_i32[2 * k + 1] = row;
  }
  float *out = jb.out;
  const float row8 = mul_rn(__fdiv_rn((float)row, (float)(H - 1)), (float)(Hc - 1));
  const float col8 = mul_rn(__fdiv_rn((float)col, (float)(W - 1)), (float)(Wc - 1));
  const int r0 = (int)floorf(row8), c0 = (int)floorf(col8);
  const float rr = sub_rn(1.0f, sub_rn(row8, (float)r0));
  const float cr = sub_rn(1.0f, sub_rn(col8, (float)c0));
  const float rr1 = sub_rn(1.0f, rr), cr1 = sub_rn(1.0f, cr);
  const int r1 = min(r0 + 1, Hc - 1), c1 = min(c0 + 1, Wc - 1);
  const float4 tl = *(const float4 *)(desc_nhwc + ((size_t)r0 * Wc + c0) * 256 + lane * 4);
  const float4 tr = *(const float4 *)(desc_nhwc + ((size_t)r0 * Wc + c1) * 256 + lane * 4);
  const float4 bl = *(const float4 *)(desc_nhwc + ((size_t)r1 * Wc + c0) * 256 + lane * 4);
  const float4 br = *(const float4 *)(desc_nhwc + ((size_t)r1 * Wc + c1) * 256 + lane * 4);
  auto blend = [&](float a, float b, float c, float d) {
    const float t0 = mul_rn(mul_rn(a, rr), cr);
    const float t1 = mul_rn(mul_rn(b, rr), cr1);
    const float t2 = mul_rn(mul_rn(c, rr1), cr);
    const float t3 = mul_rn(mul_rn(d, rr1), cr1);
    return add_rn(add_rn(add_rn(t0, t1), t2), t3);
  };
  float4 v;
  v.x = blend(tl.x, tr.x, bl.x, br.x);
  v.y = blend(tl.y, tr.y, bl.y, br.y);
  v.z = blend(tl.z, tr.z, bl.z, br.z);
  v.w = blend(tl.w, tr.w, bl.w, br.w);
  float ss = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) ss += __shfl_xor(ss, o);
  const float nrm = sqrtf(ss);
  v.x = __fdiv_rn(v.x, nrm);
  v.y = __fdiv_rn(v.y, nrm);
  v.z = __fdiv_rn(v.z, nrm);
  v.w = __fdiv_rn(v.w, nrm);
  *(float4 *)(out + (size_t)k * 256 + lane * 4) = v;
  if (jb.out_host) *(float4 *)(jb.out_host + (size_t)k * 256 + lane * 4) = v;   // 1 KiB per wave, posted writes over PCIe
  if (jb.out_sqn) {
    float s2 = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) s2 += __shfl_xor(s2, o);
    if (lane == 0) jb.out_sqn[k] = s2;
  }
}

}  // namespace spvo
