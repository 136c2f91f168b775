// orb.hip.h -- the classic front end's detector / extractor on the GPU: ORB keypoints + steered BRIEF descriptors
// (ClassicFeatureFrontEnd with DetectorType::ORB / DescriptorType::ORB, feature_detection_classic.cpp:7-79:
// cv::ORB::create(2000, 1.2f, 8, 31, 0, 2, FAST_SCORE, 31, 20) as detector, cv::ORB::create() as extractor) -- SURVEY.md
// section 8a row U / 8f rank 4.  The reference obtains these from OpenCV; the algorithm built here is the published one
// (Rublee et al., ICCV 2011) with the reference's parameters, in exactly the form oracle/cpu/orb_cpu.inc restates it (that
// file's header lists the choices the publication leaves open and the one deviation -- the 256 test pairs are drawn from
// the BRIEF Gaussian with a fixed seed because OpenCV's learned table cannot be restated): 8-level pyramid (8-bit bilinear
// resize, level from level), FAST-9/16 with threshold 20, 3x3 non-maximum suppression on the FAST score, border 31, the
// best nfeatures-per-level by (response, column-major index), intensity-centroid direction over the disc of radius 15, 256
// steered tests on the 7x7-Gaussian-smoothed level.  Integer stages are exact by nature; the float stages (smoothing, the
// direction, the rotation of the test points) use separately rounded IEEE operations in a fixed order (mul_rn / add_rn,
// __fsqrt_rn, __fdiv_rn), so keypoints and descriptors equal the CPU restatement's bit for bit.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "post.hip.h"

namespace spvo {

constexpr int ORB_LEVELS = 8, ORB_HALF = 15, ORB_EDGE = 31, ORB_FAST_T = 20, ORB_DISC = 709;   // 709 pixels in the disc of radius 15

// (OrbKeypoint: spvo_types.hip.h)

// pyramid level from the level below: cv::resize(INTER_LINEAR) on 8-bit data, the arithmetic of preprocess_kernel
__global__ __launch_bounds__(256) void orb_resize_kernel(const uint8_t *__restrict__ src, int sh, int sw, int sstride, uint8_t *__restrict__ dst, int dh, int dw,
                                                         const int *__restrict__ tab /* xi, xa0, xa1 [dw]; yi, yb0, yb1 [dh] */) {
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= dw || y >= dh) return;
  const int *xi = tab, *xa0 = tab + dw, *xa1 = tab + 2 * dw, *yi = tab + 3 * dw, *yb0 = yi + dh, *yb1 = yi + 2 * dh;
  const int x0 = xi[x], x1 = min(x0 + 1, sw - 1), y0 = yi[y], y1 = min(y0 + 1, sh - 1);
  const uint8_t *r0 = src + (size_t)y0 * sstride, *r1 = src + (size_t)y1 * sstride;
  const int s0 = ((int)r0[x0] * xa0[x] + (int)r0[x1] * xa1[x]) >> 4, s1 = ((int)r1[x0] * xa0[x] + (int)r1[x1] * xa1[x]) >> 4;
  const int v = (((yb0[y] * s0) >> 16) + ((yb1[y] * s1) >> 16) + 2) >> 2;
  dst[(size_t)y * dw + x] = (uint8_t)min(max(v, 0), 255);
}

// Everything after the pyramid runs for ALL levels in one launch per stage (blockIdx.z / .y = level; a level's workgroups beyond its
// size exit at once): an image is 7 resize launches + 7 stage launches instead of 8 x 8 -- the chain is launch-bound, not
// work-bound (a stage of one level takes 2-13 us).
struct OrbLevel {
  uint8_t *im, *score, *blur;
  float *tmp;
  unsigned long long *keys;
  int *rank, *out_xy, *counters;
  int h, w, want, cap;   // want = 0: the level is skipped (smaller than the border, or no quota)
  float scale;
};
struct OrbLevels { OrbLevel l[ORB_LEVELS]; };

// FAST-9 corner score on the Bresenham circle of radius 3: the largest threshold at which the pixel is still a corner, 0 if it is not
// one at threshold t (or lies in the border)
__global__ __launch_bounds__(256) void orb_fast_kernel(const OrbLevels lv, int t) {
  const OrbLevel L = lv.l[blockIdx.z];
  const int h = L.h, w = L.w;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (L.want <= 0 || x >= w || y >= h) return;
  int best = 0;
  if (x >= ORB_EDGE && x < w - ORB_EDGE && y >= ORB_EDGE && y < h - ORB_EDGE) {
    const uint8_t *p = L.im + (size_t)y * w + x;
    const int c = *p;
    const int off[16] = {-3 * w, -3 * w + 1, -2 * w + 2, -w + 3, 3, w + 3, 2 * w + 2, 3 * w + 1, 3 * w, 3 * w - 1, 2 * w - 2, w - 3, -3, -w - 3, -2 * w - 2, -3 * w - 1};
    int d[25];
#pragma unroll
    for (int i = 0; i < 16; ++i) d[i] = (int)p[off[i]] - c;
    const int nb = (d[0] > t) + (d[4] > t) + (d[8] > t) + (d[12] > t), nd = (d[0] < -t) + (d[4] < -t) + (d[8] < -t) + (d[12] < -t);
    if (nb >= 2 || nd >= 2) {
#pragma unroll
      for (int i = 16; i < 25; ++i) d[i] = d[i - 16];
#pragma unroll
      for (int s = 0; s < 16; ++s) {
        int mn = 255, mx = -255;
#pragma unroll
        for (int k = 0; k < 9; ++k) { mn = min(mn, d[s + k]); mx = max(mx, d[s + k]); }
        if (mn > t) best = max(best, mn);
        if (-mx > t) best = max(best, -mx);
      }
    }
  }
  L.score[(size_t)y * w + x] = (uint8_t)best;
}

// 3x3 non-maximum suppression (of equal neighbours the first in raster order wins) -> rank keys of the survivors
__global__ __launch_bounds__(256) void orb_collect_kernel(const OrbLevels lv) {
  const OrbLevel L = lv.l[blockIdx.z];
  const int h = L.h, w = L.w;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (L.want <= 0 || y >= h) return;   // (whole waves: a wave is 64 consecutive x of one row)
  const uint8_t *score = L.score;
  const bool inside = x >= ORB_EDGE && x < w - ORB_EDGE && y >= ORB_EDGE && y < h - ORB_EDGE;
  const int s = inside ? score[(size_t)y * w + x] : 0;
  bool is_max = s != 0;
  if (is_max) {
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
      for (int dx = -1; dx <= 1; ++dx)
        if (dy || dx) {
          const int later = (dy > 0 || (dy == 0 && dx > 0)) ? 1 : 0;
          if ((int)score[(size_t)(y + dy) * w + x + dx] >= s + later) is_max = false;
        }
  }
  // one atomic per wave: the survivors of a wave take consecutive slots
  const unsigned long long m = __ballot(is_max);
  if (!m) return;
  const int lane = threadIdx.x & 63;
  int base = 0;
  if (lane == __ffsll((long long)m) - 1) base = atomicAdd(&L.counters[1], __popcll(m));
  base = __shfl(base, __ffsll((long long)m) - 1);
  if (!is_max) return;
  const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
  if (slot < L.cap) L.keys[slot] = rank_key((float)s, x, y, h);   // (0xFFFFFFFF - bits(response)) << 32 | x h + y: smaller = earlier
  else L.counters[3] = 1;
}

// K9's rank-by-counting (nms_rank_kernel) with a grid that does not depend on the buffers' capacity: a level has anything between
// no corner and one per 2x2 pixels.  A fixed number of workgroups per level walks the (256 keys) x (1024-key tile) blocks that the
// survivor count on the device calls for.
__global__ __launch_bounds__(256) void orb_rank_kernel(const OrbLevels lv) {
  __shared__ __attribute__((aligned(16))) unsigned long long tile[RANK_TILE];
  const OrbLevel L = lv.l[blockIdx.y];
  if (L.want <= 0) return;
  const int n = min(L.counters[1], L.cap);
  const int nbi = (n + 255) / 256, nbj = (n + RANK_TILE - 1) / RANK_TILE;
  for (int b = blockIdx.x; b < nbi * nbj; b += gridDim.x) {
    const int bi = b % nbi, j0 = (b / nbi) * RANK_TILE;
    __syncthreads();
    for (int t = threadIdx.x; t < RANK_TILE; t += 256) tile[t] = (j0 + t < n) ? L.keys[j0 + t] : ~0ull;
    __syncthreads();
    const int i = bi * 256 + threadIdx.x;
    if (i >= n) continue;
    const unsigned long long key = L.keys[i];
    int cnt = 0;
    const ulonglong2 *t2 = (const ulonglong2 *)tile;
#pragma unroll 8
    for (int t = 0; t < RANK_TILE / 2; ++t) {
      const ulonglong2 v = t2[t];
      cnt += (v.x < key ? 1 : 0) + (v.y < key ? 1 : 0);
    }
    if (cnt) atomicAdd(&L.rank[i], cnt);
  }
}

// positions from ranks: the first `want` of a level in (response, index) order
__global__ __launch_bounds__(256) void orb_write_kernel(const OrbLevels lv) {
  const OrbLevel L = lv.l[blockIdx.y];
  if (L.want <= 0) return;
  const int n = min(L.counters[1], L.cap);
  if (blockIdx.x == 0 && threadIdx.x == 0) L.counters[2] = min(n, L.want);
  for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
    const int rank = L.rank[i];
    L.rank[i] = 0;
    if (rank < L.want) {
      const unsigned cm = (unsigned)(L.keys[i] & 0xFFFFFFFFull);
      L.out_xy[2 * rank + 0] = (int)(cm / (unsigned)L.h);
      L.out_xy[2 * rank + 1] = (int)(cm % (unsigned)L.h);
    }
  }
}

// 7x7 Gaussian, separable, float32 sums in tap order with separately rounded multiply and add, result = floor(s + 0.5) clamped
__global__ __launch_bounds__(256) void orb_blur_h_kernel(const OrbLevels lv, const float *__restrict__ taps) {
  const OrbLevel L = lv.l[blockIdx.z];
  const int h = L.h, w = L.w;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (L.want <= 0 || x >= w || y >= h) return;
  float s = 0.f;
#pragma unroll
  for (int i = -3; i <= 3; ++i) s = add_rn(s, mul_rn(taps[i + 3], (float)L.im[(size_t)y * w + min(max(x + i, 0), w - 1)]));
  L.tmp[(size_t)y * w + x] = s;
}
__global__ __launch_bounds__(256) void orb_blur_v_kernel(const OrbLevels lv, const float *__restrict__ taps) {
  const OrbLevel L = lv.l[blockIdx.z];
  const int h = L.h, w = L.w;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (L.want <= 0 || x >= w || y >= h) return;
  float s = 0.f;
#pragma unroll
  for (int i = -3; i <= 3; ++i) s = add_rn(s, mul_rn(taps[i + 3], L.tmp[(size_t)min(max(y + i, 0), h - 1) * w + x]));
  L.blur[(size_t)y * w + x] = (uint8_t)min(max((int)floorf(add_rn(s, 0.5f)), 0), 255);
}

// One wave per keypoint: moments of the disc (integers), direction, 256 steered tests on the smoothed level, the keypoint record.
// disc: the 709 (dx, dy) offsets of the disc, int8 pairs; pattern: 256 x (x1, y1, x2, y2) floats.  A level's keypoints land behind
// those of the levels below (their counts are final: the write launch before this one covered every level).
__global__ __launch_bounds__(256) void orb_describe_kernel(const OrbLevels lv, const signed char *__restrict__ disc, const float *__restrict__ pattern,
                                                           OrbKeypoint *__restrict__ kps, uint8_t *__restrict__ desc, int cap_total) {
  const int octave = blockIdx.y;
  const OrbLevel L = lv.l[octave];
  if (L.want <= 0) return;
  const int w = L.w;
  const uint8_t *im = L.im, *blur = L.blur;
  const int lane = threadIdx.x & 63;
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int n = L.counters[2];
  int base = 0;
  for (int l = 0; l < octave; ++l) base += lv.l[l].want > 0 ? lv.l[l].counters[2] : 0;
  if (i >= n || base + i >= cap_total) return;
  const int cx = L.out_xy[2 * i], cy = L.out_xy[2 * i + 1];
  int m10 = 0, m01 = 0;
  for (int p = lane; p < ORB_DISC; p += 64) {
    const int dx = disc[2 * p], dy = disc[2 * p + 1];
    const int v = im[(size_t)(cy + dy) * w + cx + dx];
    m10 += dx * v;
    m01 += dy * v;
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { m10 += __shfl_xor(m10, o); m01 += __shfl_xor(m01, o); }
  const float fx = (float)m10, fy = (float)m01;
  const float nrm = __fsqrt_rn(add_rn(mul_rn(fx, fx), mul_rn(fy, fy)));
  const float ca = nrm > 0.f ? __fdiv_rn(fx, nrm) : 1.f, sa = nrm > 0.f ? __fdiv_rn(fy, nrm) : 0.f;
  unsigned bits = 0;   // tests 4 lane .. 4 lane + 3
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float4 pt = reinterpret_cast<const float4 *>(pattern)[4 * lane + j];
    const int x1 = cx + (int)floorf(add_rn(add_rn(mul_rn(pt.x, ca), -mul_rn(pt.y, sa)), 0.5f)), y1 = cy + (int)floorf(add_rn(add_rn(mul_rn(pt.x, sa), mul_rn(pt.y, ca)), 0.5f));
    const int x2 = cx + (int)floorf(add_rn(add_rn(mul_rn(pt.z, ca), -mul_rn(pt.w, sa)), 0.5f)), y2 = cy + (int)floorf(add_rn(add_rn(mul_rn(pt.z, sa), mul_rn(pt.w, ca)), 0.5f));
    bits |= (blur[(size_t)y1 * w + x1] < blur[(size_t)y2 * w + x2] ? 1u : 0u) << j;
  }
  const unsigned hi = __shfl_down(bits, 1);   // byte m = tests 8 m .. 8 m + 7 = lanes 2 m (low nibble) and 2 m + 1
  if (!(lane & 1)) desc[(size_t)(base + i) * 32 + (lane >> 1)] = (uint8_t)(bits | (hi << 4));
  if (lane == 0) {
    OrbKeypoint k;
    k.x = mul_rn((float)cx, L.scale);
    k.y = mul_rn((float)cy, L.scale);
    k.angle = atan2f(sa, ca);
    k.response = (float)L.score[(size_t)cy * w + cx];
    k.octave = octave;
    kps[base + i] = k;
  }
}

}  // namespace spvo
