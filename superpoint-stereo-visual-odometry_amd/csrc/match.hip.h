// match.hip.h -- K12/K13: brute-force 256-d L2 descriptor matching.
//
// Replaces cv::BFMatcher(NORM_L2)::match / knnMatch(k=2) + ratio test
// (reference: src/odml_visual_odometry/src/feature_detection_base.cpp:27-28,
// 462-491).  Structure:
//   K12a  S = A * B^T on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), tile
//         32 queries x 128 train rows per workgroup, approximate
//         d2 = |a|^2 + |b|^2 - 2 S, per-(query, 128-column group) top-4 kept in LDS
//   K12b  exact re-rank: one wave per query recomputes the canonical distance
//         sum_k (a_k - b_k)^2 (sequential k, separate multiply and add roundings,
//         i.e. bit-identical to the oracle) for the <= 4*groups shortlisted rows
//         and picks the best two under (distance, train index) order -- strict
//         '<' so the lowest train index wins ties, as BFMatcher does
//   K13   selector: NN (+ cv::batchDistance's crosscheck: train rows vote for their nearest query row) or KNN ratio test
// Integer outputs (train indices) are therefore independent of the MFMA
// rounding; the MFMA only prunes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_mfma.hip.h"

namespace spvo {

constexpr int MATCH_D = 256;
constexpr int MATCH_QT = 32;     // queries per workgroup
constexpr int MATCH_TT = 128;    // train rows per workgroup (4 waves x 32)
constexpr int MATCH_KEEP = 4;    // shortlist per (query, column group)

// Row counts come either from the host (n_host) or, when the call is enqueued before the
// detector's counts are known on the host, from device memory (n_ptr).
__device__ __forceinline__ int dev_count(int n_host, const int *n_ptr) { return n_ptr ? *n_ptr : n_host; }

struct MatchJob {
  const float *A, *B;             // [na][256], [nb][256]
  int na, nb;                     // host counts (upper bounds when the *_ptr are set)
  const int *na_ptr, *nb_ptr;     // device counts or NULL
  const float *nA, *nB;           // squared row norms
  int *shortlist;                 // [na][groups][MATCH_KEEP]
  float *best_d2;                 // [na][2]
  int *best_idx;                  // [na][2]
  const unsigned char *A8, *B8;   // fp8 (e4m3) copies [n][256] of A * 16 and B * 16 for the fp8 shortlist GEMM, or NULL
  unsigned long long *train_best; // [nb] (cross-check only: the sides are swapped, nb = query rows; {distance bits, train row})
  int2 *out;                      // [na] packed {train_idx, float bits of the distance} ([nb] with cross-check)
};
struct MatchJobs { MatchJob j[2]; };   // blockIdx.z selects the job (stereo / temporal match)

__global__ __launch_bounds__(256) void row_sqnorm_kernel(const float *__restrict__ x, int n_host,
                                                         const int *__restrict__ n_ptr,
                                                         float *__restrict__ out) {
  const int n = dev_count(n_host, n_ptr);
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const float4 v = *(const float4 *)(x + (size_t)r * MATCH_D + lane * 4);
  float s = v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) s += __shfl_xor(s, o);
  if (lane == 0) out[r] = s;
}

// fp8 copy of descriptor rows for the fp8 shortlist GEMM (BASELINE config 5): x * 16 rounded to OCP e4m3 (unit-norm
// descriptors have entries around 1/16, so the scale puts them in the format's normal range).  One wave per row.
constexpr float MATCH_FP8_SCALE = 16.f;
__global__ __launch_bounds__(256) void desc_to_fp8_kernel(const float *__restrict__ x, int n_host, const int *__restrict__ n_ptr,
                                                          unsigned char *__restrict__ out) {
  const int n = dev_count(n_host, n_ptr);
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const float4 v = *(const float4 *)(x + (size_t)r * MATCH_D + lane * 4);
  int pk = 0;
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * MATCH_FP8_SCALE, v.y * MATCH_FP8_SCALE, pk, false);
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * MATCH_FP8_SCALE, v.w * MATCH_FP8_SCALE, pk, true);
  reinterpret_cast<int *>(out + (size_t)r * MATCH_D)[lane] = pk;
}

// K12a. grid = (ceil(nb/128), ceil(na/32), jobs).  shortlist[q][group][MATCH_KEEP] (train idx, -1 = none)
// LDS holds one K-slab (64 of the 256 dimensions) of the 32 query rows and the 128 train rows,
// K-MAJOR: element (row r, dim k) at [k][r] with a row pitch of 161 floats, so that both MFMA
// operand reads are consecutive-lane ds_read_b32 (conflict-free) while the staging loads stay
// coalesced along k.  41 KB per workgroup -> three workgroups per CU overlap load and compute.
constexpr int MATCH_KS = 64;                       // K-slab
constexpr int MATCH_ROWS = MATCH_QT + MATCH_TT;    // 160
constexpr int MATCH_LD = MATCH_ROWS + 1;           // 161
constexpr int MATCH_LDS_BYTES = MATCH_KS * MATCH_LD * 4;

// FP8 = true: the dot products of the shortlist come from v_mfma_f32_32x32x16_fp8_fp8 on the fp8 copies (operands straight
// from global memory: a lane's 8 consecutive dimensions of one row are 8 contiguous bytes); the shortlist is then
// APPROXIMATE -- the exact re-rank (K12b) still produces exact distances for whatever it contains -- so this mode is an
// opt-in (spvo_set_match_fp8) and the default keeps the fp32 GEMM whose shortlist error is 1e-7.
template <bool FP8>
__global__ __launch_bounds__(256) void match_gemm_kernel(MatchJobs jobs, int groups) {
  const MatchJob jb = jobs.j[blockIdx.z];
  const float *__restrict__ A = jb.A;
  const float *__restrict__ B = jb.B;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int q0 = blockIdx.y * MATCH_QT, t0 = blockIdx.x * MATCH_TT;
  const int na = dev_count(jb.na, jb.na_ptr), nb = dev_count(jb.nb, jb.nb_ptr);
  if (q0 >= na) return;
  if (t0 >= nb) {   // empty column group: the shortlist must still say "none"
    if (tid < MATCH_QT && q0 + tid < na) {
      int *o = jb.shortlist + ((size_t)(q0 + tid) * groups + blockIdx.x) * MATCH_KEEP;
#pragma unroll
      for (int k = 0; k < MATCH_KEEP; ++k) o[k] = -1;
    }
    return;
  }

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.f;

  if constexpr (FP8) {
    const int q = min(q0 + j, na - 1), t = min(t0 + wave * 32 + j, nb - 1);       // clamped rows are masked out below
    const long *pa8 = reinterpret_cast<const long *>(jb.A8 + (size_t)q * MATCH_D) + half;
    const long *pb8 = reinterpret_cast<const long *>(jb.B8 + (size_t)t * MATCH_D) + half;
#pragma unroll
    for (int s = 0; s < MATCH_D / 16; ++s)
      acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(pa8[2 * s], pb8[2 * s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] *= 1.f / (MATCH_FP8_SCALE * MATCH_FP8_SCALE);
  }
  // K slabs of 64: the global loads of slab s+1 are issued before the matrix instructions of slab s (registers), so only the
  // first slab's latency is exposed
  constexpr int NLD = MATCH_ROWS * (MATCH_KS / 4) / 256;   // float4 per thread and slab: 10
  float4 pre[NLD];
  auto load_slab = [&](int k0) {
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int i = it * 256 + tid;
      const int row = i >> 4, c4 = i & 15;
      float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
      if (row < MATCH_QT) {
        if (q0 + row < na) v = *(const float4 *)(A + (size_t)(q0 + row) * MATCH_D + k0 + c4 * 4);
      } else if (t0 + row - MATCH_QT < nb) {
        v = *(const float4 *)(B + (size_t)(t0 + row - MATCH_QT) * MATCH_D + k0 + c4 * 4);
      }
      pre[it] = v;
    }
  };
  if constexpr (!FP8) load_slab(0);
  for (int k0 = 0; k0 < (FP8 ? 0 : MATCH_D); k0 += MATCH_KS) {
    __syncthreads();
    // 160 rows x 16 float4: consecutive threads walk along k (coalesced 256-byte runs)
#pragma unroll
    for (int it = 0; it < NLD; ++it) {
      const int i = it * 256 + tid;
      const int row = i >> 4, c4 = i & 15;
      const float4 v = pre[it];
      float *dst = smem + (c4 * 4) * MATCH_LD + row;
      dst[0] = v.x; dst[MATCH_LD] = v.y; dst[2 * MATCH_LD] = v.z; dst[3 * MATCH_LD] = v.w;
    }
    __syncthreads();
    if (k0 + MATCH_KS < MATCH_D) load_slab(k0 + MATCH_KS);
    // D[i = query][jj = train]: A operand lane -> A[q = j][k = 2s + half], B operand -> B[t = j][k]
    const float *pa = smem + half * MATCH_LD + j;
    const float *pb = smem + half * MATCH_LD + MATCH_QT + wave * 32 + j;
#pragma unroll 8
    for (int s2 = 0; s2 < MATCH_KS / 2; ++s2)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[2 * s2 * MATCH_LD], pb[2 * s2 * MATCH_LD], acc, 0, 0, 0);
  }
  __syncthreads();
  // approximate squared distances -> LDS [32 q][128 t + 1]
  float *sD = smem;
  constexpr int LDD = MATCH_TT + 1;
  {
    const int t = t0 + wave * 32 + j;
    const float nbv = (t < nb) ? jb.nB[t] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int q = (r & 3) + 8 * (r >> 2) + 4 * half;
      const float nav = (q0 + q < na) ? jb.nA[q0 + q] : 0.f;
      sD[q * LDD + wave * 32 + j] = (t < nb) ? (nav + nbv - 2.f * acc[r]) : __builtin_inff();
    }
  }
  __syncthreads();
  // Per query the MATCH_KEEP smallest (distance, index) pairs of the 128 columns: 8 lanes hold 16 columns each in registers and
  // run MATCH_KEEP selection rounds -- lane-local minimum (the lower column wins ties), 3-step butterfly across the 8 lanes
  // under (distance, index) order (BFMatcher's scan order), the owner masks the winner out.  (Sorted insertion of every
  // column followed by a serial merge of the 8 lists cost 16 of the kernel's 30 us.)
  {
    const int q = tid >> 3, g = tid & 7;
    const int ncol = min(MATCH_TT, nb - t0);
    float dcol[MATCH_TT / 8];
#pragma unroll
    for (int i = 0; i < MATCH_TT / 8; ++i) {
      const int c = g + 8 * i;
      dcol[i] = (c < ncol) ? sD[q * LDD + c] : __builtin_inff();
    }
    int sel[MATCH_KEEP];
#pragma unroll
    for (int r = 0; r < MATCH_KEEP; ++r) {
      float bd = __builtin_inff();
      int bi = 0x7FFFFFFF;
#pragma unroll
      for (int i = 0; i < MATCH_TT / 8; ++i)
        if (dcol[i] < bd) { bd = dcol[i]; bi = t0 + g + 8 * i; }
#pragma unroll
      for (int m = 1; m < 8; m <<= 1) {
        const float od = __shfl_xor(bd, m);
        const int oi = __shfl_xor(bi, m);
        if (od < bd || (od == bd && (unsigned)oi < (unsigned)bi)) { bd = od; bi = oi; }
      }
      sel[r] = bi;
#pragma unroll
      for (int i = 0; i < MATCH_TT / 8; ++i)
        if (t0 + g + 8 * i == bi) dcol[i] = __builtin_inff();
    }
    if (g == 0 && q0 + q < na) {
      int *o = jb.shortlist + ((size_t)(q0 + q) * groups + blockIdx.x) * MATCH_KEEP;
#pragma unroll
      for (int k = 0; k < MATCH_KEEP; ++k) o[k] = (sel[k] == 0x7FFFFFFF) ? -1 : sel[k];
    }
  }
}

// K12b. One wave per query; lane c re-scores shortlisted candidate c exactly.
// Handles groups*MATCH_KEEP candidates in passes of 64.
// best[q] = {d2_0, d2_1 (f32 bits), idx0, idx1}
// When `select_here` is set (every mode but NN + cross-check) the selector of K13 runs at the end of
// this kernel and the packed result is written directly; with cross-check the best-of-train scatter
// happens here and match_select_kernel finishes the job.
__global__ __launch_bounds__(256) void match_rerank_kernel(MatchJobs jobs, int groups, int selector,
                                                           int cross_check, float ratio) {
  const MatchJob jb = jobs.j[blockIdx.y];
  const float *__restrict__ A = jb.A;
  const float *__restrict__ B = jb.B;
  const int *__restrict__ shortlist = jb.shortlist;
  float *__restrict__ best_d2 = jb.best_d2;
  int *__restrict__ best_idx = jb.best_idx;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  const int na = dev_count(jb.na, jb.na_ptr);
  if (q >= na) return;
  const int ncand = groups * MATCH_KEEP;
  const float *a = A + (size_t)q * MATCH_D;
  float d0 = __builtin_inff(), d1 = __builtin_inff();
  int i0 = -1, i1 = -1;
  for (int base = 0; base < ncand; base += 64) {
    const int c = base + lane;
    const int idx = (c < ncand) ? shortlist[(size_t)q * ncand + c] : -1;
    float d = __builtin_inff();
    if (idx >= 0) {
      const float *b = B + (size_t)idx * MATCH_D;
      float s = 0.f;
      for (int k = 0; k < MATCH_D; k += 4) {
        const float4 av = *(const float4 *)(a + k);
        const float4 bv = *(const float4 *)(b + k);
        float t;
        t = sub_rn(av.x, bv.x); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.y, bv.y); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.z, bv.z); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.w, bv.w); s = add_rn(s, mul_rn(t, t));
      }
      d = s;
    }
    // two rounds of wave arg-min under (d, idx) order
#pragma unroll
    for (int round = 0; round < 2; ++round) {
      float md = d;
      int mi = (idx >= 0) ? idx : 0x7FFFFFFF;
#pragma unroll
      for (int o = 32; o >= 1; o >>= 1) {
        const float od = __shfl_xor(md, o);
        const int oi = __shfl_xor(mi, o);
        if (od < md || (od == md && oi < mi)) { md = od; mi = oi; }
      }
      if (mi == 0x7FFFFFFF || md == __builtin_inff()) break;
      // merge (md, mi) into the running best two
      if (md < d0 || (md == d0 && mi < i0) || i0 < 0) {
        d1 = d0; i1 = i0; d0 = md; i0 = mi;
      } else if (md < d1 || (md == d1 && mi < i1) || i1 < 0) {
        d1 = md; i1 = mi;
      }
      if (idx == mi) d = __builtin_inff();  // remove the winner for the second round
    }
  }
  if (lane == 0) {
    best_d2[2 * q] = d0; best_d2[2 * q + 1] = d1;
    best_idx[2 * q] = i0; best_idx[2 * q + 1] = i1;
    const float s0 = sqrtf(d0), s1 = sqrtf(d1);   // BFMatcher L2 returns sqrt(sum of squares)
    if (selector == 0 && cross_check) {
      // sides swapped by the host: row q is a TRAIN row, i0 its nearest query row (lowest index on ties).  The query
      // keeps the nearest of the train rows that chose it, the lowest train row on ties (cv::batchDistance: `d < d0`
      // while the train index runs upwards) = the minimum of {distance bits, train row}
      if (i0 >= 0) atomicMin(&jb.train_best[i0], ((unsigned long long)__float_as_uint(s0) << 32) | (unsigned)q);
    } else {
      int out = -1;
      if (selector == 0) out = i0;
      else if (i0 >= 0 && i1 >= 0 && s0 < mul_rn(ratio, s1)) out = i0;   // base.cpp:469
      jb.out[q] = make_int2(out, __float_as_int(s0));
    }
  }
}

// OpenCV crossCheck (cv::batchDistance, crosscheck = true, K = 1, as BFMatcher::knnMatchImpl calls it): every train
// row votes for its nearest query row; a query row is matched to the nearest train row among its voters and stays
// unmatched when nobody voted for it.  The vote (atomicMin on {distance bits, train row}) is cast by
// match_rerank_kernel, which ran with the two sides swapped; this kernel reads it out per query row.
__global__ __launch_bounds__(256) void match_select_cross_kernel(MatchJobs jobs) {
  const MatchJob jb = jobs.j[blockIdx.y];
  const int nq = dev_count(jb.nb, jb.nb_ptr);   // swapped: the B side holds the query rows
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= nq) return;
  const unsigned long long key = jb.train_best[q];
  const bool hit = key != ~0ull;
  jb.out[q] = make_int2(hit ? (int)(key & 0xFFFFFFFFull) : -1, hit ? (int)(key >> 32) : 0);
}

}  // namespace spvo
