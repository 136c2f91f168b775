// match.hip.h -- K12/K13: brute-force 256-d L2 descriptor matching.
//
// Replaces cv::BFMatcher(NORM_L2)::match / knnMatch(k=2) + ratio test
// (reference: src/odml_visual_odometry/src/feature_detection_base.cpp:27-28,
// 462-491).  Structure:
//   K12a  S = A * B^T on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), one 64 x 128 tile per
//         workgroup, approximate squared distances dt = |a|^2 + |b|^2 - 2 S.  FUSED form (the default): the tile never
//         leaves the CU -- its epilogue reduces, per query row and in LDS, the tile's two smallest upper bounds dt + E and
//         keeps the few columns whose lower bound dt - E does not exceed the second of them ("per-row arg-min in LDS");
//         what goes to HBM is a handful of {column, lower bound} entries per (row, tile).  K12m then merges a row's <= 8
//         tile lists under the row-wide threshold and re-scores what is left canonically.  The unfused form (dt of EVERY
//         pair to HBM, 4 MB for 1000 x 1000, read back by K12b) serves the fp8 shortlist mode.
//   K12b / K12m  exact re-rank, one wave per query row.  |dt - d2| <= E = 2^-14 (|a|^2 + |b|^2) is a
//         rigorous bound on the distance between dt and the canonical fp32 distance d2 =
//         sum_k (a_k - b_k)^2 (sequential k, separately rounded multiply and add: bit-identical
//         to the oracle), derivation at MATCH_ERR_REL.  With U2 = the second smallest dt + E of
//         the row, every train row that can be among the two nearest under (d2, index) order
//         has dt - E <= U2: exactly those rows are re-scored with the canonical sum and the best
//         two picked under (distance, train index) order -- strict '<', so the lowest train
//         index wins ties, as BFMatcher does.  The number of re-scored rows adapts to the data
//         (2-4 on trained descriptors, the whole cluster on near-duplicate ones), so the result
//         is the brute-force result BY CONSTRUCTION: the MFMA only prunes.
//   K13   selector: NN (+ cv::batchDistance's crosscheck: train rows vote for their nearest query row) or KNN ratio test
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_mfma.hip.h"

namespace spvo {

constexpr int MATCH_D = 256;
constexpr int MATCH_QT = 64;     // query rows per workgroup (2 waves x 32)
constexpr int MATCH_TT = 128;    // train rows per workgroup (2 waves x 2 x 32)
constexpr int MATCH_C = 8;       // fused form: entries a (query row, column tile) may keep; a tile with more survivors (a cluster of
                                 // near-duplicates) is flagged by its count and re-scored column by column in the merge

// Error bound between the GEMM's dt and the canonical fp32 distance d2 (u = 2^-24, n = 256, N = |a|^2 + |b|^2, d^2 <= 2N):
//   canonical sum: |d2 - d^2| <= gamma(n + 2) d^2 <= 2 * 258 u N                                   = 516 u N
//   dot product on the matrix cores (an fma chain of length n): 2 |S~ - S| <= 2 * 256 u N / 2      = 256 u N
//   squared norms (any summation order, <= 64 roundings) and the two additions of the epilogue     <=  80 u N
// together 852 u N < 1024 u N = 2^-14 N.
constexpr float MATCH_ERR_REL = 6.103515625e-5f;   // 2^-14
// The fp8 shortlist GEMM (spvo_set_match_fp8) works in two passes.  Pass 1 uses this STATISTICAL window -- e4m3 operands carry
// 2^-4 relative rounding errors whose sum over 256 random-sign terms is ~5e-3 N; 2e-2 N covers four standard deviations -- and
// yields two canonical distances d0 <= d1.  Pass 2 makes the result exact: with a8 = the dequantised fp8 copy of a,
//   a.b - a8.b8 = (a - a8).b + a8.(b - b8),   |.| <= |a - a8| |b| + |a8| |b - b8|      (Cauchy-Schwarz; the four norms are
// computed per row when the copies are made), so |dt8 - d2| <= E8 = 2 (1 + 2^-10) (|a - a8| |b| + |a8| |b - b8|) + 2^-13 N
// RIGOROUSLY (the last term: fp32 accumulation of the exact fp8 products, the norms, the canonical sum itself); every column
// with dt8 - E8 <= d1 that pass 1 has not scored is scored canonically too.  On trained descriptors (nearest ~0.3, typical
// pair ~1.9, E8 ~0.15) that is a handful of columns per row; the result is the brute-force one by construction.
constexpr float MATCH_ERR_REL_FP8 = 2e-2f;
constexpr float MATCH_ERR_REL_FP8_TAIL = 1.220703125e-4f;   // 2^-13

// Row counts come either from the host (n_host) or, when the call is enqueued before the
// detector's counts are known on the host, from device memory (n_ptr).
__device__ __forceinline__ int dev_count(int n_host, const int *n_ptr) { return n_ptr ? *n_ptr : n_host; }

typedef float f32x4 __attribute__((ext_vector_type(4)));
struct MatchBest { float d0, d1; int i0, i1; };   // the two nearest so far under (distance, index) order
struct MatchTwo { float u1, u2; };                // two smallest values

struct MatchJob {
  const float *A, *B;             // [na][256], [nb][256]
  int na, nb;                     // host counts (upper bounds when the *_ptr are set)
  const int *na_ptr, *nb_ptr;     // device counts or NULL
  const float *nA, *nB;           // squared row norms
  float *dt;                      // [na][ldt] approximate squared distances (K12a -> K12b; unfused form only)
  int2 *cand;                     // [na][nt_stride][MATCH_C] {train row, float bits of dt - E}: a tile's survivors (fused form)
  int4 *meta;                     // [na][nt_stride] {survivor count (may exceed MATCH_C), float bits of the tile's two smallest dt + E, 0}
  float *best_d2;                 // [na][2]
  int *best_idx;                  // [na][2]
  const unsigned char *A8, *B8;   // fp8 (e4m3) copies [n][256] of A * 16 and B * 16 for the fp8 shortlist GEMM, or NULL
  const float2 *qA8, *qB8;        // per row {|x - x8|, |x8|} of those copies (x8 = dequantised), or NULL: the certificate of K12b's second pass
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
                                                          unsigned char *__restrict__ out, float2 *__restrict__ qnorm) {
  const int n = dev_count(n_host, n_ptr);
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int lane = threadIdx.x & 63;
  if (r >= n) return;
  const float4 v = *(const float4 *)(x + (size_t)r * MATCH_D + lane * 4);
  int pk = 0;
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(v.x * MATCH_FP8_SCALE, v.y * MATCH_FP8_SCALE, pk, false);
  pk = __builtin_amdgcn_cvt_pk_fp8_f32(v.z * MATCH_FP8_SCALE, v.w * MATCH_FP8_SCALE, pk, true);
  reinterpret_cast<int *>(out + (size_t)r * MATCH_D)[lane] = pk;
  // norms of the copy and of what the rounding took away (the copy read back exactly as the matrix cores will see it)
  const float inv = 1.f / MATCH_FP8_SCALE;
  const float c0 = __builtin_amdgcn_cvt_f32_fp8(pk, 0) * inv, c1 = __builtin_amdgcn_cvt_f32_fp8(pk, 1) * inv;
  const float c2 = __builtin_amdgcn_cvt_f32_fp8(pk, 2) * inv, c3 = __builtin_amdgcn_cvt_f32_fp8(pk, 3) * inv;
  const float e0 = v.x - c0, e1 = v.y - c1, e2 = v.z - c2, e3 = v.w - c3;
  float se = e0 * e0 + e1 * e1 + e2 * e2 + e3 * e3, sc = c0 * c0 + c1 * c1 + c2 * c2 + c3 * c3;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) { se += __shfl_xor(se, o); sc += __shfl_xor(sc, o); }
  if (lane == 0) qnorm[r] = make_float2(sqrtf(se), sqrtf(sc));
}

// K12a. grid = (ceil(nb/128), ceil(na/64), jobs): at 1000 x 1000 two jobs are 256 workgroups, one per CU.
// The K dimension is walked in slabs of 64.  A slab of the tile's 64 query and 128 train rows sits in LDS ROW-major with
// a pitch of 68 floats; the dot product does not care in which order k is visited as long as both operands agree, so
// the lane that supplies (row j, k-half h) of v_mfma_f32_32x32x2_f32 reads the 16 contiguous bytes k = 8m + 4h .. + 3
// of its row with ONE ds_read_b128 and feeds them to four consecutive matrix instructions.  (pitch / 4 = 17 is odd: the
// 16 lanes of each ds_read_b128 service group fall on 16 different bank quads.)  No transposing scalar LDS stores, 3
// 16-byte reads per 8 matrix instructions.  Staging goes global -> registers -> ds_write_b128; the loads of slab s + 1
// are issued before the matrix instructions of slab s, and the two LDS buffers alternate: one barrier per slab.
constexpr int MATCH_KS = 64;                       // K-slab
constexpr int MATCH_ROWS = MATCH_QT + MATCH_TT;    // 192
constexpr int MATCH_LD = MATCH_KS + 4;             // 68
constexpr int MATCH_BUF = MATCH_ROWS * MATCH_LD;   // floats per buffer
constexpr int MATCH_LDS_BYTES = 2 * MATCH_BUF * 4; // 104448
constexpr int MATCH_LDS_BYTES_FP8 = MATCH_QT * (MATCH_TT + 4) * 4;   // the fp8 variant stages only its output tile

__host__ __device__ inline int match_ldt(int nb_cap) { return (nb_cap + 31) & ~31; }   // row pitch of dt: 128-byte multiples

// FP8 = true: the dot products come from v_mfma_f32_32x32x16_fp8_fp8 on the fp8 copies (operands straight from global
// memory: a lane's 8 consecutive dimensions of one row are 8 contiguous bytes); dt is then approximate beyond
// MATCH_ERR_REL -- K12b runs with the statistical window MATCH_ERR_REL_FP8 and still produces exact distances for
// whatever falls into it -- so this mode is an opt-in (spvo_set_match_fp8).
template <bool FP8, bool FUSED = false>
__global__ __launch_bounds__(256) void match_gemm_kernel(MatchJobs jobs, int ldt, float err_rel, int nt_stride, int gx, int gy, int gz) {
  // Tile of this workgroup.  The grid is one-dimensional, 8 x chunk workgroups, chunk = ceil(tiles / 8): workgroups with equal
  // blockIdx.x % 8 share an XCD (MI355X_MICROARCH.md, workgroup dispatch) and take a CONTIGUOUS chunk of the tile order (job,
  // query tile, train tile): an XCD's L2 then holds a few query tiles and one job's train rows -- 10 MB leave the fabric for the
  // frame's two jobs instead of 18 (every XCD read every query row of both jobs with the (x, y, z) grid).  For speed only.
  const int n_tiles_all = gx * gy * gz, chunk = (n_tiles_all + 7) >> 3;
  const int tile_lin = (int)(blockIdx.x & 7) * chunk + (int)(blockIdx.x >> 3);
  if ((int)(blockIdx.x >> 3) >= chunk || tile_lin >= n_tiles_all) return;
  const int bx = tile_lin % gx, by = (tile_lin / gx) % gy, bz = tile_lin / (gx * gy);
  const MatchJob jb = jobs.j[bz];
  const float *__restrict__ A = jb.A;
  const float *__restrict__ B = jb.B;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, j = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int qh = wave >> 1, th = wave & 1;           // the wave's 32 x 64 piece of the 64 x 128 tile
  const int q0 = by * MATCH_QT, t0 = bx * MATCH_TT;
  const int na = dev_count(jb.na, jb.na_ptr), nb = dev_count(jb.nb, jb.nb_ptr);
  if (q0 >= na || t0 >= nb) return;

  f32x16 acc0, acc1;
#pragma unroll
  for (int r = 0; r < 16; ++r) { acc0[r] = 0.f; acc1[r] = 0.f; }
  // squared norms of the wave's rows: requested now, consumed by the epilogue
  float nav[16], nbv[2];
#pragma unroll
  for (int r = 0; r < 16; ++r) nav[r] = jb.nA[min(q0 + 32 * qh + (r & 3) + 8 * (r >> 2) + 4 * half, na - 1)];
#pragma unroll
  for (int f = 0; f < 2; ++f) nbv[f] = jb.nB[min(t0 + 64 * th + 32 * f + j, nb - 1)];

  if constexpr (FP8) {
    const int q = min(q0 + 32 * qh + j, na - 1);       // clamped rows are never stored
    const int ta = min(t0 + 64 * th + j, nb - 1), tb = min(t0 + 64 * th + 32 + j, nb - 1);
    const long *pa8 = reinterpret_cast<const long *>(jb.A8 + (size_t)q * MATCH_D) + half;
    const long *pb8 = reinterpret_cast<const long *>(jb.B8 + (size_t)ta * MATCH_D) + half;
    const long *pc8 = reinterpret_cast<const long *>(jb.B8 + (size_t)tb * MATCH_D) + half;
#pragma unroll
    for (int s = 0; s < MATCH_D / 16; ++s) {
      const long av = pa8[2 * s];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(av, pb8[2 * s], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(av, pc8[2 * s], acc1, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      acc0[r] *= 1.f / (MATCH_FP8_SCALE * MATCH_FP8_SCALE);
      acc1[r] *= 1.f / (MATCH_FP8_SCALE * MATCH_FP8_SCALE);
    }
  } else {
    constexpr int NLD = MATCH_ROWS * (MATCH_KS / 4) / 256;   // float4 per thread and slab: 12
    float4 pre[NLD];
    // 16 consecutive threads fetch the 256 bytes of one row's slab: a wave instruction covers 4 rows
    const int srow = tid >> 4, sc4 = tid & 15;
    auto load_slab = [&](int k0) {
#pragma unroll
      for (int it = 0; it < NLD; ++it) {
        const int row = it * 16 + srow;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (row < MATCH_QT) {
          if (q0 + row < na) v = *(const float4 *)(A + (size_t)(q0 + row) * MATCH_D + k0 + sc4 * 4);
        } else if (t0 + row - MATCH_QT < nb) {
          v = *(const float4 *)(B + (size_t)(t0 + row - MATCH_QT) * MATCH_D + k0 + sc4 * 4);
        }
        pre[it] = v;
      }
    };
    load_slab(0);
    const int off_a = (32 * qh + j) * MATCH_LD + 4 * half;
    const int off_b = (MATCH_QT + 64 * th + j) * MATCH_LD + 4 * half;
#pragma unroll 1
    for (int s = 0; s < MATCH_D / MATCH_KS; ++s) {
      float *buf = smem + (s & 1) * MATCH_BUF;
#pragma unroll
      for (int it = 0; it < NLD; ++it)
        *(float4 *)(buf + (it * 16 + srow) * MATCH_LD + sc4 * 4) = pre[it];
      __syncthreads();
      if (s + 1 < MATCH_D / MATCH_KS) load_slab((s + 1) * MATCH_KS);
      const float *pa = buf + off_a, *pb = buf + off_b;
      // operands of step m + 1 are read while the matrix instructions of step m run
      float4 av = *(const float4 *)pa, bv = *(const float4 *)pb, cv = *(const float4 *)(pb + 32 * MATCH_LD);
#pragma unroll
      for (int m = 0; m < MATCH_KS / 8; ++m) {
        float4 an = av, bn = bv, cn = cv;
        if (m + 1 < MATCH_KS / 8) {
          an = *(const float4 *)(pa + 8 * (m + 1));
          bn = *(const float4 *)(pb + 8 * (m + 1));
          cn = *(const float4 *)(pb + 32 * MATCH_LD + 8 * (m + 1));
        }
        __builtin_amdgcn_sched_barrier(0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, cv.x, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, cv.y, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv.z, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, cv.z, acc1, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv.w, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, cv.w, acc1, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        av = an; bv = bn; cv = cn;
      }
    }
  }
  // dt[q][t] = |a|^2 + |b|^2 - 2 S.  Register r of an accumulator holds query row (r & 3) + 8 (r >> 2) + 4 half and the
  // lanes of a half-wave 32 consecutive train rows.  32 dword stores per lane would be issue-bound, so the tile is
  // transposed through LDS (buffer 0 is free: its last readers passed the barrier of the final slab) and leaves as 8
  // 16-byte stores per lane, a wave instruction covering two 512-byte row pieces.  Columns beyond nb inside the row
  // pitch receive finite filler (zero-padded operands) that K12b never reads.
  constexpr int LDD = MATCH_TT + 4;
  float *sD = smem;
  {
    const float nba = nbv[0], nbb = nbv[1];
    float *w0 = sD + (32 * qh + 4 * half) * LDD + 64 * th + j;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2);
      w0[row * LDD] = nav[r] + nba - 2.f * acc0[r];
      w0[row * LDD + 32] = nav[r] + nbb - 2.f * acc1[r];
    }
  }
  if constexpr (FUSED) {
    // ---- per-row reduction of the tile in LDS: the tile's dt never reaches HBM.
    //   thread (row = tid / 4, part = tid % 4) owns the row's columns 16 i + 4 part .. + 3, i = 0..7 (8 ds_read_b128)
    //   pass 1: the row's two smallest upper bounds up = dt + E over the tile's columns (4 lanes combined by a butterfly)
    //   pass 2: columns with lo = dt - E <= (second smallest up) are the tile's survivors: the row's true nearest and second
    //           nearest are among the survivors of SOME tile whatever the other tiles hold, because the row-wide threshold
    //           K12m applies (second smallest up of the whole row) is <= this tile's.
    float *snb = sD + MATCH_QT * LDD;                            // [128] squared norms of the tile's train rows
    int *scnt = reinterpret_cast<int *>(snb + MATCH_TT);         // [64]  survivors per row
    int2 *slist = reinterpret_cast<int2 *>(scnt + MATCH_QT);     // [64][MATCH_C]
    if (tid < MATCH_TT) snb[tid] = jb.nB[min(t0 + tid, nb - 1)];
    if (tid < MATCH_QT) scnt[tid] = 0;
    __syncthreads();
    const int row = tid >> 2, part = tid & 3;
    const float naq = jb.nA[min(q0 + row, na - 1)];
    float lo[32];
    MatchTwo two{__builtin_inff(), __builtin_inff()};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const f32x4 d = *(const f32x4 *)(sD + row * LDD + 16 * i + 4 * part);
      const f32x4 nv = *(const f32x4 *)(snb + 16 * i + 4 * part);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = t0 + 16 * i + 4 * part + e < nb;
        const float err = err_rel * (naq + nv[e]);
        const float up = in ? d[e] + err : __builtin_inff();
        lo[4 * i + e] = in ? d[e] - err : __builtin_nanf("");   // a column beyond nb never survives: NaN <= x is false, also for x = +inf
        if (up < two.u1) { two.u2 = two.u1; two.u1 = up; } else if (up < two.u2) two.u2 = up;
      }
    }
#pragma unroll
    for (int o = 1; o < 4; o <<= 1) {
      const float o1 = __shfl_xor(two.u1, o), o2 = __shfl_xor(two.u2, o);
      const float hi = fmaxf(two.u1, o1);
      two.u1 = fminf(two.u1, o1);
      two.u2 = fminf(hi, fminf(two.u2, o2));
    }
#pragma unroll
    for (int k = 0; k < 32; ++k)
      if (lo[k] <= two.u2) {   // (u2 = +inf for a tile with a single column: that column survives, as it must)
        const int pos = atomicAdd(&scnt[row], 1);
        if (pos < MATCH_C) slist[row * MATCH_C + pos] = make_int2(t0 + 16 * (k >> 2) + 4 * part + (k & 3), __float_as_int(lo[k]));
      }
    __syncthreads();
    if (q0 + row < na) {
      const size_t slot = (size_t)(q0 + row) * nt_stride + bx;
      const int cnt = scnt[row];
      if (part == 0) jb.meta[slot] = make_int4(cnt, __float_as_int(two.u1), __float_as_int(two.u2), 0);
      if (2 * part < min(cnt, MATCH_C)) *reinterpret_cast<int4 *>(jb.cand + slot * MATCH_C + 2 * part) = *reinterpret_cast<const int4 *>(slist + row * MATCH_C + 2 * part);
      // More survivors than entries (near-duplicate descriptors: every column of the tile inside the window of its own second
      // smallest): this row's piece of the tile does go to HBM, 512 bytes, and K12m prunes it with the row-wide threshold
      // exactly as K12b would.  Trained descriptors never take this path (2-3 survivors per row).
      if (cnt > MATCH_C) {
        float *__restrict__ drow = jb.dt + (size_t)(q0 + row) * ldt + t0;
#pragma unroll
        for (int i = 0; i < 8; ++i)
          if (t0 + 16 * i + 4 * part < ldt) *(f32x4 *)(drow + 16 * i + 4 * part) = *(const f32x4 *)(sD + row * LDD + 16 * i + 4 * part);
      }
    }
    return;
  }
  __syncthreads();
  float *__restrict__ D = jb.dt;
  const int c4 = tid & 31, r0 = tid >> 5;
  const int t = t0 + c4 * 4;
  if (t < ldt) {
#pragma unroll
    for (int it = 0; it < MATCH_QT / 8; ++it) {
      const int row = it * 8 + r0;
      if (q0 + row < na) *(float4 *)(D + (size_t)(q0 + row) * ldt + t) = *(const float4 *)(sD + row * LDD + c4 * 4);
    }
  }
}

// K12b. One wave per query row.
//   pass 1: U2 = second smallest (dt + E) of the row, E = err_rel (|a|^2 + |b|^2)
//   pass 2: train rows with dt - E <= U2 are collected (ballot + prefix count into a per-wave LDS queue) and re-scored
//           in batches of up to 64 with the canonical sum; two rounds of wave arg-min under (distance, index) order
//           merge a batch into the best two.
// NCH > 0: the row has at most NCH * 256 columns and stays in registers between the passes (one round of loads, all in
// flight together); NCH = 0: any length, the row is read twice in chunks of 256 columns and the queue is drained as it
// fills (the threshold then also drops to the second-best canonical distance found so far).
// best[q] = {d2_0, d2_1 (f32 bits), idx0, idx1}
// Every mode but NN + cross-check applies the selector at the end and writes the packed result; with cross-check the
// best-of-train vote is cast here and match_select_cross_kernel finishes the job.
constexpr int MATCH_COOP = 8;                  // batches of up to 8 rows are fetched by the whole wave (one 1 KiB load per row)
constexpr int MATCH_COOP_LD = MATCH_D + 4;     // LDS pitch of those rows: lanes reading different rows hit different banks
template <int NCH> struct MatchRerankLds {
  static constexpr int LIST = NCH > 0 ? NCH * 256 : 512;   // queue entries per wave (NCH = 0: a ring, < 64 + 256 waiting)
  int list[4][LIST];
  __attribute__((aligned(16))) float a[4][MATCH_D];
  __attribute__((aligned(16))) float b[4][MATCH_COOP][MATCH_COOP_LD];
};

// Re-scores queue entries [head, head + n), n <= 64, and merges them into the best two.

template <int LIST>
__device__ __forceinline__ MatchBest match_score_batch(volatile int *list, int head, int n, const float *__restrict__ B,
                                                       const float4 *a4, float (*sb)[MATCH_COOP_LD], int lane, MatchBest best) {
  float d0 = best.d0, d1 = best.d1;
  int i0 = best.i0, i1 = best.i1;
  int idx = (lane < n) ? list[(head + lane) & (LIST - 1)] : -1;
  float s = 0.f;
  if (n <= MATCH_COOP) {
    // few rows (the usual case): the wave fetches each row with ONE coalesced 1 KiB load, all of them in flight together,
    // and parks them in LDS; lane c then sums row c -- one dependent chain of 256 additions, the canonical order
    float4 rowv[MATCH_COOP];
#pragma unroll
    for (int c = 0; c < MATCH_COOP; ++c) {
      rowv[c] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (c < n) {
        const int ic = __builtin_amdgcn_readfirstlane(list[(head + c) & (LIST - 1)]);
        rowv[c] = ((const float4 *)(B + (size_t)ic * MATCH_D))[lane];
      }
    }
#pragma unroll
    for (int c = 0; c < MATCH_COOP; ++c)
      if (c < n) *(float4 *)(&sb[c][lane * 4]) = rowv[c];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // the rows are read back by other lanes of this wave
    if (idx >= 0) {
      const float4 *b4 = (const float4 *)sb[lane];
#pragma unroll 8
      for (int i = 0; i < MATCH_D / 4; ++i) {
        const float4 av = a4[i];
        const float4 bv = b4[i];
        float t;
        t = sub_rn(av.x, bv.x); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.y, bv.y); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.z, bv.z); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.w, bv.w); s = add_rn(s, mul_rn(t, t));
      }
    }
  } else if (idx >= 0) {
    // many rows (clusters of near-duplicates): every lane streams its own row through two register blocks of 32 floats,
    // the loads of block i + 1 in flight while block i is summed
    const float4 *b4 = (const float4 *)(B + (size_t)idx * MATCH_D);
    float4 cur[8], nxt[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) cur[i] = b4[i];
#pragma unroll 1
    for (int blk = 0; blk < MATCH_D / 32; ++blk) {
      if (blk + 1 < MATCH_D / 32) {
#pragma unroll
        for (int i = 0; i < 8; ++i) nxt[i] = b4[(blk + 1) * 8 + i];
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const float4 av = a4[blk * 8 + i];
        const float4 bv = cur[i];
        float t;
        t = sub_rn(av.x, bv.x); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.y, bv.y); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.z, bv.z); s = add_rn(s, mul_rn(t, t));
        t = sub_rn(av.w, bv.w); s = add_rn(s, mul_rn(t, t));
      }
#pragma unroll
      for (int i = 0; i < 8; ++i) cur[i] = nxt[i];
    }
  }
  float d = (idx >= 0) ? s : __builtin_inff();
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
    if (mi == 0x7FFFFFFF) break;
    if (i0 < 0 || md < d0 || (md == d0 && mi < i0)) {
      d1 = d0; i1 = i0; d0 = md; i0 = mi;
    } else if (i1 < 0 || md < d1 || (md == d1 && mi < i1)) {
      d1 = md; i1 = mi;
    }
    if (idx == mi) { idx = -1; d = __builtin_inff(); }  // the winner leaves before the second round
  }
  return MatchBest{d0, d1, i0, i1};
}

__device__ __forceinline__ MatchTwo match_two_smallest(float up, MatchTwo t) {
  if (up < t.u1) { t.u2 = t.u1; t.u1 = up; } else if (up < t.u2) t.u2 = up;
  return t;
}

template <int NCH>
__global__ __launch_bounds__(256) void match_rerank_kernel(MatchJobs jobs, int ldt, float err_rel, int selector,
                                                           int cross_check, float ratio) {
  using L = MatchRerankLds<NCH>;
  constexpr int LIST = L::LIST;
  const MatchJob jb = jobs.j[blockIdx.y];
  const float *__restrict__ A = jb.A;
  const float *__restrict__ B = jb.B;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  L &lds = *reinterpret_cast<L *>(smem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.x * 4 + wave;
  const int lane = threadIdx.x & 63;
  const int na = dev_count(jb.na, jb.na_ptr), nb = dev_count(jb.nb, jb.nb_ptr);
  if (q >= na) return;
  const float *__restrict__ drow = jb.dt + (size_t)q * ldt;
  const float *__restrict__ nB = jb.nB;     // 16-byte aligned, readable up to the next multiple of 4 rows
  const float naq = jb.nA[q];
  volatile int *list = lds.list[wave];
  // the query row goes to LDS once (one coalesced 1 KiB load per wave); the re-score loops read it back with broadcast
  // ds_read_b128, which are counted apart from the global loads of the train rows they keep in flight
  ((float4 *)lds.a[wave])[lane] = ((const float4 *)(A + (size_t)q * MATCH_D))[lane];
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float4 *a4 = (const float4 *)lds.a[wave];
  MatchBest best{__builtin_inff(), __builtin_inff(), -1, -1};
  int head = 0, tail = 0;   // wave-uniform positions in the candidate queue

  // one chunk = 256 columns, 4 consecutive ones per lane; columns beyond nb never qualify
  auto load_chunk = [&](int t, f32x4 &dv, f32x4 &ev) __attribute__((always_inline)) {
    if (t < nb) { dv = *(const f32x4 *)(drow + t); ev = *(const f32x4 *)(nB + t); }
  };
  auto wave_second_smallest = [&](MatchTwo two) __attribute__((always_inline)) {
    float u1 = two.u1, u2 = two.u2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float o1 = __shfl_xor(u1, o), o2 = __shfl_xor(u2, o);
      const float hi = fmaxf(u1, o1);
      u1 = fminf(u1, o1);
      u2 = fminf(hi, fminf(u2, o2));
    }
    return u2;   // +inf when the row has fewer than two columns: everything is a candidate
  };

  if constexpr (NCH > 0) {
    f32x4 dv[NCH], ev[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      dv[c] = f32x4{0.f, 0.f, 0.f, 0.f}; ev[c] = dv[c];
      load_chunk(c * 256 + lane * 4, dv[c], ev[c]);
    }
    float lo[NCH][4];
    MatchTwo two{__builtin_inff(), __builtin_inff()};
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool in = c * 256 + lane * 4 + e < nb;
        const float err = err_rel * (naq + ev[c][e]);
        two = match_two_smallest(in ? dv[c][e] + err : __builtin_inff(), two);
        lo[c][e] = in ? dv[c][e] - err : __builtin_inff();
      }
    }
    const float thr = wave_second_smallest(two);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool cand = (c * 256 + lane * 4 + e < nb) && lo[c][e] <= thr;
        const unsigned long long mask = __ballot(cand);
        if (cand) list[tail + __popcll(mask & ((1ull << lane) - 1ull))] = c * 256 + lane * 4 + e;
        tail += __popcll(mask);
      }
    while (tail > head) {
      const int n = min(64, tail - head);
      best = match_score_batch<LIST>(list, head, n, B, a4, lds.b[wave], lane, best);
      head += n;
    }
  } else {
    MatchTwo two{__builtin_inff(), __builtin_inff()};
    for (int c0 = 0; c0 < nb; c0 += 256) {
      const int t = c0 + lane * 4;
      f32x4 dv = {0.f, 0.f, 0.f, 0.f}, ev = dv;
      load_chunk(t, dv, ev);
#pragma unroll
      for (int e = 0; e < 4; ++e) two = match_two_smallest(t + e < nb ? dv[e] + err_rel * (naq + ev[e]) : __builtin_inff(), two);
    }
    float thr = wave_second_smallest(two);
    const int nchunks = (nb + 255) >> 8;
    for (int step = 0; step <= nchunks; ++step) {
      const bool last = step == nchunks;
      if (!last) {
        const int t = step * 256 + lane * 4;
        f32x4 dv = {0.f, 0.f, 0.f, 0.f}, ev = dv;
        load_chunk(t, dv, ev);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const bool cand = (t + e < nb) && (dv[e] - err_rel * (naq + ev[e]) <= thr);
          const unsigned long long mask = __ballot(cand);
          if (cand) list[(tail + __popcll(mask & ((1ull << lane) - 1ull))) & (LIST - 1)] = t + e;
          tail += __popcll(mask);
        }
      }
      while (tail - head >= 64 || (last && tail > head)) {
        const int n = min(64, tail - head);
        best = match_score_batch<LIST>(list, head, n, B, a4, lds.b[wave], lane, best);
        head += n;
        if (best.i1 >= 0) thr = fminf(thr, best.d1);   // no row above the second-best canonical distance can enter any more
      }
    }
  }
  if (jb.qA8) {
    // fp8 shortlist, pass 2 (see MATCH_ERR_REL_FP8): every column whose rigorous lower bound dt8 - E8 does not exceed the second
    // canonical distance found so far.  A column pass 1 scored already may come again (it cannot win twice: it lost to the
    // current best two or is one of them, and those are skipped).
    const float2 qa = jb.qA8[q];
    float thr2 = best.i1 >= 0 ? best.d1 : __builtin_inff();
    head = tail = 0;
    for (int c0 = 0; c0 < nb; c0 += 256) {
      const int t = c0 + lane * 4;
      f32x4 dv = {0.f, 0.f, 0.f, 0.f}, ev = dv;
      float2 qb[4] = {};
      if (t < nb) {
        dv = *(const f32x4 *)(drow + t); ev = *(const f32x4 *)(nB + t);
#pragma unroll
        for (int e = 0; e < 4; ++e) qb[e] = jb.qB8[min(t + e, nb - 1)];
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int col = t + e;
        const float e8 = 2.001953125f * (qa.x * sqrtf(ev[e]) + qa.y * qb[e].x) + MATCH_ERR_REL_FP8_TAIL * (naq + ev[e]);
        const bool cand = col < nb && col != best.i0 && col != best.i1 && dv[e] - e8 <= thr2;
        const unsigned long long mask = __ballot(cand);
        if (cand) list[(tail + __popcll(mask & ((1ull << lane) - 1ull))) & (LIST - 1)] = col;
        tail += __popcll(mask);
      }
      while (tail - head >= 64 || (c0 + 256 >= nb && tail > head)) {
        const int n = min(64, tail - head);
        best = match_score_batch<LIST>(list, head, n, B, a4, lds.b[wave], lane, best);
        head += n;
        if (best.i1 >= 0) thr2 = fminf(thr2, best.d1);
      }
    }
  }

  const float d0 = best.d0, d1 = best.d1;
  const int i0 = best.i0, i1 = best.i1;
  if (lane == 0) {
    jb.best_d2[2 * q] = d0; jb.best_d2[2 * q + 1] = d1;
    jb.best_idx[2 * q] = i0; jb.best_idx[2 * q + 1] = i1;
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

// K12m.  The re-rank of the fused form: one wave per query row, fed by the tile lists K12a left behind instead of the row of dt.
//   threshold  U2 = second smallest of the tiles' {u1, u2} = second smallest dt + E of the whole row
//   candidates the stored entries with lo <= U2; a tile whose survivor count exceeds MATCH_C kept only some of them and left the
//              row's 128 dt values of that tile in HBM instead: they are pruned here with U2, as K12b prunes a whole row
// then exactly what K12b does: canonical re-score in batches, best two under (distance, index), selector / vote.
template <int UNUSED = 0>
__global__ __launch_bounds__(256) void match_merge_kernel(MatchJobs jobs, int nt_stride, int ldt, float err_rel, int selector, int cross_check, float ratio) {
  using L = MatchRerankLds<0>;
  constexpr int LIST = L::LIST;
  const MatchJob jb = jobs.j[blockIdx.y];
  const float *__restrict__ A = jb.A;
  const float *__restrict__ B = jb.B;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  L &lds = *reinterpret_cast<L *>(smem);
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int q = blockIdx.x * 4 + wave;
  const int lane = threadIdx.x & 63;
  const int na = dev_count(jb.na, jb.na_ptr), nb = dev_count(jb.nb, jb.nb_ptr);
  if (q >= na) return;
  volatile int *list = lds.list[wave];
  ((float4 *)lds.a[wave])[lane] = ((const float4 *)(A + (size_t)q * MATCH_D))[lane];
  const int ntiles = (nb + MATCH_TT - 1) / MATCH_TT;   // <= 64 (the host refuses larger capacities for this form)
  int cnt = 0;
  MatchTwo two{__builtin_inff(), __builtin_inff()};
  if (lane < ntiles) {
    const int4 m = jb.meta[(size_t)q * nt_stride + lane];
    cnt = m.x; two.u1 = __int_as_float(m.y); two.u2 = __int_as_float(m.z);
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  const float4 *a4 = (const float4 *)lds.a[wave];
  float thr;
  {
    float u1 = two.u1, u2 = two.u2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const float o1 = __shfl_xor(u1, o), o2 = __shfl_xor(u2, o);
      const float hi = fmaxf(u1, o1);
      u1 = fminf(u1, o1);
      u2 = fminf(hi, fminf(u2, o2));
    }
    thr = u2;   // +inf when the row has fewer than two columns: everything is a candidate
  }
  MatchBest best{__builtin_inff(), __builtin_inff(), -1, -1};
  int head = 0, tail = 0;
  auto push = [&](bool c, int col) __attribute__((always_inline)) {
    const unsigned long long mask = __ballot(c);
    if (c) list[(tail + __popcll(mask & ((1ull << lane) - 1ull))) & (LIST - 1)] = col;
    tail += __popcll(mask);
  };
  auto drain = [&](bool all) __attribute__((always_inline)) {
    while (tail - head >= 64 || (all && tail > head)) {
      const int n = min(64, tail - head);
      best = match_score_batch<LIST>(list, head, n, B, a4, lds.b[wave], lane, best);
      head += n;
    }
  };
  const int2 *__restrict__ crow = jb.cand + (size_t)q * nt_stride * MATCH_C;
  for (int base = 0; base < ntiles * MATCH_C; base += 64) {
    const int s = base + lane, tile = s / MATCH_C, c = s % MATCH_C;
    const int cnt_t = __shfl(cnt, tile & 63);
    const bool valid = s < ntiles * MATCH_C && cnt_t <= MATCH_C && c < cnt_t;
    int2 e = make_int2(0, 0);
    if (valid) e = crow[s];
    push(valid && __int_as_float(e.y) <= thr, e.x);
    drain(false);
  }
  unsigned long long over = __ballot(lane < ntiles && cnt > MATCH_C);
  if (over) {
    const float *__restrict__ drow = jb.dt + (size_t)q * ldt;
    const float naq = jb.nA[q];
    while (over) {
      const int t = __builtin_ctzll(over);
      over &= over - 1;
#pragma unroll 1
      for (int h = 0; h < MATCH_TT / 64; ++h) {
        const int col = t * MATCH_TT + h * 64 + lane;
        const bool in = col < nb;
        const float dv = in ? drow[col] : 0.f, nv = in ? jb.nB[col] : 0.f;
        push(in && dv - err_rel * (naq + nv) <= thr, col);
        drain(false);
      }
    }
  }
  drain(true);

  const float d0 = best.d0, d1 = best.d1;
  const int i0 = best.i0, i1 = best.i1;
  if (lane == 0) {
    jb.best_d2[2 * q] = d0; jb.best_d2[2 * q + 1] = d1;
    jb.best_idx[2 * q] = i0; jb.best_idx[2 * q + 1] = i1;
    const float s0 = sqrtf(d0), s1 = sqrtf(d1);   // BFMatcher L2 returns sqrt(sum of squares)
    if (selector == 0 && cross_check) {
      if (i0 >= 0) atomicMin(&jb.train_best[i0], ((unsigned long long)__float_as_uint(s0) << 32) | (unsigned)q);   // see K12b
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

// ---------------------------------------------------------------------------
// K12h: cv::BFMatcher(NORM_HAMMING) for the binary descriptors of the classic front end (ORB / BRISK / AKAZE: initMatcher,
// base.cpp:17-21; matchDescriptors, base.cpp:434-500 -- 49 ms per frame for 2000 ORB keypoints in the reference's own table,
// VO/README.md:25).  Integer work: distance = popcount(a xor b), exact by nature, so ONE pass: one wave per query row, every
// lane scans the train rows lane, lane + 64, ... (a row is NW 32-bit words, zero-padded; 64 consecutive rows are one contiguous
// 2 KB read), keeps its two best under the (distance, index) order of a strict '<' scan, and the wave merges the 64 pairs.
//   mode 0  NN:  idx = nearest, dist = its distance
//   mode 1  KNN: dist = nearest distance, idx = nearest if d0 < ratio * d1 (float32, base.cpp:469), else -1
//   mode 2  the vote of cv::batchDistance's crosscheck: the rows of A are the TRAIN rows, B the queries; each train row casts
//           (distance << 32 | own index) into vote[its nearest query] with an atomic min -- the query then keeps the nearest
//           train row that chose it, the lowest one on ties (match_hamming_cross_kernel), as in oracle/matching.py.
// ---------------------------------------------------------------------------
struct HamPair { int d0, i0, d1, i1; };
__device__ __forceinline__ bool ham_less(int da, int ia, int db, int ib) { return da < db || (da == db && (unsigned)ia < (unsigned)ib); }
__device__ __forceinline__ void ham_insert(HamPair &p, int d, int i) {
  if (ham_less(d, i, p.d0, p.i0)) { p.d1 = p.d0; p.i1 = p.i0; p.d0 = d; p.i0 = i; }
  else if (ham_less(d, i, p.d1, p.i1)) { p.d1 = d; p.i1 = i; }
}

template <int NW>
__global__ __launch_bounds__(256) void match_hamming_kernel(const uint32_t *__restrict__ A, int na, const uint32_t *__restrict__ B, int nb, int mode, float ratio,
                                                            int *__restrict__ out_idx, float *__restrict__ out_dist, unsigned long long *__restrict__ vote) {
  const int lane = threadIdx.x & 63;
  const int q = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (q >= na) return;
  uint32_t a[NW];
#pragma unroll
  for (int w = 0; w < NW; ++w) a[w] = A[(size_t)q * NW + w];
  HamPair p{0x7FFFFFFF, -1, 0x7FFFFFFF, -1};
  for (int t = lane; t < nb; t += 64) {
    const uint4 *row = reinterpret_cast<const uint4 *>(B + (size_t)t * NW);
    int d = 0;
#pragma unroll
    for (int w4 = 0; w4 < NW / 4; ++w4) {
      const uint4 v = row[w4];
      d += __popc(v.x ^ a[4 * w4]) + __popc(v.y ^ a[4 * w4 + 1]) + __popc(v.z ^ a[4 * w4 + 2]) + __popc(v.w ^ a[4 * w4 + 3]);
    }
    ham_insert(p, d, t);
  }
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) {
    const int od0 = __shfl_xor(p.d0, o), oi0 = __shfl_xor(p.i0, o), od1 = __shfl_xor(p.d1, o), oi1 = __shfl_xor(p.i1, o);
    if (oi0 >= 0) ham_insert(p, od0, oi0);
    if (oi1 >= 0) ham_insert(p, od1, oi1);
  }
  if (lane) return;
  if (mode == 2) {
    if (p.i0 >= 0) atomicMin(&vote[p.i0], ((unsigned long long)(unsigned)p.d0 << 32) | (unsigned)q);
    return;
  }
  out_dist[q] = p.i0 >= 0 ? (float)p.d0 : 0.f;
  if (mode == 0) out_idx[q] = p.i0;
  else out_idx[q] = (p.i1 >= 0 && (float)p.d0 < ratio * (float)p.d1) ? p.i0 : -1;
}

__global__ __launch_bounds__(256) void match_hamming_cross_kernel(const unsigned long long *__restrict__ vote, int na, int *__restrict__ out_idx, float *__restrict__ out_dist) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q >= na) return;
  const unsigned long long key = vote[q];
  const bool hit = key != ~0ull;
  out_idx[q] = hit ? (int)(key & 0xFFFFFFFFull) : -1;
  out_dist[q] = hit ? (float)(unsigned)(key >> 32) : 0.f;
}

}  // namespace spvo
