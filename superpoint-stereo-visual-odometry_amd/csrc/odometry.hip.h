// odometry.hip.h -- K14 (stereo triangulation), K15 (deterministic PnP-RANSAC)
// and K16 (PnP refinement: Levenberg-Marquardt on CostFunctor32 blocks), all f64.
//
// Replaces, in FeatureFrontEnd::solveStereoOdometry
// (reference: src/odml_visual_odometry/src/feature_detection_base.cpp:125-399):
//   cv::triangulatePoints + convertPointsFromHomogeneous      base.cpp:211-223
//   cv::solvePnPRansac(..., 500, 2.0, 0.999, USAC_ACCURATE)   base.cpp:237-239
//   ceres::Solve on CostFunctor32 blocks, HuberLoss(1.0)      base.cpp:282-375,
//       include/odml_visual_odometry/ceres_cost_function.hpp:27-58
// These are latency-bound kernels (a few thousand residuals): wave/block
// reductions, no MFMA.  Algorithms are stated in oracle/odometry.py.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "spvo_types.hip.h"
#include "conv_mfma.hip.h"   // mul_rn: separately rounded products

namespace spvo {

// ------------------------------------------------------------- small f64 helpers
__device__ __forceinline__ void quat_to_rot(const double *q, double *R) {
  // Eigen::Quaternion::toRotationMatrix (no normalisation)
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
  const double twx = tx * w, twy = ty * w, twz = tz * w;
  const double txx = tx * x, txy = ty * x, txz = tz * x;
  const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}

__device__ __forceinline__ void quat_mul(const double *a, const double *b, double *o) {
  const double ax = a[0], ay = a[1], az = a[2], aw = a[3];
  const double bx = b[0], by = b[1], bz = b[2], bw = b[3];
  o[0] = aw * bx + ax * bw + ay * bz - az * by;
  o[1] = aw * by - ax * bz + ay * bw + az * bx;
  o[2] = aw * bz + ax * by - ay * bx + az * bw;
  o[3] = aw * bw - ax * bx - ay * by - az * bz;
}

__device__ inline void rvec_to_quat(const double *r, double *q) {
  // base.cpp:274-278: AngleAxisd(|r|, r.normalized()) -> Quaterniond
  const double angle = sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  double ax = r[0], ay = r[1], az = r[2];
  if (angle > 0) { ax /= angle; ay /= angle; az /= angle; }
  const double s = sin(angle / 2);
  q[0] = ax * s; q[1] = ay * s; q[2] = az * s; q[3] = cos(angle / 2);
}

__device__ inline void quat_to_rvec(const double *qin, double *r) {
  double x = qin[0], y = qin[1], z = qin[2], w = qin[3];
  if (w < 0) { x = -x; y = -y; z = -z; w = -w; }
  const double n = sqrt(x * x + y * y + z * z);
  if (n < 1e-300) { r[0] = r[1] = r[2] = 0; return; }
  const double angle = 2 * atan2(n, w);
  r[0] = x / n * angle; r[1] = y / n * angle; r[2] = z / n * angle;
}

// Gaussian elimination with partial pivoting, N x N, A row-major, solves A x = b in place.
template <int N>
__device__ inline bool solve_linear(double *A, double *b) {
  for (int c = 0; c < N; ++c) {
    int piv = c;
    double best = fabs(A[c * N + c]);
    for (int r = c + 1; r < N; ++r) {
      const double v = fabs(A[r * N + c]);
      if (v > best) { best = v; piv = r; }
    }
    if (!(best > 1e-300) || !isfinite(best)) return false;
    if (piv != c) {
      for (int k = 0; k < N; ++k) { const double tmp = A[c * N + k]; A[c * N + k] = A[piv * N + k]; A[piv * N + k] = tmp; }
      const double tb = b[c]; b[c] = b[piv]; b[piv] = tb;
    }
    const double inv = 1.0 / A[c * N + c];
    for (int r = c + 1; r < N; ++r) {
      const double f = A[r * N + c] * inv;
      if (f != 0) {
        for (int k = c; k < N; ++k) A[r * N + k] -= f * A[c * N + k];
        b[r] -= f * b[c];
      }
    }
  }
  for (int r = N - 1; r >= 0; --r) {
    double s = b[r];
    for (int k = r + 1; k < N; ++k) s -= A[r * N + k] * b[k];
    b[r] = s / A[r * N + r];
  }
  for (int k = 0; k < N; ++k)
    if (!isfinite(b[k])) return false;
  return true;
}

// ------------------------------------------------------------------------- K14
// cv::triangulatePoints + convertPointsFromHomogeneous (base.cpp:211-223): the homogeneous point is the right-singular vector of
// the smallest singular value of the 4x4 DLT system A (rows x P[2] - P[0], y P[2] - P[1] of both views), then x / w in f32.
// One thread per point, f64.  That vector is the eigenvector of the smallest eigenvalue of B = A^T A, and it is computed as
// such: B + mu I = L D L^T (mu = 1e-14 trace(B): the pivots stay positive when the system is exactly rank 3, as noise-free
// synthetic data make it), then inverse iteration from e_w -- the wanted vector's w component is what a finite point never
// lacks -- which contracts by (sigma_4 / sigma_3)^2 ~ 1e-4 .. 1e-8 per step; 4 steps.  ~300 f64 instructions, 8 divisions, no
// square root in the loop.  The one-sided Jacobi SVD this replaces (30 sweeps at most, 3 square roots and 3 divisions per
// rotation: ~10 k f64 instructions, 58 us inside the pipeline) gave the same vector; with the SVD's unit 2-norm scaling (below) the
// stored f32 points are the oracle's bit for bit on >= 98 % of the points and never more than two f32 units away (tests/test_gpu_odometry.py).  A^T A squares the condition number:
// the direction error is ~ eps (sigma_1 / sigma_3)^2 <= 1e-16 x 1e8, far below the f32 rounding of the stored point.
// one correspondence: (x0, y0) in the left, (x1, y1) in the right image -> out[0..2]
__device__ __forceinline__ void triangulate_point(const double *Pl, const double *Pr, const double x0, const double y0, const double x1, const double y1, float *out) {
  double A[16];
  {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      A[0 * 4 + k] = x0 * Pl[8 + k] - Pl[0 + k];
      A[1 * 4 + k] = y0 * Pl[8 + k] - Pl[4 + k];
      A[2 * 4 + k] = x1 * Pr[8 + k] - Pr[0 + k];
      A[3 * 4 + k] = y1 * Pr[8 + k] - Pr[4 + k];
    }
  }
  double B[4][4];
#pragma unroll
  for (int p = 0; p < 4; ++p)
#pragma unroll
    for (int q = p; q < 4; ++q) {
      double sum = 0;
#pragma unroll
      for (int r = 0; r < 4; ++r) sum += A[r * 4 + p] * A[r * 4 + q];
      B[p][q] = sum;
    }
  const double mu = 1e-14 * (B[0][0] + B[1][1] + B[2][2] + B[3][3]) + 1e-300;
  // L D L^T (unit lower triangle l, diagonal d kept as reciprocals)
  const double d0 = B[0][0] + mu, i0 = 1.0 / d0;
  const double l10 = B[0][1] * i0, l20 = B[0][2] * i0, l30 = B[0][3] * i0;
  const double d1 = B[1][1] + mu - l10 * l10 * d0, i1 = 1.0 / d1;
  const double l21 = (B[1][2] - l20 * l10 * d0) * i1, l31 = (B[1][3] - l30 * l10 * d0) * i1;
  const double d2 = B[2][2] + mu - l20 * l20 * d0 - l21 * l21 * d1, i2 = 1.0 / d2;
  const double l32 = (B[2][3] - l30 * l20 * d0 - l31 * l21 * d1) * i2;
  const double d3 = B[3][3] + mu - l30 * l30 * d0 - l31 * l31 * d1 - l32 * l32 * d2, i3 = 1.0 / d3;
  double h0 = 0, h1 = 0, h2 = 0, h3 = 1, delta = 0;
  auto inverse_step = [&]() {
    // L y = h
    const double y0 = h0, y1 = h1 - l10 * y0, y2 = h2 - l20 * y0 - l21 * y1, y3 = h3 - l30 * y0 - l31 * y1 - l32 * y2;
    // D z = y, L^T x = z
    const double x3 = y3 * i3, x2 = y2 * i2 - l32 * x3, x1 = y1 * i1 - l21 * x2 - l31 * x3, x0 = y0 * i0 - l10 * x1 - l20 * x2 - l30 * x3;
    const double m = fmax(fmax(fabs(x0), fabs(x1)), fmax(fabs(x2), fabs(x3)));
    const double sc = (m > 0 && isfinite(m)) ? 1.0 / m : 1.0;
    const double n0 = x0 * sc, n1 = x1 * sc, n2 = x2 * sc, n3 = x3 * sc;
    // change of the direction (the iterates are scaled to maximum norm 1; B^-1 is positive definite, so they do not change sign)
    delta = fmax(fmax(fabs(n0 - h0), fabs(n1 - h1)), fmax(fabs(n2 - h2), fabs(n3 - h3)));
    h0 = n0; h1 = n1; h2 = n2; h3 = n3;
  };
#pragma unroll
  for (int it = 0; it < 4; ++it) inverse_step();
  // Four steps settle every correspondence the pipeline lets through (y threshold, minimum disparity: base.cpp:127-207).  A caller of
  // spvo_triangulate with unfiltered matches can hand in a point at (almost) infinity -- its null vector has w ~ 0, nearly orthogonal
  // to the start vector -- or a pair with a large vertical offset (sigma_4 / sigma_3 not small): keep iterating until the direction
  // stands still (the serial Jacobi SVD this replaced needed no such care; the result is the same vector)
  for (int it = 4; it < 200 && delta > 1e-13; ++it) inverse_step();
  // cv::triangulatePoints stores the UNIT-NORM right-singular vector (a row of the SVD's V^T) as CV_32F BEFORE the division by w
  // (base.cpp:212, 223): where each component's f32 rounding falls depends on the vector's scale.  Scaled to maximum norm 1, as the iteration
  // leaves it, four of five points came out one to four f32 units away from the oracle's -- enough to move a point across the RANSAC
  // threshold once in a few hundred frames (round 6: tests/test_gpu_long_sequence.py, frame 7).  With the 2-norm normalisation of the SVD
  // the stored f32 bits agree except where the f64 directions (~1e-11 apart) straddle an f32 rounding boundary.
  {
    const double inv_n = 1.0 / sqrt(h0 * h0 + h1 * h1 + h2 * h2 + h3 * h3);
    h0 *= inv_n; h1 *= inv_n; h2 *= inv_n; h3 *= inv_n;
  }
  const float f0 = (float)h0, f1 = (float)h1, f2 = (float)h2, f3 = (float)h3;
  const float scale = (f3 != 0.f) ? __fdiv_rn(1.0f, f3) : 1.0f;
  out[0] = mul_rn(f0, scale);
  out[1] = mul_rn(f1, scale);
  out[2] = mul_rn(f2, scale);
}

__global__ __launch_bounds__(256) void triangulate_kernel(const double *__restrict__ Pl,
                                                          const double *__restrict__ Pr,
                                                          const float *__restrict__ xyl,
                                                          const float *__restrict__ xyr, int n,
                                                          float *__restrict__ xyz) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  triangulate_point(Pl, Pr, xyl[2 * i], xyl[2 * i + 1], xyr[2 * i], xyr[2 * i + 1], xyz + 3 * i);
}

// The fused solve's first kernel (spvo_solve_submit): the same triangulation reading the call's packed inputs STRAIGHT from the pinned host
// buffer the caller filled (64 doubles of header, then cl cr pl pr [2n floats each], prev_xyz [3n], prev_valid [n]) and leaving the device
// copy the later kernels read -- a host-to-device copy in front of this kernel was one more dependent operation on the solver's stream, and
// every operation of that chain costs ~8-10 us whatever it does (the chain's latency bounds the frame loop of the small engines: a frame's
// solve needs the previous frame's pose as its prior).  Every byte crosses PCIe once: 16-byte pieces spread over the grid, the projection
// matrices once per workgroup into LDS.
__global__ __launch_bounds__(256) void solve_in_triangulate_kernel(const uint4 *__restrict__ h_in, uint4 *__restrict__ d_in, int n16, int n, float *__restrict__ xyz) {
  __shared__ double sP[24];
  const int tid = threadIdx.x, i = blockIdx.x * 256 + tid;
  for (int k = i; k < n16; k += gridDim.x * 256) d_in[k] = h_in[k];
  if (tid < 24) sP[tid] = reinterpret_cast<const double *>(h_in)[tid];
  __syncthreads();
  if (i >= n) return;
  const float2 *fw = reinterpret_cast<const float2 *>(h_in + 32);   // behind the 512-byte header
  const float2 l = fw[i], r = fw[n + i];
  triangulate_point(sP, sP + 12, l.x, l.y, r.x, r.y, xyz + 3 * i);
}

// ------------------------------------------------------------------------- K15
__device__ __forceinline__ uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}

// residual (2) and Jacobian (2x6: small left rotation, translation) of one 3D-2D pair
__device__ __forceinline__ void pnp_residual_jac(const double *K, const double *R, const double *t,
                                                 const double *X, const double *uv, double *r,
                                                 double *J) {
  const double Y0 = R[0] * X[0] + R[1] * X[1] + R[2] * X[2];
  const double Y1 = R[3] * X[0] + R[4] * X[1] + R[5] * X[2];
  const double Y2 = R[6] * X[0] + R[7] * X[1] + R[8] * X[2];
  const double c0 = Y0 + t[0], c1 = Y1 + t[1], c2 = Y2 + t[2];
  const double p0 = K[0] * c0 + K[1] * c1 + K[2] * c2;
  const double p1 = K[3] * c0 + K[4] * c1 + K[5] * c2;
  const double p2 = K[6] * c0 + K[7] * c1 + K[8] * c2;
  const double u = p0 / p2, v = p1 / p2;
  r[0] = u - uv[0];
  r[1] = v - uv[1];
  if (J) {
    const double du0 = (K[0] - u * K[6]) / p2, du1 = (K[1] - u * K[7]) / p2, du2 = (K[2] - u * K[8]) / p2;
    const double dv0 = (K[3] - v * K[6]) / p2, dv1 = (K[4] - v * K[7]) / p2, dv2 = (K[5] - v * K[8]) / p2;
    // -(d . [Y]x): column k of -[Y]x ; [Y]x = [[0,-Y2,Y1],[Y2,0,-Y0],[-Y1,Y0,0]]
    J[0] = -(du1 * Y2 - du2 * Y1);
    J[1] = -(-du0 * Y2 + du2 * Y0);
    J[2] = -(du0 * Y1 - du1 * Y0);
    J[3] = du0; J[4] = du1; J[5] = du2;
    J[6] = -(dv1 * Y2 - dv2 * Y1);
    J[7] = -(-dv0 * Y2 + dv2 * Y0);
    J[8] = -(dv0 * Y1 - dv1 * Y0);
    J[9] = dv0; J[10] = dv1; J[11] = dv2;
  }
}

__device__ __forceinline__ void apply_delta(double *q, double *t, const double *d) {
  const double dq[4] = {d[0] / 2, d[1] / 2, d[2] / 2, 1.0};
  double qn[4];
  quat_mul(dq, q, qn);
  const double n = sqrt(qn[0] * qn[0] + qn[1] * qn[1] + qn[2] * qn[2] + qn[3] * qn[3]);
  q[0] = qn[0] / n; q[1] = qn[1] / n; q[2] = qn[2] / n; q[3] = qn[3] / n;
  t[0] += d[3]; t[1] += d[4]; t[2] += d[5];
}

__device__ __forceinline__ bool reproj_inlier(const double *K, const double *R, const double *t,
                                              const float *X, const float *uv, double thr2) {
  const double x = X[0], y = X[1], z = X[2];
  const double c0 = R[0] * x + R[1] * y + R[2] * z + t[0];
  const double c1 = R[3] * x + R[4] * y + R[5] * z + t[1];
  const double c2 = R[6] * x + R[7] * y + R[8] * z + t[2];
  const double p0 = K[0] * c0 + K[1] * c1 + K[2] * c2;
  const double p1 = K[3] * c0 + K[4] * c1 + K[5] * c2;
  const double p2 = K[6] * c0 + K[7] * c1 + K[8] * c2;
  const double du = p0 / p2 - (double)uv[0], dv = p1 / p2 - (double)uv[1];
  const double e2 = du * du + dv * dv;
  return (p2 > 0) && (e2 <= thr2);
}

// (RansacWork: spvo_types.hip.h)

// Lane broadcast of a double (two v_readlane_b32; the source lane is wave-uniform).
__device__ __forceinline__ double lane_bcast(double v, int src_lane) {
  const unsigned long long u = (unsigned long long)__double_as_longlong(v);
  const unsigned lo = __builtin_amdgcn_readlane((int)(unsigned)u, src_lane), hi = __builtin_amdgcn_readlane((int)(unsigned)(u >> 32), src_lane);
  return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// solve_linear<6> with the six ROWS held by six lanes: lane r < 6 owns row r of A (a[0..5]) and b[r] (rb); every lane of the
// wave executes the same code (lanes >= 6 carry dead rows).  Gaussian elimination with partial pivoting, the same pivots,
// multipliers and update order as the serial form -- the rows are not moved, a pivot row is marked used instead -- so the result
// is the serial one bit for bit; what changes is the latency: the serial form walks a dynamically indexed 6 x 6 array in LDS
// (~250 dependent LDS accesses per solve: 90 us for a ten-iteration Newton solve in one lane), this one keeps the rows in
// registers, one row update per lane in parallel and ~45 instructions per column.  x[0..5] is returned in every lane.
__device__ __forceinline__ bool wave_solve6(double (&a)[6], double rb, int lane, double (&x)[6]) {
  bool used = lane >= 6;
  int piv_lane[6];
#pragma unroll
  for (int c = 0; c < 6; ++c) {
    // pivot: the unused row with the largest |a[c]|, the lowest row on ties (rows in their original order)
    const double mine = used ? -1.0 : fabs(a[c]);
    double best = -1.0;
    int bl = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) {
      const double v = lane_bcast(mine, r);
      if (v > best) { best = v; bl = r; }
    }
    if (!(best > 1e-300) || !isfinite(best)) return false;
    piv_lane[c] = bl;
    double pr[6];
#pragma unroll
    for (int k = c; k < 6; ++k) pr[k] = lane_bcast(a[k], bl);
    const double pb = lane_bcast(rb, bl);
    const double inv = 1.0 / pr[c];
    if (lane == bl) used = true;
    if (!used) {
      const double f = a[c] * inv;
      if (f != 0) {
#pragma unroll
        for (int k = c; k < 6; ++k) a[k] -= f * pr[k];
        rb -= f * pb;
      }
    }
  }
  // back substitution: x[r] from the row that was the pivot of column r
#pragma unroll
  for (int r = 5; r >= 0; --r) {
    double sum = rb;
#pragma unroll
    for (int k = r + 1; k < 6; ++k) sum -= a[k] * x[k];
    x[r] = lane_bcast(sum / a[r], piv_lane[r]);
  }
#pragma unroll
  for (int k = 0; k < 6; ++k)
    if (!isfinite(x[k])) return false;
  return true;
}

// one wave per hypothesis: the minimal problem is solved by lanes 0..5 together (one residual row each), all lanes score.
// A device function of (hypothesis, lane) -- nothing in it leaves the wave (no LDS, no barrier): ransac_hypothesis_kernel runs it with
// one wave per workgroup, solve_hyp_tail_kernel with eight (the same instructions on the same data: identical bits)
__device__ __forceinline__ void ransac_hypothesis_body(const double *__restrict__ Kd, const float *__restrict__ xyz, const float *__restrict__ xy, int n,
                                                       const double *__restrict__ prior,  // rvec, tvec
                                                       uint32_t seed, double thr2, RansacWork w, const int it, const int lane) {
  double K[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) K[k] = Kd[k];
  // the sample (every lane computes it: three hashes) and this lane's point: lane l < 6 owns residual row l = (point l >> 1, u or v)
  int idx[3];
  for (int k = 0; k < 3; ++k) {
    uint32_t attempt = 0;
    while (true) {
      const uint32_t r = hash32(seed * 0x9E3779B9u + (uint32_t)it * 0x85EBCA6Bu + (uint32_t)k * 0xC2B2AE35u + attempt * 0x27D4EB2Fu) % (uint32_t)n;
      bool dup = false;
      for (int m = 0; m < k; ++m) dup |= (idx[m] == (int)r);
      if (!dup) { idx[k] = (int)r; break; }
      ++attempt;
    }
  }
  const int my_pt = min(lane >> 1, 2), my_row = lane & 1;
  const int pi = my_pt == 0 ? idx[0] : my_pt == 1 ? idx[1] : idx[2];
  const double X3[3] = {xyz[3 * pi], xyz[3 * pi + 1], xyz[3 * pi + 2]};
  const double uv3[2] = {xy[2 * pi], xy[2 * pi + 1]};
  // q (xyzw), t: identical in every lane.  PRIOR-FREE (round 6): the Newton iteration of every sample starts at the identity, as
  // cv::solvePnPRansac's closed-form minimal solver needs no guess either (base.cpp:237-239) -- a frame's hypotheses no longer depend
  // on the previous frame's pose, so its whole RANSAC can run before that pose exists (oracle/odometry.py: pnp_ransac)
  (void)prior;
  double sh[7] = {0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0};
  bool ok = false, bad = false;
  auto wave_max6 = [&](double v) {   // max over lanes 0..5, in every lane
    double m = 0;
#pragma unroll
    for (int r = 0; r < 6; ++r) m = fmax(m, lane_bcast(v, r));
    return m;
  };
  for (int iter = 0; iter < 10 && !bad; ++iter) {
    double R[9], f2[2], J12[12];
    quat_to_rot(sh, R);
    pnp_residual_jac(K, R, sh + 4, X3, uv3, f2, J12);
    double row[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) row[k] = my_row ? J12[6 + k] : J12[k];
    const double fr = my_row ? f2[1] : f2[0];
    bool fin = isfinite(fr);
#pragma unroll
    for (int k = 0; k < 6; ++k) fin &= isfinite(row[k]);
    if (__ballot(!fin && lane < 6)) { bad = true; break; }
    const double fm = wave_max6(fabs(fr));
    if (fm < 1e-9) { ok = true; break; }
    double d[6];
    if (!wave_solve6(row, -fr, lane, d)) { bad = true; break; }
    double dm = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) dm = fmax(dm, fabs(d[k]));
    if (dm > 1e3) { bad = true; break; }
    apply_delta(sh, sh + 4, d);
  }
  if (!ok && !bad) {
    double R[9], f2[2];
    quat_to_rot(sh, R);
    pnp_residual_jac(K, R, sh + 4, X3, uv3, f2, nullptr);
    const double fr = my_row ? f2[1] : f2[0];
    const bool fin = !__ballot(!isfinite(fr) && lane < 6);
    ok = fin && wave_max6(fabs(fr)) < 1e-6;
  }
  if (!(ok && !bad)) {
    if (lane == 0) w.counts[it] = -1;
    return;
  }
  double R[9], t[3];
  quat_to_rot(sh, R);
  t[0] = sh[4]; t[1] = sh[5]; t[2] = sh[6];
  int cnt = 0;
  for (int i = lane; i < n; i += 64) cnt += reproj_inlier(K, R, t, xyz + 3 * i, xy + 2 * i, thr2) ? 1 : 0;
#pragma unroll
  for (int o = 32; o >= 1; o >>= 1) cnt += __shfl_xor(cnt, o);
  if (lane == 0) {
    w.counts[it] = cnt;
    for (int k = 0; k < 7; ++k) w.poses[7 * it + k] = sh[k];
  }
}

__global__ __launch_bounds__(64) void ransac_hypothesis_kernel(const double *__restrict__ Kd, const float *__restrict__ xyz, const float *__restrict__ xy, int n,
                                                               const double *__restrict__ prior, uint32_t seed, double thr2, RansacWork w) {
  ransac_hypothesis_body(Kd, xyz, xy, n, prior, seed, thr2, w, (int)blockIdx.x, (int)threadIdx.x);
}

// deterministic block reduction of NV <= 32 doubles per thread; result in `out` (LDS), valid for every thread behind the closing barrier.
// Within a wave the values are reduced TRANSPOSED: at the step over lane bit 5 a lane keeps half of its values and hands the other half to
// its partner (which keeps that half), and so on down to one value per lane pair -- 16 + 8 + 4 + 2 + 1 + 1 = 32 exchanges for all values
// instead of 6 per value (168 for the 28 sums of a Levenberg-Marquardt evaluation: ds_bpermute traffic of eight waves through one LDS was a
// third of an evaluation).  Lane l ends with the wave's sum of value l >> 1; the order of the additions is fixed by the lane numbers.
template <int NV, int NT>
__device__ inline void block_reduce(const double *v, double *out, double *scratch /*[NV][NT/64]*/) {
  static_assert(NV <= 32, "block_reduce: at most 32 values");
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  double a[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) a[k] = k < NV ? v[k] : 0.0;
#pragma unroll
  for (int h = 32, n = 16; h >= 2; h >>= 1, n >>= 1) {
    const bool up = (lane & h) != 0;
#pragma unroll
    for (int k = 0; k < n; ++k) {
      if (k >= NV && k + n >= NV) continue;                 // (both halves are padding)
      const double send = up ? a[k] : a[k + n], keep = up ? a[k + n] : a[k];
      a[k] = keep + __shfl_xor(send, h);
    }
  }
  a[0] += __shfl_xor(a[0], 1);
  if (!(lane & 1) && (lane >> 1) < NV) scratch[(lane >> 1) * (NT / 64) + wave] = a[0];
  __syncthreads();
  if (threadIdx.x < NV) {   // one value per thread, the waves' partial sums in wave order
    const int k = threadIdx.x;
    double s = 0;
#pragma unroll
    for (int wv = 0; wv < NT / 64; ++wv) s += scratch[k * (NT / 64) + wv];
    out[k] = s;
  }
  __syncthreads();
}

// pick the best hypothesis, list its inliers in ascending order, Gauss-Newton refit
// (a device function on NT threads: the stand-alone kernel below and the fused solve's tail kernel run the SAME instantiation, so that
// spvo_pnp_ransac and spvo_solve_* return identical bits)
template <int NT>
__device__ __forceinline__ void ransac_select_body(const double *Kd, const float *xyz, const float *xy, int n, const double *prior,
                                                   int iterations, double thr2, RansacWork w) {
  __shared__ int s_cnt[NT], s_it[NT];
  __shared__ double s_pose[7];
  __shared__ int s_base, s_wave_cnt[NT / 64];
  __shared__ double s_red[27 * (NT / 64)];
  __shared__ double s_sum[27];
  __shared__ int s_flag;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  double K[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) K[k] = Kd[k];
  int bc = -1, bi = 0x7FFFFFFF;
  for (int it = tid; it < iterations; it += NT) {
    const int c = w.counts[it];
    if (c > bc) { bc = c; bi = it; }   // ascending it per thread: first max kept
  }
  s_cnt[tid] = bc; s_it[tid] = bi;
  __syncthreads();
  for (int s = NT / 2; s >= 1; s >>= 1) {
    if (tid < s) {
      const int oc = s_cnt[tid + s], oi = s_it[tid + s];
      if (oc > s_cnt[tid] || (oc == s_cnt[tid] && oi < s_it[tid])) { s_cnt[tid] = oc; s_it[tid] = oi; }
    }
    __syncthreads();
  }
  const int best_cnt = s_cnt[0], best_it = s_it[0];
  if (best_cnt < 4) {
    if (tid == 0) {
      for (int k = 0; k < 6; ++k) w.result[k] = prior[k];
      w.result[6] = 0; w.result[7] = 0;
    }
    return;
  }
  if (tid < 7) s_pose[tid] = w.poses[7 * best_it + tid];
  if (tid == 0) s_base = 0;
  __syncthreads();
  double R[9], t[3];
  quat_to_rot(s_pose, R);
  t[0] = s_pose[4]; t[1] = s_pose[5]; t[2] = s_pose[6];
  // ordered compaction of the inlier indices
  for (int base = 0; base < n; base += NT) {
    const int i = base + tid;
    const bool in = (i < n) && reproj_inlier(K, R, t, xyz + 3 * i, xy + 2 * i, thr2);
    const unsigned long long m = __ballot(in);
    if (lane == 0) s_wave_cnt[wave] = __popcll(m);
    __syncthreads();
    int off = s_base;
    for (int wv = 0; wv < wave; ++wv) off += s_wave_cnt[wv];
    if (in) w.inliers[off + __popcll(m & ((1ull << lane) - 1ull))] = i;
    __syncthreads();
    if (tid == 0) for (int wv = 0; wv < NT / 64; ++wv) s_base += s_wave_cnt[wv];
    __syncthreads();
  }
  const int ninl = s_base;
  // Gauss-Newton refit on the inliers (left camera, no robust loss)
  for (int iter = 0; iter < 10; ++iter) {
    quat_to_rot(s_pose, R);
    t[0] = s_pose[4]; t[1] = s_pose[5]; t[2] = s_pose[6];
    double acc[27];
#pragma unroll
    for (int k = 0; k < 27; ++k) acc[k] = 0;
    for (int k = tid; k < ninl; k += NT) {
      const int i = w.inliers[k];
      const double X[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
      const double uv[2] = {xy[2 * i], xy[2 * i + 1]};
      double r[2], J[12];
      pnp_residual_jac(K, R, t, X, uv, r, J);
      int o = 0;
#pragma unroll
      for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int b = a; b < 6; ++b) acc[o++] += J[a] * J[b] + J[6 + a] * J[6 + b];
#pragma unroll
      for (int a = 0; a < 6; ++a) acc[21 + a] += J[a] * r[0] + J[6 + a] * r[1];
    }
    block_reduce<27, NT>(acc, s_sum, s_red);
    if (wave == 0) {   // the 6 x 6 normal equations: one row per lane (wave_solve6), lane 0 applies the step
      const int ra = min(lane, 5);
      double row[6];
#pragma unroll
      for (int b = 0; b < 6; ++b) {
        const int lo = min(ra, b), hi = max(ra, b);
        row[b] = s_sum[lo * 6 - lo * (lo - 1) / 2 + (hi - lo)];
      }
      double d[6];
      const bool solved = wave_solve6(row, -s_sum[21 + ra], lane, d);
      if (lane == 0) {
        int flag = 1;
        if (solved) {
          apply_delta(s_pose, s_pose + 4, d);
          double dm = 0;
          for (int a = 0; a < 6; ++a) dm = fmax(dm, fabs(d[a]));
          flag = (dm < 1e-10) ? 1 : 0;
        }
        s_flag = flag;
      }
    }
    __syncthreads();
    if (s_flag) break;
  }
  if (tid == 0) {
    quat_to_rvec(s_pose, w.result);
    w.result[3] = s_pose[4]; w.result[4] = s_pose[5]; w.result[5] = s_pose[6];
    w.result[6] = 1; w.result[7] = (double)ninl;
  }
}

constexpr int SOLVE_TAIL_THREADS = 512;   // the one workgroup behind the hypotheses: selection, gating, refinement

__global__ __launch_bounds__(SOLVE_TAIL_THREADS) void ransac_select_kernel(const double *Kd, const float *xyz, const float *xy, int n, const double *prior,
                                                                           int iterations, double thr2, RansacWork w) {
  ransac_select_body<SOLVE_TAIL_THREADS>(Kd, xyz, xy, n, prior, iterations, thr2, w);
}

// ------------------------------------------------------------------------- K16
// (ObsDev: spvo_types.hip.h)

// CostFunctor32 (cost.hpp:27-58): residual and analytic Jacobian wrt the
// EigenQuaternionParameterization tangent (3) and t (3).
__device__ __forceinline__ void cost32(const double *P /*3x4*/, const double *q, const double *R,
                                       const double *t, const ObsDev &ob, double *r, double *J) {
  const double X[3] = {ob.X[0], ob.X[1], ob.X[2]};
  const bool inv = ob.inverse != 0;
  const double w0 = inv ? X[0] - t[0] : X[0], w1 = inv ? X[1] - t[1] : X[1], w2 = inv ? X[2] - t[2] : X[2];
  double T0, T1, T2;
  if (!inv) {
    T0 = R[0] * w0 + R[1] * w1 + R[2] * w2 + t[0];
    T1 = R[3] * w0 + R[4] * w1 + R[5] * w2 + t[1];
    T2 = R[6] * w0 + R[7] * w1 + R[8] * w2 + t[2];
  } else {  // R^T (X - t)
    T0 = R[0] * w0 + R[3] * w1 + R[6] * w2;
    T1 = R[1] * w0 + R[4] * w1 + R[7] * w2;
    T2 = R[2] * w0 + R[5] * w1 + R[8] * w2;
  }
  const double p0 = P[0] * T0 + P[1] * T1 + P[2] * T2 + P[3];
  const double p1 = P[4] * T0 + P[5] * T1 + P[6] * T2 + P[7];
  const double p2 = P[8] * T0 + P[9] * T1 + P[10] * T2 + P[11];
  const double u = p0 / p2, v = p1 / p2;
  r[0] = u - (double)ob.uv[0];
  r[1] = v - (double)ob.uv[1];
  if (!J) return;
  const double du[3] = {(P[0] - u * P[8]) / p2, (P[1] - u * P[9]) / p2, (P[2] - u * P[10]) / p2};
  const double dv[3] = {(P[4] - v * P[8]) / p2, (P[5] - v * P[9]) / p2, (P[6] - v * P[10]) / p2};
  const double x = q[0], y = q[1], z = q[2], w = q[3];
  // dR/dq_k (Eigen polynomial), rows
  const double dRx[9] = {0, 2 * y, 2 * z, 2 * y, -4 * x, -2 * w, 2 * z, 2 * w, -4 * x};
  const double dRy[9] = {-4 * y, 2 * x, 2 * w, 2 * x, 0, 2 * z, -2 * w, 2 * z, -4 * y};
  const double dRz[9] = {-4 * z, -2 * w, 2 * x, 2 * w, -4 * z, 2 * y, 2 * x, 2 * y, 0};
  const double dRw[9] = {0, -2 * z, 2 * y, 2 * z, 0, -2 * x, -2 * y, 2 * x, 0};
  const double *dR[4] = {dRx, dRy, dRz, dRw};
  double Jq_u[4], Jq_v[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const double *D = dR[k];
    double e0, e1, e2;
    if (!inv) {
      e0 = D[0] * w0 + D[1] * w1 + D[2] * w2;
      e1 = D[3] * w0 + D[4] * w1 + D[5] * w2;
      e2 = D[6] * w0 + D[7] * w1 + D[8] * w2;
    } else {
      e0 = D[0] * w0 + D[3] * w1 + D[6] * w2;
      e1 = D[1] * w0 + D[4] * w1 + D[7] * w2;
      e2 = D[2] * w0 + D[5] * w1 + D[8] * w2;
    }
    Jq_u[k] = du[0] * e0 + du[1] * e1 + du[2] * e2;
    Jq_v[k] = dv[0] * e0 + dv[1] * e1 + dv[2] * e2;
  }
  // Plus jacobian G (4x3): rows [w,z,-y], [-z,w,x], [y,-x,w], [-x,-y,-z]
  J[0] = Jq_u[0] * w - Jq_u[1] * z + Jq_u[2] * y - Jq_u[3] * x;
  J[1] = Jq_u[0] * z + Jq_u[1] * w - Jq_u[2] * x - Jq_u[3] * y;
  J[2] = -Jq_u[0] * y + Jq_u[1] * x + Jq_u[2] * w - Jq_u[3] * z;
  J[6] = Jq_v[0] * w - Jq_v[1] * z + Jq_v[2] * y - Jq_v[3] * x;
  J[7] = Jq_v[0] * z + Jq_v[1] * w - Jq_v[2] * x - Jq_v[3] * y;
  J[8] = -Jq_v[0] * y + Jq_v[1] * x + Jq_v[2] * w - Jq_v[3] * z;
  if (!inv) {
    J[3] = du[0]; J[4] = du[1]; J[5] = du[2];
    J[9] = dv[0]; J[10] = dv[1]; J[11] = dv[2];
  } else {  // d/dt R^T (X - t) = -R^T  ->  -(d . R^T) = -(R d)
    J[3] = -(R[0] * du[0] + R[1] * du[1] + R[2] * du[2]);
    J[4] = -(R[3] * du[0] + R[4] * du[1] + R[5] * du[2]);
    J[5] = -(R[6] * du[0] + R[7] * du[1] + R[8] * du[2]);
    J[9] = -(R[0] * dv[0] + R[1] * dv[1] + R[2] * dv[2]);
    J[10] = -(R[3] * dv[0] + R[4] * dv[1] + R[5] * dv[2]);
    J[11] = -(R[6] * dv[0] + R[7] * dv[1] + R[8] * dv[2]);
  }
}

// (RefineOut: spvo_types.hip.h)

// Whole Levenberg-Marquardt loop in ONE workgroup: no host round trips.
// All threads evaluate residual blocks; thread 0 runs the trust-region logic.
// `ctl` (optional, device): ctl[0] = run flag, ctl[1] = n_obs -- written by solve_gate_build_body (solve_tail_kernel)
// when the whole of solveStereoOdometry is enqueued without a host round trip.
template <int NT>
__device__ __forceinline__ void pnp_refine_body(const double *Pl, const double *Pr, const ObsDev *obs, int n_obs_host, const int *ctl, const double *start /*q,t*/,
                                                int max_iterations, double huber_delta, RefineOut *out) {
  const int n_obs = ctl ? ctl[1] : n_obs_host;
  if (ctl && ctl[0] == 0) {   // gated out (base.cpp:244-260) or refinement_degree == 0
    if (threadIdx.x == 0) {
      for (int k = 0; k < 7; ++k) out->v[k] = start[k];
      out->v[7] = 0; out->v[8] = 0; out->v[9] = 0; out->v[10] = 0; out->v[11] = 0;
    }
    return;
  }
  __shared__ double s_P[24];
  __shared__ double s_x[7], s_c[7];       // current / candidate parameters
  __shared__ double s_red[28 * (NT / 64)];
  __shared__ double s_sum[28];
  __shared__ int s_action;                 // 0 continue with candidate eval, 1 stop
  const int tid = threadIdx.x;
  if (tid < 12) { s_P[tid] = Pl[tid]; s_P[12 + tid] = Pr[tid]; }
  if (tid < 7) s_x[tid] = start[tid];
  __syncthreads();
  const double b2 = huber_delta * huber_delta;

  // evaluate at params p: acc[0..20] JtJ upper, [21..26] Jtr, [27] = sum rho
  auto evaluate = [&](const double *p, bool want_jac) {
    double q[4] = {p[0], p[1], p[2], p[3]}, t[3] = {p[4], p[5], p[6]}, R[9];
    quat_to_rot(q, R);
    double acc[28];
#pragma unroll
    for (int k = 0; k < 28; ++k) acc[k] = 0;
    for (int i = tid; i < n_obs; i += NT) {
      const ObsDev ob = obs[i];
      double r[2], J[12];
      cost32(s_P + (ob.cam ? 12 : 0), q, R, t, ob, r, want_jac ? J : nullptr);
      const double s = r[0] * r[0] + r[1] * r[1];
      double rho, rho1;
      if (s > b2) { const double sq = sqrt(s); rho = 2 * huber_delta * sq - b2; rho1 = huber_delta / sq; }
      else { rho = s; rho1 = 1.0; }
      acc[27] += rho;
      if (want_jac) {
        // corrector with rho'' <= 0: residual and Jacobian scaled by sqrt(rho')
        int o = 0;
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
          for (int b = a; b < 6; ++b) acc[o++] += rho1 * (J[a] * J[b] + J[6 + a] * J[6 + b]);
#pragma unroll
        for (int a = 0; a < 6; ++a) acc[21 + a] += rho1 * (J[a] * r[0] + J[6 + a] * r[1]);
      }
    }
    block_reduce<28, NT>(acc, s_sum, s_red);
  };

  // thread-0 state (arrays in LDS: they are indexed dynamically)
  __shared__ double A[36], g[6], scale[6];
  double cost = 0, radius = 1e4, decrease = 2.0, x_norm = 0;
  double model_change = 0;
  int invalid = 0, it = 0, converged = 0, usable = 0;
  double initial_cost = 0, final_cost = 0;

  evaluate(s_x, true);
  // The trust-region logic runs on every lane of wave 0 (its scalars are wave-uniform: the step below needs them in all lanes); lane 0
  // writes what lives in LDS.  block_reduce's closing barrier makes s_sum visible.
  auto unpack = [&]() {   // s_sum -> A (symmetric), g
    if (tid == 0) {
      int o = 0;
      for (int a = 0; a < 6; ++a)
        for (int b = a; b < 6; ++b) { A[a * 6 + b] = s_sum[o]; A[b * 6 + a] = s_sum[o]; ++o; }
      for (int a = 0; a < 6; ++a) g[a] = s_sum[21 + a];
    }
  };
  if (tid < 64) {
    unpack();
    cost = 0.5 * s_sum[27];
    initial_cost = final_cost = cost;
    if (tid == 0) for (int a = 0; a < 6; ++a) scale[a] = 1.0 / (1.0 + sqrt(s_sum[a * 6 - a * (a - 1) / 2]));   // (the diagonal entry A[a][a])
    x_norm = 0;
    for (int k = 0; k < 7; ++k) x_norm += s_x[k] * s_x[k];
    x_norm = sqrt(x_norm);
    int stop = 0;
    if (!isfinite(cost)) stop = 1;
    else {
      usable = 1;
      double gm = 0;
      for (int a = 0; a < 6; ++a) gm = fmax(gm, fabs(s_sum[21 + a]));
      if (gm <= 1e-10) { converged = 1; stop = 1; }
    }
    if (n_obs == 0) { converged = 1; usable = 1; stop = 1; }
    if (tid == 0) s_action = stop;
  }
  __syncthreads();

  while (!s_action) {
    // ---- wave 0: compute a trust-region step (possibly several invalid ones).  The 6 x 6 system is solved across lanes (wave_solve6: lane
    // r < 6 holds row r in registers; the same pivots, multipliers and update order as the serial solve_linear<6>, which walked a
    // dynamically indexed array in LDS -- ~9 us per solve, the largest single piece of an iteration and of the whole solver chain, whose
    // latency is the cycle time of the small engines' frame loop).  Every lane carries the loop's scalars (it, invalid, radius, decrease,
    // model_change): they are wave-uniform, the other waves get them back through LDS below.
    if (tid < 64) {
      const int lane = tid, ra = min(lane, 5);
      int stop = 0, have_step = 0;
      while (!stop && !have_step) {
        if (it >= max_iterations) { stop = 1; break; }
        ++it;
        const double sc_r = scale[ra], gs_r = g[ra] * sc_r;
        double As_r[6], M_r[6];
#pragma unroll
        for (int b = 0; b < 6; ++b) { As_r[b] = A[ra * 6 + b] * sc_r * scale[b]; M_r[b] = As_r[b]; }
        double diag = 0;
#pragma unroll
        for (int b = 0; b < 6; ++b) diag = (b == ra) ? As_r[b] : diag;
        const double dg = fmin(fmax(diag, 1e-6), 1e32) / radius;
#pragma unroll
        for (int b = 0; b < 6; ++b) M_r[b] += (b == ra) ? dg : 0.0;
        double x[6];
        const bool ok = wave_solve6(M_r, -gs_r, lane, x);
        model_change = 0;
        if (ok) {
          double Ad = 0;
#pragma unroll
          for (int b = 0; b < 6; ++b) Ad += As_r[b] * x[b];
          const double inner = gs_r + 0.5 * Ad;          // row ra's (gs[a] + 0.5 (As d)[a])
#pragma unroll
          for (int a = 0; a < 6; ++a) model_change -= x[a] * lane_bcast(inner, a);   // in the order a = 0 .. 5 of the serial form
        }
        if (!ok || !(model_change > 0)) {
          if (++invalid >= 5) { usable = 0; stop = 1; break; }
          radius /= decrease; decrease *= 2;
          continue;
        }
        invalid = 0;
        // candidate = Plus(x, ds * scale)
        double d[6];
#pragma unroll
        for (int a = 0; a < 6; ++a) d[a] = x[a] * scale[a];
        const double nd = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        double qc[4];
        if (nd > 0) {
          const double sn = sin(nd) / nd;
          const double dq[4] = {sn * d[0], sn * d[1], sn * d[2], cos(nd)};
          quat_mul(dq, s_x, qc);
        } else {
          for (int k = 0; k < 4; ++k) qc[k] = s_x[k];
        }
        if (lane == 0) {
          for (int k = 0; k < 4; ++k) s_c[k] = qc[k];
          for (int k = 0; k < 3; ++k) s_c[4 + k] = s_x[4 + k] + d[3 + k];
        }
        have_step = 1;
      }
      if (lane == 0) s_action = stop;
    }
    __syncthreads();
    if (s_action) break;
    // cost AND normal equations at the candidate in ONE pass over the residual blocks: an accepted candidate is the next iterate, and its
    // Jacobian pass (a second evaluation of the same point, a second reduction, a second pair of barriers) was a third of an iteration;
    // a rejected candidate (rare) wasted the Jacobian part
    evaluate(s_c, true);
    if (tid < 64) {
      const double cand_cost = 0.5 * s_sum[27];
      double sn2 = 0;
      for (int k = 0; k < 7; ++k) sn2 += (s_c[k] - s_x[k]) * (s_c[k] - s_x[k]);
      int stop = 0, accept = 0;
      if (sqrt(sn2) <= 1e-8 * (x_norm + 1e-8)) { converged = 1; stop = 1; }
      else {
        const double cost_change = cost - cand_cost;
        if (fabs(cost_change) <= 1e-6 * cost) { converged = 1; stop = 1; }
        else {
          const double rel = cost_change / model_change;
          if (isfinite(cand_cost) && rel > 1e-3) {
            accept = 1;
            radius = fmin(1e16, radius / fmax(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) * (2.0 * rel - 1.0) * (2.0 * rel - 1.0)));
            decrease = 2.0;
          } else {
            radius /= decrease; decrease *= 2;
          }
        }
      }
      if (accept) {   // the candidate becomes the iterate: its sums are the next step's system
        unpack();
        cost = cand_cost;
        final_cost = cost;
        x_norm = 0;
        for (int k = 0; k < 7; ++k) x_norm += s_c[k] * s_c[k];
        x_norm = sqrt(x_norm);
        double gm = 0;
        for (int a = 0; a < 6; ++a) gm = fmax(gm, fabs(s_sum[21 + a]));
        if (gm <= 1e-10 || radius < 1e-32) { converged = 1; stop = 1; }
        if (tid == 0) for (int k = 0; k < 7; ++k) s_x[k] = s_c[k];
      }
      if (tid == 0) s_action = stop;
    }
    __syncthreads();
  }
  if (tid == 0) {
    for (int k = 0; k < 7; ++k) out->v[k] = s_x[k];
    out->v[7] = it; out->v[8] = converged; out->v[9] = usable;
    out->v[10] = initial_cost; out->v[11] = final_cost;
  }
}

template <int NT>
__global__ __launch_bounds__(NT) void pnp_refine_kernel(const double *__restrict__ Pl, const double *__restrict__ Pr, const ObsDev *__restrict__ obs, int n_obs_host,
                                                        const int *__restrict__ ctl, const double *__restrict__ start /*q,t*/, int max_iterations, double huber_delta,
                                                        RefineOut *__restrict__ out) {
  pnp_refine_body<NT>(Pl, Pr, obs, n_obs_host, ctl, start, max_iterations, huber_delta, out);
}

// ------------------------------------------------------------------------- fused solve glue
// rvec -> quaternion (base.cpp:274-280) and the residual-block list in the order base.cpp:291-356 adds it, on the device so that
// triangulation, RANSAC and the refinement run back to back.  One workgroup.
// The GATE (base.cpp:241-272: keep the prediction when solvePnPRansac failed, or when frame_count > IGNORE_FRAME_COUNT and the
// acceleration is too large) is NOT decided here since round 6: it is the only step of a frame's solve that needs the previous frame's
// pose, it costs three subtractions, and the host evaluates it in spvo_solve_wait from the prior it holds by then.  The device goes on
// as if the gate accepts whenever RANSAC found a model -- refinement included, which starts from the RANSAC pose, never from the prior --
// and the host discards what a rejecting gate makes void.  So a frame's whole chain can be enqueued before the previous frame's has been
// collected (spvo_solve_submit with `late_prior`).
// A submission whose prior IS known when it is made (`late_prior` = 0: the one-piece call, the synchronous call sequence) still has the gate
// evaluated here ([44] = 1), so that a frame the gate rejects does not run a refinement nobody will use; the wait then takes this decision.
//   hdr (doubles): [0..11] P_l, [12..23] P_r, [24..32] K, [33..38] prior rvec,tvec, [39] frame_count, [40] refinement_degree,
//                  [41] max_acceleration, [42] time_interval, [43] ignore_frame_count, [44] 1 = gate on the device (the prior fields are valid)
//   gate_out (doubles): [0..6] start q,t = the RANSAC pose   [7] refinement enqueued (pnp_ok, and the gate where it ran)   [8] pnp_ok   [10..15] RANSAC rvec, tvec
template <int NT>
__device__ __forceinline__ void solve_gate_build_body(const double *hdr, const double *ransac_result, const int *inliers, const float *xyz,
                                                      const float *xy_cl, const float *xy_cr, const float *xy_pl, const float *xy_pr,
                                                      const float *prev_xyz, const int *prev_valid, const int *prev_index, const float *prev_pts,
                                                      ObsDev *obs, int *ctl, double *gate_out) {
  __shared__ int s_scan[NT];
  __shared__ int s_run, s_base;
  const int tid = threadIdx.x;
  const int degree = (int)hdr[40];
  const int ninl = (int)ransac_result[7];
  if (tid == 0) {
    const bool ok = ransac_result[6] != 0;
    double r[3], t[3];
    for (int k = 0; k < 3; ++k) { r[k] = ransac_result[k]; t[k] = ransac_result[3 + k]; }   // (no model: the host substitutes its prior)
    int do_opt = ok ? 1 : 0;                                                                  // late prior: the gate itself is spvo_solve_wait_prior's (host)
    if (ok && hdr[44] != 0) {   // the prior was known at submit time (hdr[33..38], frame count hdr[39]): the gate right here, so that a rejected frame skips its refinement
      const double dx = t[0] - hdr[36], dy = t[1] - hdr[37], dz = t[2] - hdr[38];
      const double acc = sqrt(dx * dx + dy * dy + dz * dz) / hdr[42];
      if ((int)hdr[39] > (int)hdr[43] && acc > hdr[41]) do_opt = 0;
    }
    double q[4];
    rvec_to_quat(r, q);
    for (int k = 0; k < 4; ++k) gate_out[k] = q[k];
    for (int k = 0; k < 3; ++k) { gate_out[4 + k] = t[k]; gate_out[10 + k] = r[k]; gate_out[13 + k] = t[k]; }
    gate_out[7] = do_opt;
    gate_out[8] = ok ? 1 : 0;
    s_run = (do_opt && degree > 0) ? 1 : 0;
    s_base = 0;
  }
  __syncthreads();
  const int run = s_run;
  if (run) {
    for (int base = 0; base < ninl; base += NT) {
      const int k = base + tid;
      int vi = 0, cnt = 0, pv = 0;
      if (k < ninl) {
        vi = inliers[k];
        // previous-frame 3-D point of the correspondence (base.cpp:323-332): handed in by value (prev_xyz / prev_valid), or as an index
        // into the points the PREVIOUS solve of this context triangulated -- they are still on the device, the host need not have seen them
        pv = prev_index ? (prev_index[vi] >= 0) : ((prev_xyz && prev_valid) ? (prev_valid[vi] != 0) : 0);
        cnt = 1 + (degree >= 2 ? 1 : 0) + (pv ? ((degree >= 3 ? 1 : 0) + (degree >= 4 ? 1 : 0)) : 0);
      }
      s_scan[tid] = cnt;
      __syncthreads();
      for (int o = 1; o < NT; o <<= 1) {
        const int v = (tid >= o) ? s_scan[tid - o] : 0;
        __syncthreads();
        s_scan[tid] += v;
        __syncthreads();
      }
      int off = s_base + s_scan[tid] - cnt;
      if (k < ninl) {
        auto put = [&](const float *X, const float *uv, int cam, int inv) {
          ObsDev o;
          o.X[0] = X[0]; o.X[1] = X[1]; o.X[2] = X[2];
          o.uv[0] = uv[0]; o.uv[1] = uv[1];
          o.cam = cam; o.inverse = inv;
          obs[off++] = o;
        };
        put(xyz + 3 * vi, xy_pl + 2 * vi, 0, 0);
        if (degree >= 2) put(xyz + 3 * vi, xy_pr + 2 * vi, 1, 0);
        const float *Xp = pv ? (prev_index ? prev_pts + 3 * prev_index[vi] : prev_xyz + 3 * vi) : nullptr;
        if (pv && degree >= 3) put(Xp, xy_cl + 2 * vi, 0, 1);
        if (pv && degree >= 4) put(Xp, xy_cr + 2 * vi, 1, 1);
      }
      __syncthreads();
      if (tid == NT - 1) s_base += s_scan[NT - 1];
      __syncthreads();
    }
  }
  if (tid == 0) {
    ctl[0] = run;
    ctl[1] = run ? s_base : 0;
  }
}

// The fused solve's TAIL: everything behind the hypotheses in ONE launch of one workgroup -- selection + refit (K15), gating and the
// residual-block list, the Levenberg-Marquardt loop (K16), then the results into the call's pinned host buffers.  Three dependent launches
// were three times the wait for a CU beside the other streams' kernels (rounds 1-5: the chain's latency was the cycle time of the small
// engines' frame loop; since round 6 nothing in it needs the previous frame's pose and two frames' chains may be in flight).  The phases hand their results over through global memory exactly as the
// separate kernels do; __syncthreads() between them orders those writes within the workgroup.
struct SolveTailArgs {
  const double *hdr;           // the call's header (device copy): P_l, P_r, K [24..32], prior [33..38], ...
  const float *xyz, *xy_cl, *xy_cr, *xy_pl, *xy_pr, *prev_xyz;
  const int *prev_valid;
  const int *prev_index;       // or: index of each correspondence's previous-frame point in `prev_pts` (-1: none) ...
  const float *prev_pts;       // ... the points the previous solve of this context triangulated (its output block on the device)
  int n, iterations;
  double thr2;
  RansacWork w;                // result = the 40-double result block, inliers = behind the points in the output block
  ObsDev *obs;
  int *ctl;
  double *res;                 // [0..7] RANSAC, [8..23] gate, [24..] refinement
  int max_iterations;
  double huber_delta;
  const unsigned *d_o;         // points + inliers (device) -> h_o, `o_words` words
  unsigned *h_o;
  int o_words;
  double *h_res;
};

__device__ __forceinline__ void solve_tail_body(const SolveTailArgs &a) {
  constexpr int NT = SOLVE_TAIL_THREADS;
  ransac_select_body<NT>(a.hdr + 24, a.xyz, a.xy_pl, a.n, a.hdr + 33, a.iterations, a.thr2, a.w);
  __threadfence_block();
  __syncthreads();
  solve_gate_build_body<NT>(a.hdr, a.res, a.w.inliers, a.xyz, a.xy_cl, a.xy_cr, a.xy_pl, a.xy_pr, a.prev_xyz, a.prev_valid, a.prev_index, a.prev_pts, a.obs, a.ctl, a.res + 8);
  __threadfence_block();
  __syncthreads();
  pnp_refine_body<NT>(a.hdr, a.hdr + 12, a.obs, 0, a.ctl, a.res + 8, a.max_iterations, a.huber_delta, reinterpret_cast<RefineOut *>(a.res + 24));
  __threadfence_block();
  __syncthreads();   // (every thread comes back from the bodies: their early exits are workgroup-uniform)
  for (int k = threadIdx.x; k < a.o_words; k += NT) a.h_o[k] = a.d_o[k];
  if (threadIdx.x < 40) a.h_res[threadIdx.x] = reinterpret_cast<const volatile double *>(a.res)[threadIdx.x];
}

__global__ __launch_bounds__(SOLVE_TAIL_THREADS) void solve_tail_kernel(const SolveTailArgs a) { solve_tail_body(a); }

// Frame k + 1's hypotheses BESIDE frame k's tail, in one launch (round 6): workgroup 0 is the tail of the solve submitted before, the other
// workgroups run eight hypotheses each (one per wave) of the solve being submitted.  On one stream the chain of a frame was
// 6 + 35 + 90 us of dependent kernels -- longer than a frame of the small engines, whose loop it paced (NOTES.md round 6) -- and a second
// solver stream is not to be had (a fifth active stream slows the trunk by half): within one launch the two overlap, and the
// stream carries max(35, 90) us per frame.  The two halves touch disjoint buffer sets (spvo_solve.hip).
struct SolveHypArgs {
  const double *Kd; const float *xyz; const float *xy; int n; const double *prior; uint32_t seed; double thr2; RansacWork w; int iterations;
};
__global__ __launch_bounds__(SOLVE_TAIL_THREADS) void solve_hyp_tail_kernel(const SolveHypArgs h, const SolveTailArgs t) {
  if (blockIdx.x == 0) { solve_tail_body(t); return; }
  const int it = ((int)blockIdx.x - 1) * (SOLVE_TAIL_THREADS / 64) + (int)(threadIdx.x >> 6);
  if (it < h.iterations) ransac_hypothesis_body(h.Kd, h.xyz, h.xy, h.n, h.prior, h.seed, h.thr2, h.w, it, (int)(threadIdx.x & 63));
}

}  // namespace spvo
