// spvo_cpu.cpp -- CPU restatement of the hot path in plain C++17 + OpenMP (see spvo_cpu.h).
//
// TEST INFRASTRUCTURE: the compiled second oracle (tests/test_cpu_backend.py compares it with oracle/*.py) and the CPU
// timing baseline of bench.py.  The product library never links or loads it.
//
// "nn.cpp" = src/odml_visual_odometry/src/feature_detection_neural_network.cpp, "base.cpp" = .../feature_detection_base.cpp,
// "cost.hpp" = .../include/odml_visual_odometry/ceres_cost_function.hpp, "hpp" = .../feature_detection.hpp,
// "node.cpp" = .../visual_odometry_node.cpp.  Third-party semantics (cv::resize, BFMatcher, triangulatePoints, Ceres LM) are
// restated from their published behaviour exactly as oracle/*.py states them ("parity unpinned": DESIGN.md section 4).
#include "spvo_cpu.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <numeric>
#include <string>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

thread_local std::string g_err;
int fail(int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

// ------------------------------------------------------------------------------------------------ network plan
enum { OP_CONV = 1, OP_MAXPOOL = 2, OP_L2NORM = 3, OP_DWCONV = 4 };
enum { FLAG_RELU = 1, FLAG_POOL = 2, FLAG_BN = 4, FLAG_ADD = 8 };

struct Op {
  uint32_t type = 0, in = 0, out = 0, out_c_off = 0, cin = 0, in_c_off = 0, cout = 0, ksize = 0, flags = 0, residual = 0;
  std::vector<float> wpack;   // CONV: [co_block][ci][tap][8]; DWCONV: [c][9]
  std::vector<float> bias, bn_scale, bn_shift;
};
struct TensorInfo { uint32_t channels = 0, level = 0; };

// Activation tensor: channel planes with a zero border (1 row above / below, 1 column left, enough columns on the right for
// whole register-tile strips), so the convolution reads its halo without bounds checks.
struct Act {
  int C = 0, H = 0, W = 0, Hp = 0, Wp = 0;
  std::vector<float> v;
  void shape(int c, int h, int w) {
    C = c; H = h; W = w; Hp = h + 2; Wp = ((w + 63) / 64) * 64 + 2 + 64;
    v.assign((size_t)C * Hp * Wp, 0.f);
  }
  float *at(int c, int y, int x) { return v.data() + ((size_t)c * Hp + (y + 1)) * Wp + (x + 1); }
  const float *at(int c, int y, int x) const { return v.data() + ((size_t)c * Hp + (y + 1)) * Wp + (x + 1); }
};

// Register tile of the convolution: COB output channels x NV vectors of VW pixels, accumulators in registers (GCC vector
// extensions; the loops over o and j have constant bounds and unroll).  AVX-512: 6 x 4 x 16 (24 + 4 + 1 of 32 registers),
// AVX2: 4 x 3 x 8 (12 + 3 + 1 of 16).
#if defined(__AVX512F__)
constexpr int VW = 16, NV = 4, COB = 6;
#else
constexpr int VW = 8, NV = 3, COB = 4;
#endif
constexpr int XT = VW * NV;
typedef float vf __attribute__((vector_size(VW * 4)));
typedef float vfu __attribute__((vector_size(VW * 4), aligned(4)));

// conv (ksize 1 or 3, stride 1, pad ksize/2) + bias + epilogue flags, one image.  Output rows are produced at full
// resolution into `full` (a scratch tensor when a 2x2 max-pool follows).
template <int KS>
void conv_forward_k(const Op &op, const Act &in, Act &full, const Act *residual) {
  constexpr int ks = KS, taps = KS * KS, pad = KS / 2;   // compile-time: the tap loops unroll and the accumulators stay in registers
  const int cin = (int)op.cin, cout = (int)op.cout;
  const int H = in.H, W = in.W, nblk = (cout + COB - 1) / COB, nxt = (W + XT - 1) / XT;
  const bool relu = op.flags & FLAG_RELU, bn = op.flags & FLAG_BN, add = op.flags & FLAG_ADD;
#pragma omp parallel for collapse(2) schedule(dynamic, 2)
  for (int blk = 0; blk < nblk; ++blk)
    for (int y = 0; y < H; ++y) {
      const float *wb = op.wpack.data() + (size_t)blk * cin * taps * COB;
      for (int xt = 0; xt < nxt; ++xt) {
        const int x0 = xt * XT;
        vf acc[COB][NV];
        for (int o = 0; o < COB; ++o) {
          const int co = blk * COB + o;
          const float b = co < cout ? op.bias[co] : 0.f;
          for (int j = 0; j < NV; ++j) acc[o][j] = b - (vf){};
        }
        for (int ci = 0; ci < cin; ++ci) {
          const float *wci = wb + (size_t)ci * taps * COB;
#pragma GCC unroll 3
          for (int ky = 0; ky < ks; ++ky) {
            const float *row = in.at((int)op.in_c_off + ci, y + ky - pad, x0 - pad);
#pragma GCC unroll 3
            for (int kx = 0; kx < ks; ++kx) {
              const float *w = wci + (ky * ks + kx) * COB;
              vf src[NV];
              for (int j = 0; j < NV; ++j) src[j] = *reinterpret_cast<const vfu *>(row + kx + j * VW);
              for (int o = 0; o < COB; ++o) {
                const vf wv = w[o] - (vf){};
                for (int j = 0; j < NV; ++j) acc[o][j] += wv * src[j];
              }
            }
          }
        }
        const int xn = std::min(XT, W - x0);
        for (int o = 0; o < COB; ++o) {
          const int co = blk * COB + o;
          if (co >= cout) break;
          float tmp[XT];
          for (int j = 0; j < NV; ++j) std::memcpy(tmp + j * VW, &acc[o][j], sizeof(vf));
          float *dst = full.at((int)op.out_c_off + co, y, x0);
          for (int x = 0; x < xn; ++x) {
            float v = tmp[x];
            if (relu) v = v > 0.f ? v : 0.f;
            if (bn) { v = v * op.bn_scale[co] + op.bn_shift[co]; v = v > 0.f ? v : 0.f; }
            if (add) { v += *residual->at(co, y, x0 + x); v = v > 0.f ? v : 0.f; }
            dst[x] = v;
          }
        }
      }
    }
}

void conv_forward(const Op &op, const Act &in, Act &full, const Act *residual) {
  if (op.ksize == 3) conv_forward_k<3>(op, in, full, residual);
  else conv_forward_k<1>(op, in, full, residual);
}

void dwconv_forward(const Op &op, const Act &in, Act &full, const Act *residual) {
  const int C = (int)op.cout, H = in.H, W = in.W;
  const bool relu = op.flags & FLAG_RELU, bn = op.flags & FLAG_BN, add = op.flags & FLAG_ADD;
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < C; ++c)
    for (int y = 0; y < H; ++y) {
      const float *w = op.wpack.data() + (size_t)c * 9;
      float *dst = full.at((int)op.out_c_off + c, y, 0);
      for (int x = 0; x < W; ++x) {
        float v = op.bias[c];
        for (int ky = 0; ky < 3; ++ky) {
          const float *src = in.at((int)op.in_c_off + c, y + ky - 1, x - 1);
          v += w[ky * 3] * src[0] + w[ky * 3 + 1] * src[1] + w[ky * 3 + 2] * src[2];
        }
        if (relu) v = v > 0.f ? v : 0.f;
        if (bn) { v = v * op.bn_scale[c] + op.bn_shift[c]; v = v > 0.f ? v : 0.f; }
        if (add) { v += *residual->at(c, y, x); v = v > 0.f ? v : 0.f; }
        dst[x] = v;
      }
    }
}

void pool2(const Act &src, int c_off_src, int channels, Act &dst, int c_off_dst) {
  const int H = src.H / 2, W = src.W / 2;
#pragma omp parallel for collapse(2) schedule(static)
  for (int c = 0; c < channels; ++c)
    for (int y = 0; y < H; ++y) {
      const float *a = src.at(c_off_src + c, 2 * y, 0), *b = src.at(c_off_src + c, 2 * y + 1, 0);
      float *d = dst.at(c_off_dst + c, y, 0);
      for (int x = 0; x < W; ++x) d[x] = std::max(std::max(a[2 * x], a[2 * x + 1]), std::max(b[2 * x], b[2 * x + 1]));
    }
}

// ------------------------------------------------------------------------------------------------ small linear algebra (f64)
using V3 = double[3];
struct Quat { double x, y, z, w; };

Quat quat_mul(const Quat &a, const Quat &b) {
  return Quat{a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
              a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}
void quat_to_rot(const Quat &q, double R[9]) {   // Eigen::Quaternion::toRotationMatrix, no normalisation
  const double tx = 2 * q.x, ty = 2 * q.y, tz = 2 * q.z;
  const double twx = tx * q.w, twy = ty * q.w, twz = tz * q.w, txx = tx * q.x, txy = ty * q.x, txz = tz * q.x;
  const double tyy = ty * q.y, tyz = tz * q.y, tzz = tz * q.z;
  R[0] = 1 - (tyy + tzz); R[1] = txy - twz;       R[2] = txz + twy;
  R[3] = txy + twz;       R[4] = 1 - (txx + tzz); R[5] = tyz - twx;
  R[6] = txz - twy;       R[7] = tyz + twx;       R[8] = 1 - (txx + tyy);
}
Quat rvec_to_quat(const double r[3]) {   // base.cpp:274-278
  const double angle = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
  double ax[3] = {r[0], r[1], r[2]};
  if (angle > 0) for (double &v : ax) v /= angle;
  const double s = std::sin(angle / 2);
  return Quat{ax[0] * s, ax[1] * s, ax[2] * s, std::cos(angle / 2)};
}
void quat_to_rvec(Quat q, double r[3]) {
  if (q.w < 0) q = Quat{-q.x, -q.y, -q.z, -q.w};
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z);
  if (n < 1e-300) { r[0] = r[1] = r[2] = 0; return; }
  const double angle = 2 * std::atan2(n, q.w);
  r[0] = q.x / n * angle; r[1] = q.y / n * angle; r[2] = q.z / n * angle;
}

// Gaussian elimination with partial pivoting (what LAPACK's gesv does for these 6x6 systems); false = singular / not finite
template <int N>
bool solve_lu(double A[N][N], double b[N], double x[N]) {
  for (int k = 0; k < N; ++k) {
    int p = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < N; ++i)
      if (std::fabs(A[i][k]) > best) { best = std::fabs(A[i][k]); p = i; }
    if (!(best > 0) || !std::isfinite(best)) return false;
    if (p != k) { for (int j = 0; j < N; ++j) std::swap(A[k][j], A[p][j]); std::swap(b[k], b[p]); }
    for (int i = k + 1; i < N; ++i) {
      const double f = A[i][k] / A[k][k];
      A[i][k] = 0;
      for (int j = k + 1; j < N; ++j) A[i][j] -= f * A[k][j];
      b[i] -= f * b[k];
    }
  }
  for (int i = N - 1; i >= 0; --i) {
    double s = b[i];
    for (int j = i + 1; j < N; ++j) s -= A[i][j] * x[j];
    x[i] = s / A[i][i];
  }
  for (int i = 0; i < N; ++i)
    if (!std::isfinite(x[i])) return false;
  return true;
}

bool cholesky_solve6(const double A[6][6], const double b[6], double x[6]) {   // A x = b, A symmetric positive definite
  double L[6][6] = {};
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j <= i; ++j) {
      double s = A[i][j];
      for (int k = 0; k < j; ++k) s -= L[i][k] * L[j][k];
      if (i == j) {
        if (!(s > 0) || !std::isfinite(s)) return false;
        L[i][i] = std::sqrt(s);
      } else {
        L[i][j] = s / L[j][j];
      }
    }
  double y[6];
  for (int i = 0; i < 6; ++i) {
    double s = b[i];
    for (int k = 0; k < i; ++k) s -= L[i][k] * y[k];
    y[i] = s / L[i][i];
  }
  for (int i = 5; i >= 0; --i) {
    double s = y[i];
    for (int k = i + 1; k < 6; ++k) s -= L[k][i] * x[k];
    x[i] = s / L[i][i];
  }
  return true;
}

// right singular vector of the smallest singular value of a 4x4 matrix: one-sided Jacobi (Hestenes) on the columns
void null_vector4(const double Ain[4][4], double out[4]) {
  double A[4][4], V[4][4] = {};
  for (int i = 0; i < 4; ++i) { V[i][i] = 1; for (int j = 0; j < 4; ++j) A[i][j] = Ain[i][j]; }
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0;
    for (int p = 0; p < 3; ++p)
      for (int q = p + 1; q < 4; ++q) {
        double alpha = 0, beta = 0, gamma = 0;
        for (int i = 0; i < 4; ++i) { alpha += A[i][p] * A[i][p]; beta += A[i][q] * A[i][q]; gamma += A[i][p] * A[i][q]; }
        if (gamma == 0) continue;
        off = std::max(off, std::fabs(gamma) / std::sqrt(std::max(alpha * beta, 1e-300)));
        const double zeta = (beta - alpha) / (2 * gamma);
        const double t = (zeta >= 0 ? 1.0 : -1.0) / (std::fabs(zeta) + std::sqrt(1 + zeta * zeta));
        const double c = 1 / std::sqrt(1 + t * t), s = c * t;
        for (int i = 0; i < 4; ++i) {
          const double ap = A[i][p], aq = A[i][q];
          A[i][p] = c * ap - s * aq; A[i][q] = s * ap + c * aq;
          const double vp = V[i][p], vq = V[i][q];
          V[i][p] = c * vp - s * vq; V[i][q] = s * vp + c * vq;
        }
      }
    if (off < 1e-15) break;
  }
  int best = 0;
  double bn = std::numeric_limits<double>::infinity();
  for (int j = 0; j < 4; ++j) {
    double n = 0;
    for (int i = 0; i < 4; ++i) n += A[i][j] * A[i][j];
    if (n < bn) { bn = n; best = j; }
  }
  for (int i = 0; i < 4; ++i) out[i] = V[i][best];
}

// ------------------------------------------------------------------------------------------------ PnP pieces (base.cpp:227-375)
uint32_t hash32(uint32_t x) {
  x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16;
  return x;
}
void sample_triplet(uint32_t seed, uint32_t it, int n, int idx[3]) {
  for (int k = 0; k < 3; ++k) {
    for (uint32_t attempt = 0;; ++attempt) {
      const int r = (int)(hash32(seed * 0x9E3779B9u + it * 0x85EBCA6Bu + (uint32_t)k * 0xC2B2AE35u + attempt * 0x27D4EB2Fu) % (uint32_t)n);
      bool dup = false;
      for (int j = 0; j < k; ++j) dup |= idx[j] == r;
      if (!dup) { idx[k] = r; break; }
    }
  }
}

// residual (2) and Jacobian (2x6) of K (R X + t) wrt a left-multiplied small rotation and t
void project_jac(const double K[9], const Quat &q, const double t[3], const double X[3], const double uv[2], double r[2], double J[2][6]) {
  double R[9];
  quat_to_rot(q, R);
  const double Y[3] = {R[0] * X[0] + R[1] * X[1] + R[2] * X[2], R[3] * X[0] + R[4] * X[1] + R[5] * X[2], R[6] * X[0] + R[7] * X[1] + R[8] * X[2]};
  const double Xc[3] = {Y[0] + t[0], Y[1] + t[1], Y[2] + t[2]};
  const double p[3] = {K[0] * Xc[0] + K[1] * Xc[1] + K[2] * Xc[2], K[3] * Xc[0] + K[4] * Xc[1] + K[5] * Xc[2], K[6] * Xc[0] + K[7] * Xc[1] + K[8] * Xc[2]};
  const double u = p[0] / p[2], v = p[1] / p[2];
  double du[3], dv[3];
  for (int k = 0; k < 3; ++k) { du[k] = (K[k] - u * K[6 + k]) / p[2]; dv[k] = (K[3 + k] - v * K[6 + k]) / p[2]; }
  const double S[3][3] = {{0, -Y[2], Y[1]}, {Y[2], 0, -Y[0]}, {-Y[1], Y[0], 0}};   // d Xc / d theta = -[Y]x
  for (int k = 0; k < 3; ++k) {
    J[0][k] = -(du[0] * S[0][k] + du[1] * S[1][k] + du[2] * S[2][k]);
    J[1][k] = -(dv[0] * S[0][k] + dv[1] * S[1][k] + dv[2] * S[2][k]);
    J[0][3 + k] = du[k];
    J[1][3 + k] = dv[k];
  }
  r[0] = u - uv[0];
  r[1] = v - uv[1];
}
void apply_delta(Quat &q, double t[3], const double d[6]) {
  Quat qn = quat_mul(Quat{d[0] / 2, d[1] / 2, d[2] / 2, 1.0}, q);
  const double n = std::sqrt(qn.x * qn.x + qn.y * qn.y + qn.z * qn.z + qn.w * qn.w);
  q = Quat{qn.x / n, qn.y / n, qn.z / n, qn.w / n};
  for (int k = 0; k < 3; ++k) t[k] += d[3 + k];
}
bool minimal_solve(const double K[9], const double X3[3][3], const double uv3[3][2], Quat &q, double t[3]) {
  bool ok = false;
  for (int iter = 0; iter < 10; ++iter) {
    double f[6], J[6][6];
    for (int i = 0; i < 3; ++i) {
      double r[2], Ji[2][6];
      project_jac(K, q, t, X3[i], uv3[i], r, Ji);
      f[2 * i] = r[0]; f[2 * i + 1] = r[1];
      for (int k = 0; k < 6; ++k) { J[2 * i][k] = Ji[0][k]; J[2 * i + 1][k] = Ji[1][k]; }
    }
    double fmax = 0;
    for (int i = 0; i < 6; ++i) {
      if (!std::isfinite(f[i])) return false;
      for (int k = 0; k < 6; ++k) if (!std::isfinite(J[i][k])) return false;
      fmax = std::max(fmax, std::fabs(f[i]));
    }
    if (fmax < 1e-9) { ok = true; break; }
    double nb[6], d[6];
    for (int i = 0; i < 6; ++i) nb[i] = -f[i];
    if (!solve_lu<6>(J, nb, d)) return false;
    double dmax = 0;
    for (double v : d) dmax = std::max(dmax, std::fabs(v));
    if (dmax > 1e3) return false;
    apply_delta(q, t, d);
  }
  if (!ok) {
    double fmax = 0;
    for (int i = 0; i < 3; ++i) {
      double r[2], Ji[2][6];
      project_jac(K, q, t, X3[i], uv3[i], r, Ji);
      if (!std::isfinite(r[0]) || !std::isfinite(r[1])) return false;
      fmax = std::max(fmax, std::max(std::fabs(r[0]), std::fabs(r[1])));
    }
    ok = fmax < 1e-6;
  }
  return ok;
}

int pnp_ransac(const double K[9], const float *xyz, const float *xy, int n, int iterations, double thr, uint32_t seed, double rvec[3], double tvec[3],
               std::vector<int32_t> &inliers, bool &ok) {
  inliers.clear();
  ok = false;
  if (n < 4) return 0;
  // prior-free minimal solver (oracle/odometry.py: pnp_ransac): every sample's Newton iteration starts at the identity; rvec / tvec
  // (the motion prior) are only what stays in place when no model is found
  const Quat q0{0.0, 0.0, 0.0, 1.0};
  const double t0[3] = {0.0, 0.0, 0.0};
  int best_count = -1;
  Quat best_q = q0;
  double best_t[3] = {t0[0], t0[1], t0[2]};
  std::vector<char> mask(n), best_mask(n, 0);
  for (int it = 0; it < iterations; ++it) {
    int s[3];
    sample_triplet(seed, (uint32_t)it, n, s);
    double X3[3][3], uv3[3][2];
    for (int i = 0; i < 3; ++i) {
      for (int k = 0; k < 3; ++k) X3[i][k] = (double)xyz[3 * s[i] + k];
      for (int k = 0; k < 2; ++k) uv3[i][k] = (double)xy[2 * s[i] + k];
    }
    Quat q = q0;
    double t[3] = {t0[0], t0[1], t0[2]};
    if (!minimal_solve(K, X3, uv3, q, t)) continue;
    double R[9];
    quat_to_rot(q, R);
    int count = 0;
    for (int i = 0; i < n; ++i) {
      const double X[3] = {(double)xyz[3 * i], (double)xyz[3 * i + 1], (double)xyz[3 * i + 2]};
      const double Xc[3] = {R[0] * X[0] + R[1] * X[1] + R[2] * X[2] + t[0], R[3] * X[0] + R[4] * X[1] + R[5] * X[2] + t[1],
                            R[6] * X[0] + R[7] * X[1] + R[8] * X[2] + t[2]};
      const double p0 = K[0] * Xc[0] + K[1] * Xc[1] + K[2] * Xc[2], p1 = K[3] * Xc[0] + K[4] * Xc[1] + K[5] * Xc[2],
                   p2 = K[6] * Xc[0] + K[7] * Xc[1] + K[8] * Xc[2];
      const double du = p0 / p2 - (double)xy[2 * i], dv = p1 / p2 - (double)xy[2 * i + 1];
      const double e2 = du * du + dv * dv;
      mask[i] = (p2 > 0) && (e2 <= thr * thr);
      count += mask[i];
    }
    if (count > best_count) { best_count = count; best_q = q; for (int k = 0; k < 3; ++k) best_t[k] = t[k]; best_mask = mask; }
  }
  if (best_count < 4) return 0;
  for (int i = 0; i < n; ++i) if (best_mask[i]) inliers.push_back(i);
  // Gauss-Newton refit on the inliers of the best minimal model
  Quat q = best_q;
  double t[3] = {best_t[0], best_t[1], best_t[2]};
  for (int iter = 0; iter < 10; ++iter) {
    double A[6][6] = {}, g[6] = {};
    for (int idx : inliers) {
      const double X[3] = {(double)xyz[3 * idx], (double)xyz[3 * idx + 1], (double)xyz[3 * idx + 2]};
      const double uv[2] = {(double)xy[2 * idx], (double)xy[2 * idx + 1]};
      double r[2], J[2][6];
      project_jac(K, q, t, X, uv, r, J);
      for (int a = 0; a < 6; ++a) {
        for (int b = 0; b < 6; ++b) A[a][b] += J[0][a] * J[0][b] + J[1][a] * J[1][b];
        g[a] += J[0][a] * r[0] + J[1][a] * r[1];
      }
    }
    double nb[6], d[6];
    for (int i = 0; i < 6; ++i) nb[i] = -g[i];
    if (!solve_lu<6>(A, nb, d)) break;
    apply_delta(q, t, d);
    double dmax = 0;
    for (double v : d) dmax = std::max(dmax, std::fabs(v));
    if (dmax < 1e-10) break;
  }
  quat_to_rvec(q, rvec);
  for (int k = 0; k < 3; ++k) tvec[k] = t[k];
  ok = true;
  return (int)inliers.size();
}

// cost.hpp:27-58 with an analytic Jacobian in the local parameterisation (3 rotation via EigenQuaternionParameterization + 3 translation)
struct ObsD { double X[3], uv[2]; int cam, inv; };

void drot(const Quat &q, double dR[4][9]) {
  const double x = q.x, y = q.y, z = q.z, w = q.w;
  const double dx[9] = {0, 2 * y, 2 * z, 2 * y, -4 * x, -2 * w, 2 * z, 2 * w, -4 * x};
  const double dy[9] = {-4 * y, 2 * x, 2 * w, 2 * x, 0, 2 * z, -2 * w, 2 * z, -4 * y};
  const double dz[9] = {-4 * z, -2 * w, 2 * x, 2 * w, -4 * z, 2 * y, 2 * x, 2 * y, 0};
  const double dw[9] = {0, -2 * z, 2 * y, 2 * z, 0, -2 * x, -2 * y, 2 * x, 0};
  std::memcpy(dR[0], dx, sizeof dx); std::memcpy(dR[1], dy, sizeof dy); std::memcpy(dR[2], dz, sizeof dz); std::memcpy(dR[3], dw, sizeof dw);
}

// cost = 1/2 sum rho(|r|^2); with want_jac also A = J^T J and g = J^T r of the Huber-re-weighted problem
double lm_evaluate(const double Pl[12], const double Pr[12], const std::vector<ObsD> &obs, const Quat &q, const double t[3], double delta, bool want_jac,
                   double A[6][6], double g[6]) {
  double R[9], dR[4][9];
  quat_to_rot(q, R);
  if (want_jac) {
    drot(q, dR);
    for (int a = 0; a < 6; ++a) { g[a] = 0; for (int b = 0; b < 6; ++b) A[a][b] = 0; }
  }
  const double G[4][3] = {{q.w, q.z, -q.y}, {-q.z, q.w, q.x}, {q.y, -q.x, q.w}, {-q.x, -q.y, -q.z}};   // plus Jacobian
  const double b2 = delta * delta;
  double cost = 0;
  for (const ObsD &o : obs) {
    const double *P = o.cam ? Pr : Pl;
    double wv[3], Xt[3];
    if (!o.inv) {
      for (int k = 0; k < 3; ++k) wv[k] = o.X[k];
      for (int i = 0; i < 3; ++i) Xt[i] = R[3 * i] * wv[0] + R[3 * i + 1] * wv[1] + R[3 * i + 2] * wv[2] + t[i];
    } else {
      for (int k = 0; k < 3; ++k) wv[k] = o.X[k] - t[k];
      for (int i = 0; i < 3; ++i) Xt[i] = R[i] * wv[0] + R[3 + i] * wv[1] + R[6 + i] * wv[2];   // R^T (X - t)
    }
    double p[3];
    for (int i = 0; i < 3; ++i) p[i] = P[4 * i] * Xt[0] + P[4 * i + 1] * Xt[1] + P[4 * i + 2] * Xt[2] + P[4 * i + 3];
    const double u = p[0] / p[2], v = p[1] / p[2];
    const double r0 = u - o.uv[0], r1 = v - o.uv[1];
    const double s = r0 * r0 + r1 * r1;
    const bool big = s > b2;
    const double sq = std::sqrt(big ? s : 1.0);
    cost += big ? 2 * delta * sq - b2 : s;
    if (!want_jac) continue;
    const double rho1 = big ? delta / sq : 1.0, wgt = std::sqrt(rho1);
    double du[3], dv[3];
    for (int k = 0; k < 3; ++k) { du[k] = (P[k] - u * P[8 + k]) / p[2]; dv[k] = (P[4 + k] - v * P[8 + k]) / p[2]; }
    double Jq[2][4];
    for (int k = 0; k < 4; ++k) {
      double dXt[3];
      for (int i = 0; i < 3; ++i)
        dXt[i] = !o.inv ? dR[k][3 * i] * wv[0] + dR[k][3 * i + 1] * wv[1] + dR[k][3 * i + 2] * wv[2]
                        : dR[k][i] * wv[0] + dR[k][3 + i] * wv[1] + dR[k][6 + i] * wv[2];
      Jq[0][k] = du[0] * dXt[0] + du[1] * dXt[1] + du[2] * dXt[2];
      Jq[1][k] = dv[0] * dXt[0] + dv[1] * dXt[1] + dv[2] * dXt[2];
    }
    double J[2][6];
    for (int c = 0; c < 3; ++c) {
      J[0][c] = Jq[0][0] * G[0][c] + Jq[0][1] * G[1][c] + Jq[0][2] * G[2][c] + Jq[0][3] * G[3][c];
      J[1][c] = Jq[1][0] * G[0][c] + Jq[1][1] * G[1][c] + Jq[1][2] * G[2][c] + Jq[1][3] * G[3][c];
      if (!o.inv) { J[0][3 + c] = du[c]; J[1][3 + c] = dv[c]; }
      else {   // d/dt of R^T (X - t) = -R^T
        J[0][3 + c] = -(du[0] * R[3 * c] + du[1] * R[3 * c + 1] + du[2] * R[3 * c + 2]);
        J[1][3 + c] = -(dv[0] * R[3 * c] + dv[1] * R[3 * c + 1] + dv[2] * R[3 * c + 2]);
      }
    }
    const double rw0 = r0 * wgt, rw1 = r1 * wgt;
    for (int a = 0; a < 6; ++a) {
      const double ja0 = J[0][a] * wgt, ja1 = J[1][a] * wgt;
      for (int b = 0; b < 6; ++b) A[a][b] += ja0 * (J[0][b] * wgt) + ja1 * (J[1][b] * wgt);
      g[a] += ja0 * rw0 + ja1 * rw1;
    }
  }
  return 0.5 * cost;
}

Quat quat_plus(const Quat &q, const double d[3]) {   // EigenQuaternionParameterization::Plus
  const double nd = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
  if (!(nd > 0)) return q;
  const double s = std::sin(nd) / nd;
  return quat_mul(Quat{s * d[0], s * d[1], s * d[2], std::cos(nd)}, q);
}

// Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy as oracle/odometry.py::pnp_refine restates them
void pnp_refine(const double Pl[12], const double Pr[12], const std::vector<ObsD> &obs, int max_iterations, double delta, Quat &q, double t[3],
                spvo_cpu_refine_summary &S) {
  S = spvo_cpu_refine_summary{0, 0, 0, 0.0, 0.0};
  if (obs.empty()) { S.converged = S.usable = 1; return; }
  double A[6][6], g[6];
  double cost = lm_evaluate(Pl, Pr, obs, q, t, delta, true, A, g);
  S.initial_cost = S.final_cost = cost;
  if (!std::isfinite(cost)) return;
  double scale[6];
  for (int i = 0; i < 6; ++i) scale[i] = 1.0 / (1.0 + std::sqrt(A[i][i]));   // Jacobi scaling, fixed at iteration 0
  double radius = 1e4, decrease = 2.0;
  int invalid = 0;
  auto xnorm = [&]() { return std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w + t[0] * t[0] + t[1] * t[1] + t[2] * t[2]); };
  double x_norm = xnorm();
  S.usable = 1;
  auto gmax = [&]() { double m = 0; for (double v : g) m = std::max(m, std::fabs(v)); return m; };
  if (gmax() <= 1e-10) { S.converged = 1; return; }
  int it = 0;
  for (;;) {
    if (it >= max_iterations) break;
    ++it;
    S.iterations = it;
    double As[6][6], gs[6], M[6][6], ds[6];
    for (int a = 0; a < 6; ++a) { gs[a] = g[a] * scale[a]; for (int b = 0; b < 6; ++b) As[a][b] = A[a][b] * scale[a] * scale[b]; }
    for (int a = 0; a < 6; ++a) for (int b = 0; b < 6; ++b) M[a][b] = As[a][b];
    for (int a = 0; a < 6; ++a) M[a][a] += std::min(std::max(As[a][a], 1e-6), 1e32) / radius;
    double ngs[6];
    for (int a = 0; a < 6; ++a) ngs[a] = -gs[a];
    bool solved = cholesky_solve6(M, ngs, ds);
    double model_change = 0;
    if (solved) {
      double gd = 0, quad = 0;
      for (int a = 0; a < 6; ++a) {
        gd += gs[a] * ds[a];
        double row = 0;
        for (int b = 0; b < 6; ++b) row += As[a][b] * ds[b];
        quad += ds[a] * row;
        solved = solved && std::isfinite(ds[a]);
      }
      model_change = -(gd + 0.5 * quad);
    }
    if (!solved || !(model_change > 0)) {
      if (++invalid >= 5) { S.usable = 0; break; }
      radius /= decrease;
      decrease *= 2;
      continue;
    }
    invalid = 0;
    double d[6];
    for (int a = 0; a < 6; ++a) d[a] = ds[a] * scale[a];
    const Quat qc = quat_plus(q, d);
    const double tc[3] = {t[0] + d[3], t[1] + d[4], t[2] + d[5]};
    const double cand = lm_evaluate(Pl, Pr, obs, qc, tc, delta, false, nullptr, nullptr);
    const double step_norm = std::sqrt((qc.x - q.x) * (qc.x - q.x) + (qc.y - q.y) * (qc.y - q.y) + (qc.z - q.z) * (qc.z - q.z) + (qc.w - q.w) * (qc.w - q.w) +
                                       (tc[0] - t[0]) * (tc[0] - t[0]) + (tc[1] - t[1]) * (tc[1] - t[1]) + (tc[2] - t[2]) * (tc[2] - t[2]));
    if (step_norm <= 1e-8 * (x_norm + 1e-8)) { S.converged = 1; break; }
    const double cost_change = cost - cand;
    if (std::fabs(cost_change) <= 1e-6 * cost) { S.converged = 1; break; }
    const double rel = cost_change / model_change;
    if (std::isfinite(cand) && rel > 1e-3) {
      q = qc;
      for (int k = 0; k < 3; ++k) t[k] = tc[k];
      cost = lm_evaluate(Pl, Pr, obs, q, t, delta, true, A, g);
      x_norm = xnorm();
      const double c3 = 2.0 * rel - 1.0;
      radius = std::min(1e16, radius / std::max(1.0 / 3.0, 1.0 - c3 * c3 * c3));
      decrease = 2.0;
      S.final_cost = cost;
      if (gmax() <= 1e-10) { S.converged = 1; break; }
      if (radius < 1e-32) { S.converged = 1; break; }
    } else {
      radius /= decrease;
      decrease *= 2;
    }
  }
}

double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// ------------------------------------------------------------------------------------------------ context
struct spvo_cpu {
  spvo_cpu_config cfg;
  int threads = 1;
  bool weights = false;
  std::vector<TensorInfo> tensors;
  std::vector<Op> ops;
  uint32_t t_input = 0, t_det = 0, t_desc = 0;
  std::vector<Act> acts;      // one image at a time
  Act scratch;                // full-resolution output of a conv with a fused pool
  // front-end state (hpp:96-178)
  int selector = 1, cross_check = 0, refinement_degree = 4;
  bool classic = false;                              // ORB front end (BASELINE config 1): native resolution, Hamming descriptors
  std::vector<std::vector<uint8_t>> bdesc_dq;        // its 32-byte descriptors, parallel to kp_dq
  float stereo_threshold = 2.f, min_disparity = 0.25f;
  std::vector<std::vector<float>> kp_dq, desc_dq;   // last 4: prevL, prevR, currL, currR
  std::vector<int32_t> maps[3];
  double P_l[12] = {}, P_r[12] = {}, r_pred[3] = {}, t_pred[3] = {};
  int frame_count = 0;
  bool prev_pts_inited = false;
  std::vector<float> prev_pts3d;
  std::vector<int32_t> prev_matched_to_valid;
};

extern "C" {

const char *spvo_cpu_last_error(void) { return g_err.c_str(); }

void spvo_cpu_default_config(spvo_cpu_config *cfg) {
  if (!cfg) return;
  *cfg = spvo_cpu_config{360, 1176, 0.015f, 4, 4, 1000, 1, 0};
}

int spvo_cpu_create(const spvo_cpu_config *cfg, spvo_cpu **out) {
  if (!cfg || !out) return fail(-1, "null argument");
  if (cfg->net_height <= 0 || cfg->net_width <= 0 || cfg->net_height % 8 || cfg->net_width % 8) return fail(-1, "net size must be positive multiples of 8");
  spvo_cpu *c = new spvo_cpu;
  c->cfg = *cfg;
#ifdef _OPENMP
  c->threads = cfg->num_threads > 0 ? cfg->num_threads : omp_get_max_threads();
  omp_set_num_threads(c->threads);
#endif
  *out = c;
  return 0;
}

void spvo_cpu_destroy(spvo_cpu *c) { delete c; }
int spvo_cpu_threads(const spvo_cpu *c) { return c ? c->threads : 0; }

int spvo_cpu_load_weights(spvo_cpu *c, const char *path) {
  if (!c || !path) return fail(-1, "null argument");
  FILE *f = std::fopen(path, "rb");
  if (!f) return fail(-3, "no such engine file: %s", path);
  std::fseek(f, 0, SEEK_END);
  const long size = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> buf((size_t)std::max(size, 0L));
  const bool read_ok = size > 0 && std::fread(buf.data(), 1, buf.size(), f) == buf.size();
  std::fclose(f);
  if (!read_ok || buf.size() < 40 || std::memcmp(buf.data(), "SPVW0003", 8) != 0) return fail(-3, "%s is not an SPVW0003 file", path);
  auto u32 = [&](size_t off) { uint32_t v; std::memcpy(&v, buf.data() + off, 4); return v; };
  auto u64 = [&](size_t off) { uint64_t v; std::memcpy(&v, buf.data() + off, 8); return v; };
  const uint32_t nt = u32(8), no = u32(12);
  if (u32(28) != 0) return fail(-3, "the CPU restatement runs FP32 engines only");
  size_t pos = 40;
  if (nt > 4096 || no > 4096 || buf.size() < pos + (size_t)nt * 8 + (size_t)no * 72 + 8) return fail(-3, "truncated plan");
  std::vector<TensorInfo> tensors(nt);
  for (uint32_t i = 0; i < nt; ++i, pos += 8) tensors[i] = TensorInfo{u32(pos), u32(pos + 4)};
  struct Rec { uint32_t w[12]; uint64_t off[3]; };
  std::vector<Rec> recs(no);
  for (uint32_t i = 0; i < no; ++i, pos += 72) {
    for (int k = 0; k < 12; ++k) recs[i].w[k] = u32(pos + 4 * k);
    for (int k = 0; k < 3; ++k) recs[i].off[k] = u64(pos + 48 + 8 * k);
  }
  const uint64_t nfl = u64(pos);
  pos += 8;
  if (nfl > (buf.size() - pos) / 4) return fail(-3, "truncated payload");
  const float *payload = reinterpret_cast<const float *>(buf.data() + pos);
  std::vector<Op> ops(no);
  for (uint32_t i = 0; i < no; ++i) {
    const Rec &r = recs[i];
    Op &op = ops[i];
    op.type = r.w[0]; op.in = r.w[1]; op.out = r.w[2]; op.out_c_off = r.w[3]; op.cin = r.w[4] & 0xFFFF; op.in_c_off = r.w[4] >> 16;
    op.cout = r.w[5]; op.ksize = r.w[6]; op.flags = r.w[7]; op.residual = r.w[8];
    if (op.in >= nt || op.out >= nt || op.residual >= nt) return fail(-3, "op %u: tensor id out of range", i);
    if (op.type == OP_CONV || op.type == OP_DWCONV) {
      const int taps = (int)(op.ksize * op.ksize);
      const size_t nw = op.type == OP_CONV ? (size_t)op.cout * op.cin * taps : (size_t)op.cout * 9;
      if ((op.ksize != 1 && op.ksize != 3) || r.off[0] + nw > nfl || r.off[1] + op.cout > nfl) return fail(-3, "op %u: bad weights", i);
      const float *w = payload + r.off[0];
      op.bias.assign(payload + r.off[1], payload + r.off[1] + op.cout);
      if (op.type == OP_CONV) {   // OIHW -> [co block][ci][tap][8]
        const int nblk = ((int)op.cout + COB - 1) / COB;
        op.wpack.assign((size_t)nblk * op.cin * taps * COB, 0.f);
        for (uint32_t co = 0; co < op.cout; ++co)
          for (uint32_t ci = 0; ci < op.cin; ++ci)
            for (int tp = 0; tp < taps; ++tp)
              op.wpack[(((size_t)(co / COB) * op.cin + ci) * taps + tp) * COB + co % COB] = w[((size_t)co * op.cin + ci) * taps + tp];
      } else {
        op.wpack.assign(w, w + nw);
      }
      if (op.flags & FLAG_BN) {   // ONNX BatchNormalization, inference: (x - mean) / sqrt(var + eps) * gamma + beta
        const size_t c = op.cout;
        if (r.off[2] + 4 * c + 1 > nfl) return fail(-3, "op %u: bad BatchNorm block", i);
        const float *bn = payload + r.off[2];
        op.bn_scale.resize(c); op.bn_shift.resize(c);
        for (size_t k = 0; k < c; ++k) {
          const float inv = 1.0f / std::sqrt(bn[3 * c + k] + bn[4 * c]);
          op.bn_scale[k] = bn[k] * inv;
          op.bn_shift[k] = bn[c + k] - bn[2 * c + k] * bn[k] * inv;
        }
      }
    }
  }
  c->tensors = tensors; c->ops = ops;
  c->t_input = u32(16); c->t_det = u32(20); c->t_desc = u32(24);
  if (c->t_input >= nt || c->t_det >= nt || c->t_desc >= nt) return fail(-3, "binding tensor id out of range");
  c->acts.assign(nt, Act());
  for (uint32_t i = 0; i < nt; ++i) c->acts[i].shape((int)tensors[i].channels, c->cfg.net_height >> tensors[i].level, c->cfg.net_width >> tensors[i].level);
  c->weights = true;
  return 0;
}

// ------------------------------------------------------------------------------------------------ preprocessing (base.cpp:68-121)
static void linear_coeffs(int dst, int src, std::vector<int> &idx, std::vector<int> &a0, std::vector<int> &a1) {
  idx.resize(dst); a0.resize(dst); a1.resize(dst);
  const double scale = (double)src / (double)dst;
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f = f - (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src - 1) { s = src - 1; f = 0.f; }
    idx[d] = s;
    a0[d] = (int)std::nearbyintf((1.0f - f) * 2048.f);   // saturate_cast<short>: round to nearest even
    a1[d] = (int)std::nearbyintf(f * 2048.f);
  }
}

#include "orb_cpu.inc"

int spvo_cpu_preprocess(spvo_cpu *c, const uint8_t *img, int rows, int cols, size_t stride, double P[12], uint8_t *resized) {
  if (!c || !img || !P || rows <= 0 || cols <= 0) return fail(-1, "bad argument");
  const int H = c->cfg.net_height, W = c->cfg.net_width;
  const float real = (float)cols / (float)rows, expected = (float)W / (float)H;
  int crop_rows = rows, crop_cols = cols, row_off = 0, col_off = 0;
  if (expected > real) { crop_rows = (int)((float)cols / expected); row_off = (rows - crop_rows) / 2; }
  else if (expected < real) { crop_cols = (int)((float)rows * expected); col_off = (cols - crop_cols) / 2; }
  const float scale = (float)W / (float)crop_cols;
  if (c->cfg.bug_compat_p) {   // base.cpp:95,111: at<float>(r, 2) on a CV_64F matrix = the low half of P[r][1]
    float lo;
    if (crop_rows != rows) { std::memcpy(&lo, reinterpret_cast<char *>(&P[5]), 4); lo -= (float)row_off; std::memcpy(reinterpret_cast<char *>(&P[5]), &lo, 4); }
    else if (crop_cols != cols) { std::memcpy(&lo, reinterpret_cast<char *>(&P[1]), 4); lo -= (float)col_off; std::memcpy(reinterpret_cast<char *>(&P[1]), &lo, 4); }
  } else {
    if (crop_rows != rows) P[6] -= (double)(float)row_off;
    else if (crop_cols != cols) P[2] -= (double)(float)col_off;
  }
  for (int k = 0; k < 8; ++k) P[k] *= (double)scale;   // base.cpp:118-120
  if (!resized) return 0;
  const uint8_t *src = img + (size_t)row_off * stride + col_off;
  if (crop_rows == H && crop_cols == W) {
    for (int y = 0; y < H; ++y) std::memcpy(resized + (size_t)y * W, src + (size_t)y * stride, W);
    return 0;
  }
  std::vector<int> xi, xa0, xa1, yi, yb0, yb1;
  linear_coeffs(W, crop_cols, xi, xa0, xa1);
  linear_coeffs(H, crop_rows, yi, yb0, yb1);
#pragma omp parallel for schedule(static)
  for (int y = 0; y < H; ++y) {
    const uint8_t *r0 = src + (size_t)yi[y] * stride, *r1 = src + (size_t)std::min(yi[y] + 1, crop_rows - 1) * stride;
    for (int x = 0; x < W; ++x) {
      const int x0 = xi[x], x1 = std::min(xi[x] + 1, crop_cols - 1);
      const int s0 = ((int)r0[x0] * xa0[x] + (int)r0[x1] * xa1[x]) >> 4, s1 = ((int)r1[x0] * xa0[x] + (int)r1[x1] * xa1[x]) >> 4;
      const int v = (((yb0[y] * s0) >> 16) + ((yb1[y] * s1) >> 16) + 2) >> 2;
      resized[(size_t)y * W + x] = (uint8_t)std::min(std::max(v, 0), 255);
    }
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ network (nn.cpp:163-176)
static int forward_one(spvo_cpu *c, const float *input, float *det, float *desc) {
  const int H = c->cfg.net_height, W = c->cfg.net_width;
  Act &in = c->acts[c->t_input];
  for (int y = 0; y < H; ++y) std::memcpy(in.at(0, y, 0), input + (size_t)y * W, (size_t)W * sizeof(float));
  for (const Op &op : c->ops) {
    const Act &src = c->acts[op.in];
    Act &dst = c->acts[op.out];
    if (op.type == OP_CONV || op.type == OP_DWCONV) {
      const Act *res = (op.flags & FLAG_ADD) ? &c->acts[op.residual] : nullptr;
      Act *full = &dst;
      if (op.flags & FLAG_POOL) {
        if (c->scratch.C < (int)op.cout || c->scratch.H != src.H || c->scratch.W != src.W) c->scratch.shape((int)op.cout, src.H, src.W);
        full = &c->scratch;
      }
      Op tmp_off;   // a pooled conv writes channel 0.. of the scratch tensor
      const Op *run = &op;
      if (op.flags & FLAG_POOL) { tmp_off = op; tmp_off.out_c_off = 0; run = &tmp_off; }
      if (op.type == OP_CONV) conv_forward(*run, src, *full, res);
      else dwconv_forward(*run, src, *full, res);
      if (op.flags & FLAG_POOL) pool2(c->scratch, 0, (int)op.cout, dst, (int)op.out_c_off);
    } else if (op.type == OP_MAXPOOL) {
      pool2(src, 0, src.C, dst, 0);
    } else if (op.type == OP_L2NORM) {   // ReduceL2 over channels + Div, no epsilon
#pragma omp parallel for schedule(static)
      for (int y = 0; y < src.H; ++y)
        for (int x = 0; x < src.W; ++x) {
          float s = 0.f;
          for (int ch = 0; ch < src.C; ++ch) { const float v = *src.at(ch, y, x); s += v * v; }
          const float n = std::sqrt(s);
          for (int ch = 0; ch < src.C; ++ch) *dst.at(ch, y, x) = *src.at(ch, y, x) / n;
        }
    } else {
      return fail(-3, "unknown op type %u", op.type);
    }
  }
  auto dump = [&](const Act &a, float *out) {
    for (int ch = 0; ch < a.C; ++ch)
      for (int y = 0; y < a.H; ++y) std::memcpy(out + ((size_t)ch * a.H + y) * a.W, a.at(ch, y, 0), (size_t)a.W * sizeof(float));
  };
  if (det) dump(c->acts[c->t_det], det);
  if (desc) dump(c->acts[c->t_desc], desc);
  return 0;
}

int spvo_cpu_forward(spvo_cpu *c, const float *input, int batch, float *det, float *desc) {
  if (!c || !input || batch < 1) return fail(-1, "bad argument");
  if (!c->weights) return fail(-4, "no weights loaded");
  const int H = c->cfg.net_height, W = c->cfg.net_width, Hc = H / 8, Wc = W / 8;
  for (int b = 0; b < batch; ++b) {
    const int rc = forward_one(c, input + (size_t)b * H * W, det ? det + (size_t)b * 65 * Hc * Wc : nullptr, desc ? desc + (size_t)b * 256 * Hc * Wc : nullptr);
    if (rc) return rc;
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ post-processing (nn.cpp:188-431)
int spvo_cpu_heatmap(spvo_cpu *c, const float *det, float *heat) {
  if (!c || !det || !heat) return fail(-1, "bad argument");
  const int H = c->cfg.net_height, W = c->cfg.net_width, Hc = H / 8, Wc = W / 8;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < Hc; ++i)
    for (int j = 0; j < Wc; ++j) {
      float e[65], s = 0.f;
      for (int ch = 0; ch < 65; ++ch) { e[ch] = std::exp(det[((size_t)ch * Hc + i) * Wc + j]); s += e[ch]; }   // no max-subtraction (nn.cpp:271)
      s += 0.00001f;                                                                                          // nn.cpp:274-283
      for (int ch = 0; ch < 64; ++ch) heat[(size_t)(8 * i + ch / 8) * W + 8 * j + ch % 8] = e[ch] / s;          // nn.cpp:289-326
    }
  return 0;
}

int spvo_cpu_nms(spvo_cpu *c, const float *heat, int32_t *xy, int *n) {
  if (!c || !heat || !xy || !n) return fail(-1, "bad argument");
  const int H = c->cfg.net_height, W = c->cfg.net_width, dist = c->cfg.dist_thresh, border = c->cfg.border_remove;
  struct Cand { float conf; int64_t cm; int x, y; };
  std::vector<Cand> cand;
  for (int x = 0; x < W; ++x)            // column-major visiting order of the reference's sparse matrix (nn.cpp:205-213)
    for (int y = 0; y < H; ++y) {
      const float v = heat[(size_t)y * W + x];
      if (v > c->cfg.conf_thresh) cand.push_back(Cand{v, (int64_t)x * H + y, x, y});   // strict > (nn.cpp:203)
    }
  // std::sort by confidence is unstable in the reference; the pinned total order is (confidence desc, column-major index asc)
  std::sort(cand.begin(), cand.end(), [](const Cand &a, const Cand &b) { return a.conf != b.conf ? a.conf > b.conf : a.cm < b.cm; });
  std::vector<uint8_t> sup((size_t)H * W, 0);
  int count = 0;
  for (const Cand &k : cand) {
    if (sup[(size_t)k.y * W + k.x]) continue;
    if (border <= k.y && k.y + border < H && border <= k.x && k.x + border < W) { xy[2 * count] = k.x; xy[2 * count + 1] = k.y; ++count; }
    for (int yy = std::max(0, k.y - dist); yy <= std::min(H - 1, k.y + dist); ++yy)
      std::memset(&sup[(size_t)yy * W + std::max(0, k.x - dist)], 1, (size_t)(std::min(W - 1, k.x + dist) - std::max(0, k.x - dist) + 1));
    if (count >= c->cfg.max_keypoints) break;   // nn.cpp:256-257
  }
  *n = count;
  return 0;
}

int spvo_cpu_sample_descriptors(spvo_cpu *c, const float *desc, const int32_t *xy, int n, float *out) {
  if (!c || !desc || (n > 0 && (!xy || !out))) return fail(-1, "bad argument");
  const int H = c->cfg.net_height, W = c->cfg.net_width, Hc = H / 8, Wc = W / 8;
  const size_t plane = (size_t)Hc * Wc;
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    const int col = xy[2 * i], row = xy[2 * i + 1];
    const float row8 = (float)row / (float)(H - 1) * (float)(Hc - 1), col8 = (float)col / (float)(W - 1) * (float)(Wc - 1);   // nn.cpp:377-382
    const int r0 = (int)std::floor(row8), c0 = (int)std::floor(col8);
    const float rr = 1.0f - (row8 - (float)r0), cr = 1.0f - (col8 - (float)c0);
    const int r1 = std::min(r0 + 1, Hc - 1), c1 = std::min(c0 + 1, Wc - 1);
    float v[256], s = 0.f;
    for (int ch = 0; ch < 256; ++ch) {
      const float *p = desc + ch * plane;
      const float tl = p[(size_t)r0 * Wc + c0], tr = p[(size_t)r0 * Wc + c1], bl = p[(size_t)r1 * Wc + c0], br = p[(size_t)r1 * Wc + c1];
      v[ch] = tl * rr * cr + tr * rr * (1.0f - cr) + bl * (1.0f - rr) * cr + br * (1.0f - rr) * (1.0f - cr);   // nn.cpp:423-427
      s += v[ch] * v[ch];
    }
    const float nrm = std::sqrt(s);
    for (int ch = 0; ch < 256; ++ch) out[(size_t)i * 256 + ch] = v[ch] / nrm;                                  // nn.cpp:428
  }
  return 0;
}

int spvo_cpu_detect(spvo_cpu *c, const uint8_t *img, int rows, int cols, size_t stride, double P[12], float *xy, float *desc, int *n) {
  if (!c || !xy || !desc || !n) return fail(-1, "bad argument");
  if (!c->weights) return fail(-4, "no weights loaded");
  const int H = c->cfg.net_height, W = c->cfg.net_width, Hc = H / 8, Wc = W / 8;
  std::vector<uint8_t> resized((size_t)H * W);
  int rc = spvo_cpu_preprocess(c, img, rows, cols, stride, P, resized.data());
  if (rc) return rc;
  std::vector<float> x((size_t)H * W), det((size_t)65 * Hc * Wc), dsc((size_t)256 * Hc * Wc), heat((size_t)H * W);
  for (size_t i = 0; i < x.size(); ++i) x[i] = (float)resized[i] * (1.0f / 255.0f);   // nn.cpp:159
  if ((rc = forward_one(c, x.data(), det.data(), dsc.data()))) return rc;
  if ((rc = spvo_cpu_heatmap(c, det.data(), heat.data()))) return rc;
  std::vector<int32_t> kp((size_t)c->cfg.max_keypoints * 2);
  if ((rc = spvo_cpu_nms(c, heat.data(), kp.data(), n))) return rc;
  for (int i = 0; i < 2 * *n; ++i) xy[i] = (float)kp[i];
  return spvo_cpu_sample_descriptors(c, dsc.data(), kp.data(), *n, desc);
}

// ------------------------------------------------------------------------------------------------ matching (base.cpp:434-491)
// canonical distance: sequential sum over k with separately rounded multiply and add (the scalar loop of normL2Sqr_).
// fp-contract is switched off for these two functions only (a fused multiply-add would round differently); four train rows
// are summed side by side -- four independent chains, each in the canonical order.
__attribute__((optimize("fp-contract=off"), noinline)) static void sqdist4(const float *a, const float *b, int rows, float out[4]) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  const float *b0 = b, *b1 = b + 256 * (rows > 1), *b2 = b + 512 * (rows > 2), *b3 = b + 768 * (rows > 3);
  for (int k = 0; k < 256; ++k) {
    const float av = a[k];
    const float t0 = av - b0[k], t1 = av - b1[k], t2 = av - b2[k], t3 = av - b3[k];
    const float p0 = t0 * t0, p1 = t1 * t1, p2 = t2 * t2, p3 = t3 * t3;
    s0 = s0 + p0; s1 = s1 + p1; s2 = s2 + p2; s3 = s3 + p3;
  }
  out[0] = s0; out[1] = s1; out[2] = s2; out[3] = s3;
}
static void best_two_rows(const float *a, int na, const float *b, int nb, std::vector<float> &d0, std::vector<float> &d1, std::vector<int> &i0, std::vector<int> &i1) {
  const float inf = std::numeric_limits<float>::infinity();
  d0.assign(na, inf); d1.assign(na, inf); i0.assign(na, -1); i1.assign(na, -1);
#pragma omp parallel for schedule(static)
  for (int q = 0; q < na; ++q) {
    float b0 = inf, b1 = inf;
    int j0 = -1, j1 = -1;
    for (int tb = 0; tb < nb; tb += 4) {
      float d4[4];
      const int rows = std::min(4, nb - tb);
      sqdist4(a + (size_t)q * 256, b + (size_t)tb * 256, rows, d4);
      for (int r = 0; r < rows; ++r) {   // strict '<': the lowest train index wins ties
        const float d = d4[r];
        const int t = tb + r;
        if (j0 < 0 || d < b0) { b1 = b0; j1 = j0; b0 = d; j0 = t; }
        else if (j1 < 0 || d < b1) { b1 = d; j1 = t; }
      }
    }
    d0[q] = b0; d1[q] = b1; i0[q] = j0; i1[q] = j1;
  }
}

int spvo_cpu_match(spvo_cpu *c, const float *a, int na, const float *b, int nb, int selector, int cross_check, float ratio, int32_t *train_idx, float *distance) {
  (void)c;
  if (na < 0 || nb < 0 || (na > 0 && (!a || !train_idx || !distance))) return fail(-1, "bad argument");
  for (int i = 0; i < na; ++i) { train_idx[i] = -1; distance[i] = 0.f; }
  if (na == 0 || nb == 0) return 0;
  std::vector<float> d0, d1;
  std::vector<int> i0, i1;
  if (selector == 0 && cross_check) {
    // cv::batchDistance crosscheck: every TRAIN row votes for its nearest query row (lowest index on ties); a query keeps the
    // nearest of its voters (the first such train row on ties)
    best_two_rows(b, nb, a, na, d0, d1, i0, i1);
    std::vector<float> best(na, std::numeric_limits<float>::infinity());
    for (int t = 0; t < nb; ++t) {
      const int q = i0[t];
      const float dt = std::sqrt(d0[t]);
      if (dt < best[q]) { best[q] = dt; train_idx[q] = t; distance[q] = dt; }
    }
    return 0;
  }
  best_two_rows(a, na, b, nb, d0, d1, i0, i1);
  for (int q = 0; q < na; ++q) {
    const float s0 = std::sqrt(d0[q]), s1 = std::sqrt(d1[q]);
    distance[q] = s0;
    if (selector == 0) train_idx[q] = i0[q];
    else if (nb >= 2 && s0 < ratio * s1) train_idx[q] = i0[q];   // base.cpp:469
  }
  return 0;
}

// ------------------------------------------------------------------------------------------------ odometry stages
int spvo_cpu_triangulate(const double P_l[12], const double P_r[12], const float *xy_l, const float *xy_r, int n, float *xyz) {
  if (!P_l || !P_r || n < 0 || (n > 0 && (!xy_l || !xy_r || !xyz))) return fail(-1, "bad argument");
#pragma omp parallel for schedule(static)
  for (int i = 0; i < n; ++i) {
    double A[4][4];
    const double *P[2] = {P_l, P_r};
    const float *pt[2] = {xy_l + 2 * i, xy_r + 2 * i};
    for (int j = 0; j < 2; ++j) {
      const double x = (double)pt[j][0], y = (double)pt[j][1];
      for (int k = 0; k < 4; ++k) { A[2 * j][k] = x * P[j][8 + k] - P[j][k]; A[2 * j + 1][k] = y * P[j][8 + k] - P[j][4 + k]; }
    }
    double h[4];
    null_vector4(A, h);
    const float hf[4] = {(float)h[0], (float)h[1], (float)h[2], (float)h[3]};   // points4D is CV_32F for Point2f input
    const float scale = hf[3] != 0.f ? 1.0f / hf[3] : 1.0f;                      // convertPointsFromHomogeneous
    for (int k = 0; k < 3; ++k) xyz[3 * i + k] = hf[k] * scale;
  }
  return 0;
}

int spvo_cpu_pnp_ransac(const double K[9], const float *xyz, const float *xy, int n, int iterations, double reproj_error, uint32_t seed, double rvec[3],
                        double tvec[3], int32_t *inliers, int *n_inliers, int *ok) {
  if (!K || !rvec || !tvec || !n_inliers || !ok || n < 0) return fail(-1, "bad argument");
  std::vector<int32_t> inl;
  bool good = false;
  pnp_ransac(K, xyz, xy, n, iterations, reproj_error, seed, rvec, tvec, inl, good);
  *ok = good ? 1 : 0;
  *n_inliers = (int)inl.size();
  if (inliers) std::copy(inl.begin(), inl.end(), inliers);
  return 0;
}

int spvo_cpu_pnp_refine(const double P_l[12], const double P_r[12], const spvo_cpu_obs *obs, int n_obs, int max_iterations, double huber_delta, double q[4],
                        double t[3], spvo_cpu_refine_summary *summary) {
  if (!P_l || !P_r || !q || !t || !summary || n_obs < 0) return fail(-1, "bad argument");
  std::vector<ObsD> o(n_obs);
  for (int i = 0; i < n_obs; ++i) {
    for (int k = 0; k < 3; ++k) o[i].X[k] = (double)obs[i].X[k];
    for (int k = 0; k < 2; ++k) o[i].uv[k] = (double)obs[i].uv[k];
    o[i].cam = obs[i].cam; o[i].inv = obs[i].inverse;
  }
  Quat qq{q[0], q[1], q[2], q[3]};
  pnp_refine(P_l, P_r, o, max_iterations, huber_delta, qq, t, *summary);
  q[0] = qq.x; q[1] = qq.y; q[2] = qq.z; q[3] = qq.w;
  return 0;
}

// ------------------------------------------------------------------------------------------------ stereoCallback (node.cpp:150-262)
int spvo_cpu_frontend_reset(spvo_cpu *c, int selector, int cross_check, float stereo_threshold, float min_disparity, int refinement_degree) {
  if (!c) return fail(-1, "null context");
  c->selector = selector; c->cross_check = cross_check && selector == 0;   // base.cpp:27-28: crossCheck only without KNN
  c->stereo_threshold = stereo_threshold; c->min_disparity = min_disparity; c->refinement_degree = refinement_degree;
  c->kp_dq.clear(); c->desc_dq.clear(); c->bdesc_dq.clear();
  c->classic = false;
  for (auto &m : c->maps) m.clear();
  for (int k = 0; k < 3; ++k) c->r_pred[k] = c->t_pred[k] = 0;
  c->frame_count = 0;
  c->prev_pts_inited = false;
  c->prev_pts3d.clear(); c->prev_matched_to_valid.clear();
  return 0;
}

int spvo_cpu_frontend_reset_classic(spvo_cpu *c, int selector, int cross_check, float stereo_threshold, int refinement_degree) {
  // ClassicFeatureFrontEnd's constructor hands stereo_threshold to the base class as min_disparity too (hpp:203-206)
  const int rc = spvo_cpu_frontend_reset(c, selector, cross_check, stereo_threshold, stereo_threshold, refinement_degree);
  if (!rc) c->classic = true;
  return rc;
}

int spvo_cpu_orb(spvo_cpu *c, const uint8_t *img, int rows, int cols, size_t stride, float *xy, float *angle_response_octave, uint8_t *desc, int cap, int *n) {
  (void)c;
  if (!img || !n || rows <= 0 || cols <= 0) return fail(-1, "bad argument");
  std::vector<orb::Kp> kps;
  std::vector<uint8_t> bd;
  orb::detect_and_compute(img, rows, cols, stride, kps, bd);
  *n = (int)kps.size();
  for (int i = 0; i < *n && i < cap; ++i) {
    if (xy) { xy[2 * i] = kps[i].x; xy[2 * i + 1] = kps[i].y; }
    if (angle_response_octave) { angle_response_octave[3 * i] = kps[i].angle; angle_response_octave[3 * i + 1] = kps[i].response; angle_response_octave[3 * i + 2] = (float)kps[i].octave; }
    if (desc) std::memcpy(desc + (size_t)i * 32, &bd[(size_t)i * 32], 32);
  }
  return 0;
}

// the 256 test pairs (x1, y1, x2, y2) and the 7 smoothing taps: a second implementation has to use the same tables
int spvo_cpu_orb_tables(float *pattern /* [1024] */, float *taps /* [7] */) {
  orb::make_pattern();
  if (pattern) std::memcpy(pattern, orb::g_pattern.data(), 1024 * sizeof(float));
  if (taps) orb::gauss_taps(taps);
  return 0;
}

static void frontend_match(spvo_cpu *c, int type) {   // base.cpp:434-491
  static const int pos[3][2] = {{-2, -1}, {-2, -4}, {-4, -3}};   // hpp:87-90
  const int nd = (int)c->kp_dq.size();
  if (c->classic) {
    const std::vector<uint8_t> &ba = c->bdesc_dq[nd + pos[type][0]], &bb = c->bdesc_dq[nd + pos[type][1]];
    std::vector<int32_t> idx;
    orb::hamming_match(ba.data(), (int)ba.size() / 32, bb.data(), (int)bb.size() / 32, c->selector, c->cross_check, 0.8f, idx);
    if (type == 0) c->maps[2] = c->maps[0];
    c->maps[type] = idx;
    return;
  }
  const std::vector<float> &da = c->desc_dq[nd + pos[type][0]], &db = c->desc_dq[nd + pos[type][1]];
  const int na = (int)da.size() / 256, nb = (int)db.size() / 256;
  std::vector<int32_t> idx(std::max(na, 1));
  std::vector<float> dist(std::max(na, 1));
  spvo_cpu_match(c, da.data(), na, db.data(), nb, c->selector, c->cross_check, 0.8f, idx.data(), dist.data());
  idx.resize(na);
  if (type == 0) c->maps[2] = c->maps[0];   // base.cpp:475-481
  c->maps[type] = idx;
}

int spvo_cpu_frontend_map(spvo_cpu *c, int match_type, int32_t *out, int cap) {
  if (!c || match_type < 0 || match_type > 2) return -1;
  const auto &m = c->maps[match_type];
  for (int i = 0; i < (int)m.size() && i < cap; ++i) out[i] = m[i];
  return (int)m.size();
}

// keypoints of deque position -4..-1 (ImagePosition, hpp:66-72) after the last step: xy [cap][2]; returns their number (-1: no such entry)
int spvo_cpu_frontend_keypoints(spvo_cpu *c, int position, float *xy, int cap) {
  if (!c || position >= 0 || (int)c->kp_dq.size() + position < 0) return -1;
  const std::vector<float> &k = c->kp_dq[c->kp_dq.size() + position];
  const int n = (int)k.size() / 2;
  for (int i = 0; i < n && i < cap; ++i) { xy[2 * i] = k[2 * i]; xy[2 * i + 1] = k[2 * i + 1]; }
  return n;
}

int spvo_cpu_frontend_step(spvo_cpu *c, const uint8_t *img_l, const uint8_t *img_r, int rows, int cols, size_t stride, const double P_l[12],
                           const double P_r[12], spvo_cpu_step_result *res) {
  if (!c || !img_l || !img_r || !P_l || !P_r || !res) return fail(-1, "bad argument");
  std::memset(res, 0, sizeof *res);
  res->q[3] = 1;
  const double t0 = now_ms();
  // addStereoImagePair (nn.cpp:449-498)
  const int cap = c->cfg.max_keypoints;
  for (int side = 0; side < 2; ++side) {
    double *P = side ? c->P_r : c->P_l;
    std::memcpy(P, side ? P_r : P_l, 12 * sizeof(double));
    if (c->classic) {   // classic.cpp:81-120 with image_height = image_width = 0: native resolution, P unchanged
      std::vector<orb::Kp> kps;
      std::vector<uint8_t> bd;
      orb::detect_and_compute(side ? img_r : img_l, rows, cols, stride, kps, bd);
      std::vector<float> xy(kps.size() * 2);
      for (size_t i = 0; i < kps.size(); ++i) { xy[2 * i] = kps[i].x; xy[2 * i + 1] = kps[i].y; }
      (side ? res->n_kp_r : res->n_kp_l) = (int)kps.size();
      c->kp_dq.push_back(std::move(xy)); c->desc_dq.emplace_back(); c->bdesc_dq.push_back(std::move(bd));
      continue;
    }
    std::vector<float> xy((size_t)cap * 2), desc((size_t)cap * 256);
    int n = 0;
    const int rc = spvo_cpu_detect(c, side ? img_r : img_l, rows, cols, stride, P, xy.data(), desc.data(), &n);
    if (rc) return rc;
    xy.resize((size_t)n * 2); desc.resize((size_t)n * 256);
    c->kp_dq.push_back(std::move(xy)); c->desc_dq.push_back(std::move(desc)); c->bdesc_dq.emplace_back();
    (side ? res->n_kp_r : res->n_kp_l) = n;
  }
  while (c->kp_dq.size() > 4) { c->kp_dq.erase(c->kp_dq.begin()); c->desc_dq.erase(c->desc_dq.begin()); c->bdesc_dq.erase(c->bdesc_dq.begin()); }
  const double t1 = now_ms();
  res->t_detect_ms = (float)(t1 - t0);
  frontend_match(c, 0);
  for (int v : c->maps[0]) res->n_stereo += v >= 0;
  if (c->kp_dq.size() < 4) {   // first frame (node.cpp:188-193)
    res->t_match_ms = (float)(now_ms() - t1);
    res->t_total_ms = (float)(now_ms() - t0);
    return 0;
  }
  frontend_match(c, 1);
  for (int v : c->maps[1]) res->n_temporal += v >= 0;
  const double t2 = now_ms();
  res->t_match_ms = (float)(t2 - t1);

  // solveStereoOdometry: the 4-way join (base.cpp:127-207)
  const std::vector<float> &pl = c->kp_dq[0], &pr = c->kp_dq[1], &cl = c->kp_dq[2], &cr = c->kp_dq[3];
  const int n_cl = (int)cl.size() / 2;
  std::vector<float> j_cl, j_cr, j_pl, j_pr;
  std::vector<int32_t> valid_to_prev, cur_matched_to_valid(n_cl, -1);
  for (int qi = 0; qi < n_cl; ++qi) {
    const int ti = c->maps[0][qi];
    if (ti < 0 || c->maps[1][qi] == -1) continue;
    const float ax = cl[2 * qi], ay = cl[2 * qi + 1], bx = cr[2 * ti], by = cr[2 * ti + 1];
    if (std::fabs(ay - by) > c->stereo_threshold || std::fabs(ax - bx) < c->min_disparity) continue;   // base.cpp:169-172
    const int pi = c->maps[1][qi];
    if (pi >= (int)c->maps[2].size() || c->maps[2][pi] == -1) continue;
    const int pj = c->maps[2][pi];
    j_cl.insert(j_cl.end(), {ax, ay}); j_cr.insert(j_cr.end(), {bx, by});
    j_pl.insert(j_pl.end(), {pl[2 * pi], pl[2 * pi + 1]}); j_pr.insert(j_pr.end(), {pr[2 * pj], pr[2 * pj + 1]});
    if (c->refinement_degree >= 3) { cur_matched_to_valid[qi] = (int)j_cl.size() / 2 - 1; valid_to_prev.push_back(pi); }
  }
  const int n = (int)j_cl.size() / 2;
  res->n_joined = n;
  std::vector<float> pts3d((size_t)std::max(n, 1) * 3);
  spvo_cpu_triangulate(c->P_l, c->P_r, j_cl.data(), j_cr.data(), n, pts3d.data());
  const double K[9] = {c->P_l[0], c->P_l[1], c->P_l[2], c->P_l[4], c->P_l[5], c->P_l[6], c->P_l[8], c->P_l[9], c->P_l[10]};
  double rvec[3] = {c->r_pred[0], c->r_pred[1], c->r_pred[2]}, tvec[3] = {c->t_pred[0], c->t_pred[1], c->t_pred[2]};
  std::vector<int32_t> inl;
  bool ok = false;
  pnp_ransac(K, pts3d.data(), j_pl.data(), n, 500, 2.0, 0, rvec, tvec, inl, ok);   // base.cpp:237-239
  res->pnp_ok = ok; res->n_inliers = (int)inl.size();
  const double acc = std::sqrt((tvec[0] - c->t_pred[0]) * (tvec[0] - c->t_pred[0]) + (tvec[1] - c->t_pred[1]) * (tvec[1] - c->t_pred[1]) +
                               (tvec[2] - c->t_pred[2]) * (tvec[2] - c->t_pred[2])) / 0.1;   // base.cpp:241-242
  bool do_opt = false;
  if (!ok || (c->frame_count > 10 && acc > 8.0)) { for (int k = 0; k < 3; ++k) { rvec[k] = c->r_pred[k]; tvec[k] = c->t_pred[k]; } }
  else { for (int k = 0; k < 3; ++k) { c->r_pred[k] = rvec[k]; c->t_pred[k] = tvec[k]; } do_opt = true; }
  res->accepted = do_opt;
  Quat q = rvec_to_quat(rvec);
  double t[3] = {tvec[0], tvec[1], tvec[2]};
  if (do_opt && c->refinement_degree > 0) {   // residual blocks in the order base.cpp:291-356 adds them
    std::vector<ObsD> obs;
    auto push = [&](const float *X, const float *uv, int cam, int inv) {
      ObsD o;
      for (int k = 0; k < 3; ++k) o.X[k] = (double)X[k];
      o.uv[0] = (double)uv[0]; o.uv[1] = (double)uv[1]; o.cam = cam; o.inv = inv;
      obs.push_back(o);
    };
    for (int vi : inl) {
      push(&pts3d[3 * vi], &j_pl[2 * vi], 0, 0);
      if (c->refinement_degree <= 1) continue;
      push(&pts3d[3 * vi], &j_pr[2 * vi], 1, 0);
      if (c->refinement_degree <= 2 || !c->prev_pts_inited) continue;
      const int pm = valid_to_prev[vi];
      if (pm >= (int)c->prev_matched_to_valid.size()) continue;
      const int pv = c->prev_matched_to_valid[pm];
      if (pv == -1) continue;
      push(&c->prev_pts3d[3 * pv], &j_cl[2 * vi], 0, 1);
      if (c->refinement_degree <= 3) continue;
      push(&c->prev_pts3d[3 * pv], &j_cr[2 * vi], 1, 1);
    }
    Quat q2 = q;
    double t2v[3] = {t[0], t[1], t[2]};
    spvo_cpu_refine_summary S;
    pnp_refine(c->P_l, c->P_r, obs, 40, 1.0, q2, t2v, S);
    res->lm_iterations = S.iterations;
    if (S.usable && S.converged) { q = q2; for (int k = 0; k < 3; ++k) t[k] = t2v[k]; res->refined = 1; }   // base.cpp:366-374
  }
  // cam0_curr_T_cam0_prev = (q, t)^-1   base.cpp:377-385
  const double nq = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  const Quat qn{q.x / nq, q.y / nq, q.z / nq, q.w / nq};
  double R[9];
  quat_to_rot(qn, R);
  res->q[0] = -qn.x; res->q[1] = -qn.y; res->q[2] = -qn.z; res->q[3] = qn.w;
  for (int i = 0; i < 3; ++i) res->t[i] = -(R[i] * t[0] + R[3 + i] * t[1] + R[6 + i] * t[2]);
  if (c->refinement_degree >= 3) {   // base.cpp:388-394
    c->prev_matched_to_valid = cur_matched_to_valid;
    c->prev_pts3d.assign(pts3d.begin(), pts3d.begin() + (size_t)n * 3);
    c->prev_pts_inited = true;
  }
  ++c->frame_count;
  const double t3 = now_ms();
  res->t_solve_ms = (float)(t3 - t2);
  res->t_total_ms = (float)(t3 - t0);
  return 0;
}

}  // extern "C"
