// conv_first_pre.hip.h -- K0 + K1 in one launch: crop / resize / normalise (post.hip.h's preprocess_kernel: preprocessImageImpl,
// feature_detection_base.cpp:72-122, and the SuperPoint front end's / 255, feature_detection_neural_network.cpp:144-152) straight into the
// first 3x3 layer (conv_first4_kernel, conv_mfma.hip.h: the Conv/Relu node the TensorRT engine runs first, nn.cpp:169).
//
// Why: a detector submission's preprocess launch sits on the network stream between two trunks (8-10 us per stereo pair of a 660 us step,
// a launch boundary, the fp32 input planes written and read straight back).  Here the workgroup that convolves 4 rows x 256 columns of an
// image evaluates its own 6 x 258 input pixels from the caller's u8 image (the same integer arithmetic, table-driven: bit-identical), keeps
// them in LDS, and writes what the preprocess kernel wrote -- the resized u8 image (images_dq, nn.cpp:154) and the fp32 input plane (kept: the
// stand-alone entry points and spvo_debug_tensor read it) -- for its interior pixels on the way.  1.5 evaluations per pixel instead of 1;
// the layer stays bound by its 64 output planes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "conv_mfma.hip.h"
#include "post.hip.h"

namespace spvo {

struct FirstPreArgs {
  const uint8_t *src[4];      // the launch's images (a group of one or two stereo pairs), caller's u8, `stride` bytes per row
  uint8_t *out_u8[4];         // resized u8 image of each (H x W), or NULL
  size_t stride;
  int row_off, col_off, crop_rows, crop_cols, identity;   // the crop of preprocessImageImpl: the same for every image of the launch
  ResizeTab tab;
  float *in_plane;            // fp32 input planes [img][hp][wp]
  size_t in_per_image;
  float *out;                 // [img][out_ctot][hp][wp]
  const float *w, *bias;      // [cout][9], [cout]
  int H, W, hp, wp, out_ctot, out_coff, cout;
};

template <bool RELU>
__global__ __launch_bounds__(256) void conv_first4_pre_kernel(const FirstPreArgs a) {
  constexpr int TW = 256, LW = TW + 4;   // (row pitch 260 floats: a lane's 16-byte piece stays aligned)
  __shared__ __attribute__((aligned(16))) float s[6 * LW];   // rows y0 - 1 .. y0 + 4, columns x0 - 1 .. x0 + 256 at index 3 + (column - x0 + 1) ... see below
  const int tid = threadIdx.x, img = blockIdx.z;
  const int x0 = blockIdx.x * TW, y0 = blockIdx.y * 4;
  const uint8_t *src = a.src[img];
  uint8_t *o8 = a.out_u8[img];
  float *ipl = a.in_plane + (size_t)img * a.in_per_image;
  // column c of the staged tile (0 .. 257) = image column x0 - 1 + c, kept at s[row * LW + 3 + c]: image column x0 + 4 l sits at a multiple of 4.
  // Thread t stages column t of all six rows (its three column coefficients are fetched once, the row coefficients are wave-uniform, the
  // 24 source bytes are independent loads: two memory latencies for the whole tile), threads 0 .. 11 columns 256 and 257 as well.
  auto stage = [&](const int cx, const int ry0, const int nrows) {
    const int gx = x0 - 1 + cx;
    const bool col_ok = gx >= 0 && gx < a.W;
    const int gxc = col_ok ? gx : 0;
    int xs0 = 0, xs1 = 0, a0 = 0, a1 = 0;
    if (!a.identity) { xs0 = a.tab.xi[gxc]; xs1 = min(xs0 + 1, a.crop_cols - 1); a0 = a.tab.xa0[gxc]; a1 = a.tab.xa1[gxc]; }
    int vv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k >= nrows) break;
      const int gy = y0 - 1 + ry0 + k;
      const int gyc = min(max(gy, 0), a.H - 1);
      if (a.identity) {  // cv::resize copies when the sizes already match
        vv[k] = src[(size_t)(a.row_off + gyc) * a.stride + a.col_off + gxc];
      } else {           // preprocess_kernel's arithmetic, line by line
        const int ys0 = a.tab.yi[gyc], ys1 = min(ys0 + 1, a.crop_rows - 1);
        const int b0 = a.tab.yb0[gyc], b1 = a.tab.yb1[gyc];
        const uint8_t *r0 = src + (size_t)(a.row_off + ys0) * a.stride + a.col_off;
        const uint8_t *r1 = src + (size_t)(a.row_off + ys1) * a.stride + a.col_off;
        const int h0 = (int)r0[xs0] * a0 + (int)r0[xs1] * a1;
        const int h1 = (int)r1[xs0] * a0 + (int)r1[xs1] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        vv[k] = min(max(v, 0), 255);
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
      if (k >= nrows) break;
      const int ry = ry0 + k, gy = y0 - 1 + ry;
      const bool ok = col_ok && gy >= 0 && gy < a.H;
      const float f = ok ? mul_rn((float)vv[k], 1.0f / 255.0f) : 0.f;   // outside the image: the planes' zero padding
      if (ok && ry >= 1 && ry <= 4 && cx >= 1 && cx <= TW) {   // the workgroup's own pixels: what the preprocess kernel wrote
        if (o8) o8[(size_t)gy * a.W + gx] = (uint8_t)vv[k];
        ipl[(size_t)(gy + PADY) * a.wp + (gx + PADX)] = f;
      }
      s[ry * LW + 3 + cx] = f;
    }
  };
  stage(tid, 0, 6);
  if (tid < 12) stage(TW + (tid & 1), tid >> 1, 1);
  __syncthreads();
  const int l = tid & 63, r = tid >> 6;
  const int x = x0 + 4 * l, y = y0 + r;
  if (x >= a.W || y >= a.H) return;   // W is a multiple of 8: a thread's 4 pixels are all inside or all outside
  float v[3][6];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const float *rp = s + (r + ky) * LW + 4 + 4 * l;   // image column x
    const float4 m = *reinterpret_cast<const float4 *>(rp);
    v[ky][0] = rp[-1]; v[ky][1] = m.x; v[ky][2] = m.y; v[ky][3] = m.z; v[ky][4] = m.w; v[ky][5] = rp[4];
  }
  const size_t plane = (size_t)a.hp * a.wp;
  float *op = a.out + ((size_t)img * a.out_ctot + a.out_coff) * plane + (size_t)(y + PADY) * a.wp + (x + PADX);
  for (int co = 0; co < a.cout; ++co) {
    float sum[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      sum[i] = a.bias[co];
#pragma unroll
      for (int t = 0; t < 9; ++t) sum[i] = fmaf(a.w[co * 9 + t], v[t / 3][i + t % 3], sum[i]);   // conv_first4_kernel's tap order
      if (RELU) sum[i] = fmaxf(sum[i], 0.f);
    }
    *reinterpret_cast<float4 *>(op + (size_t)co * plane) = make_float4(sum[0], sum[1], sum[2], sum[3]);
  }
}

}  // namespace spvo
