// spvo_net_s3.hip -- FP32 engines in split mode (conv_bf16x3.hip.h): kernel launchers.
#include "spvo_internal.hip.h"
#include "conv_bf16x3.hip.h"

namespace spvo_int {

// ---------------------------------------------------------------- FP32 engines in split mode (bf16x3)
template <int KS, int CKG, int WR, int WC, bool POOL, bool RELU, bool OUT_F32>
int launch_conv_s3_instance(spvo_ctx *c, ConvArgsS3 args, hipStream_t stream) {
  using T = ConvTileS3<KS, CKG, WR, WC>;
  auto k = conv_s3_kernel<KS, CKG, WR, WC, POOL, RELU, OUT_F32>;
  static int per_cu[64] = {};
  const int dev = c->cfg.device & 63;
  if (!per_cu[dev]) {
    HIP_TRY(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    int n = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k, 256, T::LDS_BYTES));
    per_cu[dev] = std::max(n, 1);
  }
  args.tiles_x = (args.W + T::TW - 1) / T::TW;
  args.tiles_y = (args.H + T::TH - 1) / T::TH;
  const int n_tiles = args.tiles_x * args.tiles_y * args.co_tiles * args.batch;
  hipLaunchKernelGGL(k, dim3(std::min(n_tiles, c->num_cus * per_cu[dev])), dim3(256), T::LDS_BYTES, stream, args);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

template <int KS, int CKG, int WR, int WC, bool POOL>
int launch_conv_s3_variant(spvo_ctx *c, const ConvArgsS3 &a, bool relu, bool out_f32, hipStream_t stream) {
  if constexpr (!POOL) {
    if (out_f32) return relu ? launch_conv_s3_instance<KS, CKG, WR, WC, false, true, true>(c, a, stream) : launch_conv_s3_instance<KS, CKG, WR, WC, false, false, true>(c, a, stream);
  }
  return relu ? launch_conv_s3_instance<KS, CKG, WR, WC, POOL, true, false>(c, a, stream) : launch_conv_s3_instance<KS, CKG, WR, WC, POOL, false, false>(c, a, stream);
}

int launch_conv_s3(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream) {
  const Tensor &ti = c->tensors[op.in];
  const Tensor &to = c->tensors[op.out];
  const float *tin = (ti.dr[c->cur_ring] ? ti.dr[c->cur_ring] : ti.d) + (size_t)img0 * ti.per_image;
  float *tout = (to.dr[c->cur_ring] ? to.dr[c->cur_ring] : to.d) + (size_t)img0 * to.per_image;
  const bool relu = op.flags & FLAG_RELU, pool = op.flags & FLAG_POOL;
  if (op.cin == 1) {
    dim3 grid((ti.W + 63) / 64, (ti.H + 3) / 4, batch);
#define SPVO_FIRST_S3(KS, RELU) hipLaunchKernelGGL((conv_first_s3_kernel<KS, RELU>), grid, dim3(256), 0, stream, tin, (unsigned short *)tout, op.d_w, op.d_b, \
                                                   ti.H, ti.W, ti.hp, ti.wp, to.ch / 8, op.out_c_off / 8, op.cout)
    if (op.ks == 3) { if (relu) SPVO_FIRST_S3(3, true); else SPVO_FIRST_S3(3, false); }
    else            { if (relu) SPVO_FIRST_S3(1, true); else SPVO_FIRST_S3(1, false); }
#undef SPVO_FIRST_S3
    HIP_TRY(c, hipGetLastError());
    return SPVO_OK;
  }
  ConvArgsS3 a;
  a.in = (const unsigned short *)tin; a.out = tout; a.wpack = op.d_ws3;
  a.H = ti.H; a.W = ti.W;
  a.in_hp = ti.hp; a.in_wp = ti.wp; a.in_gtot = ti.ch / 8; a.in_goff = op.in_c_off / 8;
  a.out_hp = to.hp; a.out_wp = to.wp; a.out_ctot = to.ch; a.out_coff = op.out_c_off;
  a.cout = op.cout; a.n_chunks = op.n_chunks; a.co_tiles = op.co_tiles;
  a.tiles_x = a.tiles_y = 0;
  a.batch = batch;
  const bool out_f32 = !to.s3;
  const int key = op.ks * 1000 + op.wr * 100 + op.wc * 10 + (pool ? 1 : 0);
  switch (key) {
    case 3220: return launch_conv_s3_variant<3, 1, 2, 2, false>(c, a, relu, out_f32, stream);
    case 3210: return launch_conv_s3_variant<3, 1, 2, 1, false>(c, a, relu, out_f32, stream);
    case 3120: return launch_conv_s3_variant<3, 1, 1, 2, false>(c, a, relu, out_f32, stream);
    case 3110: return launch_conv_s3_variant<3, 1, 1, 1, false>(c, a, relu, out_f32, stream);
    case 3221: return launch_conv_s3_variant<3, 1, 2, 2, true>(c, a, relu, out_f32, stream);
    case 3211: return launch_conv_s3_variant<3, 1, 2, 1, true>(c, a, relu, out_f32, stream);
    case 1220: return launch_conv_s3_variant<1, 2, 2, 2, false>(c, a, relu, out_f32, stream);
    case 1120: return launch_conv_s3_variant<1, 2, 1, 2, false>(c, a, relu, out_f32, stream);
    case 1110: return launch_conv_s3_variant<1, 2, 1, 1, false>(c, a, relu, out_f32, stream);
    case 1221: return launch_conv_s3_variant<1, 2, 2, 2, true>(c, a, relu, out_f32, stream);
    case 1211: return launch_conv_s3_variant<1, 2, 2, 1, true>(c, a, relu, out_f32, stream);
    default: return fail(c, SPVO_ERR_INVALID, "no split-fp32 conv kernel variant for key %d", key);
  }
}

void launch_unpad_s3(const Tensor &t, int batch, float *dst, hipStream_t stream) {
  hipLaunchKernelGGL(unpad_s3_kernel<>, dim3((t.W + 63) / 64, t.H, batch * t.ch), dim3(64), 0, stream, (const unsigned short *)t.d, dst, t.ch, t.H, t.W, t.hp, t.wp);
}

}  // namespace spvo_int

