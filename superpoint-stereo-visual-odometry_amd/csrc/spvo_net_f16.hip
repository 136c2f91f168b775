// spvo_net_f16.hip -- FP16 engines (conv_f16.hip.h): kernel launchers.
#include "spvo_internal.hip.h"
#include "conv_f16.hip.h"

namespace spvo_int {

// ---------------------------------------------------------------- FP16 engines
template <int KS, int CKG, int WR, int WC, bool POOL, bool RELU, bool OUT_F32, int EPI = 0>
int launch_conv16_instance(spvo_ctx *c, ConvArgs16 args, hipStream_t stream) {
  using T = ConvTile16<KS, CKG, WR, WC>;
  auto k = conv_f16_kernel<KS, CKG, WR, WC, POOL, RELU, OUT_F32, EPI>;
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
int launch_conv16_variant(spvo_ctx *c, const ConvArgs16 &a, bool relu, bool out_f32, hipStream_t stream) {
  if constexpr (!POOL) {
    if (out_f32) return relu ? launch_conv16_instance<KS, CKG, WR, WC, false, true, true>(c, a, stream) : launch_conv16_instance<KS, CKG, WR, WC, false, false, true>(c, a, stream);
  }
  return relu ? launch_conv16_instance<KS, CKG, WR, WC, POOL, true, false>(c, a, stream) : launch_conv16_instance<KS, CKG, WR, WC, POOL, false, false>(c, a, stream);
}

// MobileNet 1x1 layers of an FP16 engine: EPI 1 = ReLU, BatchNorm, ReLU (mbv1); EPI 2 = residual add, ReLU (mbv2)
template <int CKG, int WR, int WC, bool POOL>
int launch_conv16_epi(spvo_ctx *c, const ConvArgs16 &a, int epi, hipStream_t stream) {
  return epi == 1 ? launch_conv16_instance<1, CKG, WR, WC, POOL, true, false, 1>(c, a, stream)
                  : launch_conv16_instance<1, CKG, WR, WC, POOL, false, false, 2>(c, a, stream);
}

int launch_conv16(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream) {
  const Tensor &ti = c->tensors[op.in];
  const Tensor &to = c->tensors[op.out];
  const float *tin = (ti.dr[c->cur_ring] ? ti.dr[c->cur_ring] : ti.d) + (size_t)img0 * ti.per_image;
  float *tout = (to.dr[c->cur_ring] ? to.dr[c->cur_ring] : to.d) + (size_t)img0 * to.per_image;
  const bool relu = op.flags & FLAG_RELU, pool = op.flags & FLAG_POOL;
  if (op.type == OP_DWCONV) {
    dim3 grid((ti.W + 63) / 64, (ti.H + 3) / 4, batch * (op.cout / 8));
    if (relu) hipLaunchKernelGGL(dwconv3x3_f16_kernel<true>, grid, dim3(256), 0, stream, (const _Float16 *)tin, (_Float16 *)tout, op.d_w, op.d_b, op.cout / 8, ti.H, ti.W, ti.hp, ti.wp);
    else hipLaunchKernelGGL(dwconv3x3_f16_kernel<false>, grid, dim3(256), 0, stream, (const _Float16 *)tin, (_Float16 *)tout, op.d_w, op.d_b, op.cout / 8, ti.H, ti.W, ti.hp, ti.wp);
    HIP_TRY(c, hipGetLastError());
    return SPVO_OK;
  }
  if (op.cin == 1) {
    dim3 grid((ti.W + 63) / 64, (ti.H + 3) / 4, batch);
    if (to.f16) {
#define SPVO_FIRST16(KS, RELU) hipLaunchKernelGGL((conv_first_f16_kernel<KS, RELU>), grid, dim3(256), 0, stream, tin, (_Float16 *)tout, op.d_w, op.d_b, \
                                                  op.d_bn_scale, op.d_bn_shift, ti.H, ti.W, ti.hp, ti.wp, to.ch / 8, op.out_c_off / 8, op.cout)
      if (op.ks == 3) { if (relu) SPVO_FIRST16(3, true); else SPVO_FIRST16(3, false); }
      else            { if (relu) SPVO_FIRST16(1, true); else SPVO_FIRST16(1, false); }
#undef SPVO_FIRST16
    } else {   // a stem with fewer than 8 channels stays an fp32 plane that holds fp16 values
#define SPVO_FIRST(KS, RELU) hipLaunchKernelGGL((conv_first_kernel<KS, RELU>), grid, dim3(256), 0, stream, tin, tout, op.d_w, op.d_b, \
                                                op.d_bn_scale, op.d_bn_shift, ti.H, ti.W, ti.hp, ti.wp, to.ch, op.out_c_off, op.cout, 1)
      if (op.ks == 3) { if (relu) SPVO_FIRST(3, true); else SPVO_FIRST(3, false); }
      else            { if (relu) SPVO_FIRST(1, true); else SPVO_FIRST(1, false); }
#undef SPVO_FIRST
    }
    HIP_TRY(c, hipGetLastError());
    return SPVO_OK;
  }
  ConvArgs16 a;
  a.in = (const _Float16 *)tin; a.out = tout; a.wpack = op.d_w16;
  a.H = ti.H; a.W = ti.W;
  a.in_hp = ti.hp; a.in_wp = ti.wp; a.in_gtot = ti.ch / 8; a.in_goff = op.in_c_off / 8;
  a.out_hp = to.hp; a.out_wp = to.wp; a.out_ctot = to.ch; a.out_coff = op.out_c_off;
  a.cout = op.cout; a.n_chunks = op.n_chunks; a.co_tiles = op.co_tiles;
  a.tiles_x = a.tiles_y = 0;
  a.batch = batch;
  const bool out_f32 = !to.f16;
  const int key = op.ks * 10000 + (op.ck / 8) * 1000 + op.wr * 100 + op.wc * 10 + (pool ? 1 : 0);   // ks, groups per chunk, wr, wc, pool
  const int epi = (op.flags & FLAG_BN) ? 1 : (op.flags & FLAG_ADD) ? 2 : 0;
  if (epi) {
    a.bn_scale = op.d_bn_scale; a.bn_shift = op.d_bn_shift;
    if (epi == 2) a.residual = (const _Float16 *)(ring_ptr(c, c->tensors[op.residual]) + (size_t)img0 * c->tensors[op.residual].per_image);
    switch (key) {
      case 14220: return launch_conv16_epi<4, 2, 2, false>(c, a, epi, stream);
      case 14120: return launch_conv16_epi<4, 1, 2, false>(c, a, epi, stream);
      case 14110: return launch_conv16_epi<4, 1, 1, false>(c, a, epi, stream);
      case 14221: return launch_conv16_epi<4, 2, 2, true>(c, a, epi, stream);
      case 14211: return launch_conv16_epi<4, 2, 1, true>(c, a, epi, stream);
      case 12220: return launch_conv16_epi<2, 2, 2, false>(c, a, epi, stream);
      case 12120: return launch_conv16_epi<2, 1, 2, false>(c, a, epi, stream);
      case 12110: return launch_conv16_epi<2, 1, 1, false>(c, a, epi, stream);
      case 12221: return launch_conv16_epi<2, 2, 2, true>(c, a, epi, stream);
      case 12211: return launch_conv16_epi<2, 2, 1, true>(c, a, epi, stream);
      default: return fail(c, SPVO_ERR_INVALID, "no fp16 conv kernel variant for key %d with epilogue %d", key, epi);
    }
  }
  switch (key) {
    case 32220: return launch_conv16_variant<3, 2, 2, 2, false>(c, a, relu, out_f32, stream);
    case 32210: return launch_conv16_variant<3, 2, 2, 1, false>(c, a, relu, out_f32, stream);
    case 32120: return launch_conv16_variant<3, 2, 1, 2, false>(c, a, relu, out_f32, stream);
    case 32110: return launch_conv16_variant<3, 2, 1, 1, false>(c, a, relu, out_f32, stream);
    case 32221: return launch_conv16_variant<3, 2, 2, 2, true>(c, a, relu, out_f32, stream);
    case 32211: return launch_conv16_variant<3, 2, 2, 1, true>(c, a, relu, out_f32, stream);
    case 14220: return launch_conv16_variant<1, 4, 2, 2, false>(c, a, relu, out_f32, stream);
    case 14120: return launch_conv16_variant<1, 4, 1, 2, false>(c, a, relu, out_f32, stream);
    case 14110: return launch_conv16_variant<1, 4, 1, 1, false>(c, a, relu, out_f32, stream);
    case 14221: return launch_conv16_variant<1, 4, 2, 2, true>(c, a, relu, out_f32, stream);
    case 14211: return launch_conv16_variant<1, 4, 2, 1, true>(c, a, relu, out_f32, stream);
    case 12220: return launch_conv16_variant<1, 2, 2, 2, false>(c, a, relu, out_f32, stream);
    case 12120: return launch_conv16_variant<1, 2, 1, 2, false>(c, a, relu, out_f32, stream);
    case 12110: return launch_conv16_variant<1, 2, 1, 1, false>(c, a, relu, out_f32, stream);
    case 12221: return launch_conv16_variant<1, 2, 2, 2, true>(c, a, relu, out_f32, stream);
    case 12211: return launch_conv16_variant<1, 2, 2, 1, true>(c, a, relu, out_f32, stream);
    default: return fail(c, SPVO_ERR_INVALID, "no fp16 conv kernel variant for key %d", key);
  }
}

int launch_maxpool_f16(spvo_ctx *c, const Tensor &ti, const Tensor &to, const float *tin, float *tout, int batch, hipStream_t stream) {
  dim3 grid((to.W + 63) / 64, (to.H + 3) / 4, batch * (to.ch / 8));
  hipLaunchKernelGGL(maxpool2_f16_kernel<>, grid, dim3(256), 0, stream, (const _Float16 *)tin, (_Float16 *)tout, to.H, to.W, ti.hp, ti.wp, to.hp, to.wp);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

void launch_unpad_c8(const Tensor &t, int batch, float *dst, hipStream_t stream) {
  hipLaunchKernelGGL(unpad_c8_kernel<>, dim3((t.W + 63) / 64, t.H, batch * t.ch), dim3(64), 0, stream, (const _Float16 *)t.d, dst, t.ch, t.H, t.W, t.hp, t.wp);
}

}  // namespace spvo_int

