// spvo_net_i8.hip -- INT8 engines (conv_i8.hip.h): kernel launchers.
#include "spvo_internal.hip.h"
#include "conv_i8.hip.h"

namespace spvo_int {

// ---------------------------------------------------------------- INT8 engines
template <int KS, int CKG, int WR, int WC, bool POOL, bool RELU, bool OUT_F32, int EPI = 0>
int launch_conv8_instance(spvo_ctx *c, ConvArgs8 args, hipStream_t stream) {
  using T = ConvTile8<KS, CKG, WR, WC>;
  auto k = conv_i8_kernel<KS, CKG, WR, WC, POOL, RELU, OUT_F32, EPI>;
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
int launch_conv8_variant(spvo_ctx *c, const ConvArgs8 &a, bool relu, bool out_f32, int epi, hipStream_t stream) {
  if constexpr (KS == 1) {
    if (epi == 1) return launch_conv8_instance<1, CKG, WR, WC, POOL, true, false, 1>(c, a, stream);
    if (epi == 2) return launch_conv8_instance<1, CKG, WR, WC, POOL, false, false, 2>(c, a, stream);
  }
  if constexpr (!POOL) {
    if (out_f32) return relu ? launch_conv8_instance<KS, CKG, WR, WC, false, true, true>(c, a, stream) : launch_conv8_instance<KS, CKG, WR, WC, false, false, true>(c, a, stream);
  }
  return relu ? launch_conv8_instance<KS, CKG, WR, WC, POOL, true, false>(c, a, stream) : launch_conv8_instance<KS, CKG, WR, WC, POOL, false, false>(c, a, stream);
}

int launch_conv8(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream) {
  const Tensor &ti = c->tensors[op.in];
  const Tensor &to = c->tensors[op.out];
  const float *tin = (ti.dr[c->cur_ring] ? ti.dr[c->cur_ring] : ti.d) + (size_t)img0 * ti.per_image;
  float *tout = (to.dr[c->cur_ring] ? to.dr[c->cur_ring] : to.d) + (size_t)img0 * to.per_image;
  const bool relu = op.flags & FLAG_RELU, pool = op.flags & FLAG_POOL;
  if (op.type == OP_DWCONV) {
    dim3 grid((ti.W + 63) / 64, (ti.H + 3) / 4, batch * (op.cout / 16));
    if (relu) hipLaunchKernelGGL(dwconv3x3_i8_kernel<true>, grid, dim3(256), 0, stream, (const int8_t *)tin, (int8_t *)tout, op.d_wq32, op.d_qm, op.d_b, op.inv_s_out, op.cout / 16, ti.H, ti.W, ti.hp, ti.wp);
    else hipLaunchKernelGGL(dwconv3x3_i8_kernel<false>, grid, dim3(256), 0, stream, (const int8_t *)tin, (int8_t *)tout, op.d_wq32, op.d_qm, op.d_b, op.inv_s_out, op.cout / 16, ti.H, ti.W, ti.hp, ti.wp);
    HIP_TRY(c, hipGetLastError());
    return SPVO_OK;
  }
  if (op.cin == 1) {   // fp32 stem
    dim3 grid((ti.W + 63) / 64, (ti.H + 3) / 4, batch);
#define SPVO_STEM8(KS, RELU, OUTQ) hipLaunchKernelGGL((conv_first_i8_kernel<KS, RELU, OUTQ>), grid, dim3(256), 0, stream, tin, (void *)tout, op.d_w, op.d_b, \
                                                      op.d_bn_scale, op.d_bn_shift, op.inv_s_out, ti.H, ti.W, ti.hp, ti.wp, to.ch, op.out_c_off, op.cout)
    if (to.i8) {
      if (op.ks == 3) { if (relu) SPVO_STEM8(3, true, true); else SPVO_STEM8(3, false, true); }
      else            { if (relu) SPVO_STEM8(1, true, true); else SPVO_STEM8(1, false, true); }
    } else {
      if (op.ks == 3) { if (relu) SPVO_STEM8(3, true, false); else SPVO_STEM8(3, false, false); }
      else            { if (relu) SPVO_STEM8(1, true, false); else SPVO_STEM8(1, false, false); }
    }
#undef SPVO_STEM8
    HIP_TRY(c, hipGetLastError());
    return SPVO_OK;
  }
  ConvArgs8 a;
  a.in = (const int8_t *)tin; a.out = tout; a.wpack = op.d_w8; a.qm = op.d_qm; a.bias = op.d_b;
  a.bn_scale = op.d_bn_scale; a.bn_shift = op.d_bn_shift;
  a.inv_s_out = op.inv_s_out; a.s_res = op.s_res;
  a.H = ti.H; a.W = ti.W;
  a.in_hp = ti.hp; a.in_wp = ti.wp; a.in_gtot = ti.ch / 16; a.in_goff = op.in_c_off / 16;
  a.out_hp = to.hp; a.out_wp = to.wp; a.out_ctot = to.ch; a.out_coff = op.out_c_off;
  a.cout = op.cout; a.n_chunks = op.n_chunks; a.co_tiles = op.co_tiles;
  a.tiles_x = a.tiles_y = 0;
  a.batch = batch;
  const int epi = (op.flags & FLAG_BN) ? 1 : (op.flags & FLAG_ADD) ? 2 : 0;
  if (epi == 2) a.residual = (const int8_t *)(ring_ptr(c, c->tensors[op.residual]) + (size_t)img0 * c->tensors[op.residual].per_image);
  const bool out_f32 = !to.i8;
  const int key = op.ks * 10000 + (op.ck / 16) * 1000 + op.wr * 100 + op.wc * 10 + (pool ? 1 : 0);   // ks, groups per chunk, wr, wc, pool
  switch (key) {
    case 32220: return launch_conv8_variant<3, 2, 2, 2, false>(c, a, relu, out_f32, epi, stream);
    case 32210: return launch_conv8_variant<3, 2, 2, 1, false>(c, a, relu, out_f32, epi, stream);
    case 32120: return launch_conv8_variant<3, 2, 1, 2, false>(c, a, relu, out_f32, epi, stream);
    case 32110: return launch_conv8_variant<3, 2, 1, 1, false>(c, a, relu, out_f32, epi, stream);
    case 32221: return launch_conv8_variant<3, 2, 2, 2, true>(c, a, relu, out_f32, epi, stream);
    case 32211: return launch_conv8_variant<3, 2, 2, 1, true>(c, a, relu, out_f32, epi, stream);
    case 14220: return launch_conv8_variant<1, 4, 2, 2, false>(c, a, relu, out_f32, epi, stream);
    case 14120: return launch_conv8_variant<1, 4, 1, 2, false>(c, a, relu, out_f32, epi, stream);
    case 14110: return launch_conv8_variant<1, 4, 1, 1, false>(c, a, relu, out_f32, epi, stream);
    case 14221: return launch_conv8_variant<1, 4, 2, 2, true>(c, a, relu, out_f32, epi, stream);
    case 14211: return launch_conv8_variant<1, 4, 2, 1, true>(c, a, relu, out_f32, epi, stream);
    case 12220: return launch_conv8_variant<1, 2, 2, 2, false>(c, a, relu, out_f32, epi, stream);
    case 12120: return launch_conv8_variant<1, 2, 1, 2, false>(c, a, relu, out_f32, epi, stream);
    case 12110: return launch_conv8_variant<1, 2, 1, 1, false>(c, a, relu, out_f32, epi, stream);
    case 12221: return launch_conv8_variant<1, 2, 2, 2, true>(c, a, relu, out_f32, epi, stream);
    case 12211: return launch_conv8_variant<1, 2, 2, 1, true>(c, a, relu, out_f32, epi, stream);
    default: return fail(c, SPVO_ERR_INVALID, "no int8 conv kernel variant for key %d", key);
  }
}

void launch_unpad_c16(const Tensor &t, int batch, float *dst, hipStream_t stream) {
  hipLaunchKernelGGL(unpad_c16_kernel<>, dim3((t.W + 63) / 64, t.H, batch * t.ch), dim3(64), 0, stream, (const int8_t *)t.d, dst, t.ch, t.H, t.W, t.hp, t.wp);
}

}  // namespace spvo_int

