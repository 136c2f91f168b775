// spvo_net_i8.hip -- INT8 engines (conv_i8.hip.h): kernel launchers.
#include "spvo_internal.hip.h"
#include "conv_i8.hip.h"
#include "conv_i8_fused.hip.h"
#include "heads_i8.hip.h"

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

// ---------------------------------------------------------------- MobileNet blocks as one launch (conv_i8_fused.hip.h)
// A depthwise 3x3 (+ ReLU) whose only reader is the pointwise 1x1 (+ ReLU [+ BatchNorm] [+ pool]) behind it, both int8, 64 or 128
// channels in, 64 or 128 out: the pointwise op's launch does both (`fused_dw`), the depthwise op launches nothing (`fused_away`).
// If that block is ops 2, 3 of the plan and ops 0, 1 are the sp_mbv1 stem (3x3 1 -> 1 + ReLU on the fp32 input plane, 1x1 1 -> 64 +
// ReLU + BatchNorm into int8), the same launch computes the stem as well (`fused_stem`).
void plan_int8_fusion(spvo_ctx *c) {
  auto readers = [&](int tensor) { int n = 0; for (const auto &o : c->ops) n += (o.in == tensor) + ((o.flags & FLAG_ADD) && o.residual == tensor); return n; };
  for (size_t i = 0; i + 1 < c->ops.size(); ++i) {
    Op &dw = c->ops[i], &pw = c->ops[i + 1];
    if (dw.type != OP_DWCONV || pw.type != OP_CONV || pw.ks != 1 || pw.in != dw.out || pw.in_c_off || pw.out_c_off || pw.cin != dw.cout) continue;
    if (dw.flags != FLAG_RELU || !(pw.flags & FLAG_RELU) || (pw.flags & FLAG_ADD)) continue;
    const Tensor &ti = c->tensors[dw.in], &tm = c->tensors[dw.out], &to = c->tensors[pw.out];
    if (!ti.i8 || !tm.i8 || !to.i8 || (dw.cout != 64 && dw.cout != 128) || (pw.cout != 64 && pw.cout != 128) || to.ch != pw.cout) continue;
    if (readers(dw.out) != 1 || (int)dw.out == c->t_det || (int)dw.out == c->t_desc || !dw.d_wsel || !pw.d_w8) continue;
    if ((pw.flags & FLAG_POOL) && ((ti.H | ti.W) & 1)) continue;
    pw.fused_dw = (int)i;
    dw.fused_away = true;
    if (i == 2) {
      Op &s0 = c->ops[0], &s1 = c->ops[1];
      const bool stem = s0.type == OP_CONV && s0.cin == 1 && s0.cout == 1 && s0.ks == 3 && s0.flags == FLAG_RELU && s0.in == c->t_input && !c->tensors[s0.out].i8 &&
                        s1.type == OP_CONV && s1.cin == 1 && s1.cout == 64 && s1.ks == 1 && s1.flags == (FLAG_RELU | FLAG_BN) && s1.in == s0.out && s1.out == dw.in &&
                        !s1.out_c_off && dw.cout == 64 && readers(s0.out) == 1 && readers(s1.out) == 1 && s1.d_bn_scale;
      if (stem) { pw.fused_stem = true; s0.fused_away = s1.fused_away = true; }
    }
  }
}

template <int G, bool STEM, bool POOL, int EPI>
static int launch_dwpw8_instance(spvo_ctx *c, const DwPwArgs8 &a, hipStream_t stream) {
  using T = DwPwTile<G, STEM>;
  auto k = dwpw_i8_kernel<G, STEM, POOL, EPI>;
  static int per_cu[64] = {};
  const int dev = c->cfg.device & 63;
  if (!per_cu[dev]) {
    HIP_TRY(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    int n = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k, DWPW_THREADS, T::LDS_BYTES));
    per_cu[dev] = std::max(n, 1);
  }
  const int n_tiles = a.tiles_x * a.tiles_y * a.batch;
  const int force = tuning("int8_fused", 1);   // (measurements: 2, 3, 4 = one, two, three workgroups per CU)
  const int wg_per_cu = force > 1 ? force - 1 : per_cu[dev];
  hipLaunchKernelGGL(k, dim3(std::min(n_tiles, c->num_cus * wg_per_cu)), dim3(DWPW_THREADS), T::LDS_BYTES, stream, a);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

static int launch_dwpw8(spvo_ctx *c, const Op &pw, int img0, int batch, hipStream_t stream) {
  const Op &dw = c->ops[pw.fused_dw];
  const Tensor &ti = c->tensors[dw.in], &tm = c->tensors[dw.out], &to = c->tensors[pw.out];
  DwPwArgs8 a;
  a.hp = ti.hp; a.wp = ti.wp; a.H = ti.H; a.W = ti.W;
  a.in = (const int8_t *)(ring_ptr(c, ti) + (size_t)img0 * ti.per_image);
  a.in_per_image = ti.per_image * 4;
  a.dw_wsel = dw.d_wsel; a.dw_qm = dw.d_qm; a.dw_bias = dw.d_b; a.inv_s_dw = dw.inv_s_out;
  a.pw_w = pw.d_w8; a.pw_qm = pw.d_qm; a.pw_bias = pw.d_b; a.bn_scale = pw.d_bn_scale; a.bn_shift = pw.d_bn_shift; a.inv_s_out = pw.inv_s_out;
  a.out = (int8_t *)(ring_ptr(c, to) + (size_t)img0 * to.per_image);
  a.out_per_image = to.per_image * 4;
  a.out_hp = to.hp; a.out_wp = to.wp; a.cout = pw.cout; a.co_tiles = pw.cout / CO_TILE;
  a.tiles_x = (ti.W + 31) / 32; a.tiles_y = (ti.H + 7) / 8; a.batch = batch;
  const bool keep = c->heads_keep_raw;   // the synchronous entry points expose every tensor (spvo_debug_tensor): the skipped ones are stored too
  if (keep) { a.dbg_dw_out = (int8_t *)(tm.d + (size_t)img0 * tm.per_image); a.dbg_c16_per_image = tm.per_image * 4; }
  const bool pool = pw.flags & FLAG_POOL, bn = pw.flags & FLAG_BN;
  if (pw.fused_stem) {
    const Op &s0 = c->ops[0], &s1 = c->ops[1];
    const Tensor &t0 = c->tensors[s0.in], &t1 = c->tensors[s0.out];
    a.in = nullptr;
    a.in_f32 = ring_ptr(c, t0) + (size_t)img0 * t0.per_image;
    a.in_per_image = t0.per_image;
    a.w0 = s0.d_w; a.b0 = s0.d_b; a.w1 = s1.d_w; a.b1 = s1.d_b; a.bn1_scale = s1.d_bn_scale; a.bn1_shift = s1.d_bn_shift; a.inv_s_stem = s1.inv_s_out;
    if (keep) { a.dbg_stem_plane = t1.d + (size_t)img0 * t1.per_image; a.dbg_stem_plane_per_image = t1.per_image; a.dbg_stem_out = (int8_t *)(ti.d + (size_t)img0 * ti.per_image); }
    if (pool) return bn ? launch_dwpw8_instance<4, true, true, 1>(c, a, stream) : launch_dwpw8_instance<4, true, true, 0>(c, a, stream);
    return bn ? launch_dwpw8_instance<4, true, false, 1>(c, a, stream) : launch_dwpw8_instance<4, true, false, 0>(c, a, stream);
  }
  if (dw.cout == 64) {
    if (pool) return bn ? launch_dwpw8_instance<4, false, true, 1>(c, a, stream) : launch_dwpw8_instance<4, false, true, 0>(c, a, stream);
    return bn ? launch_dwpw8_instance<4, false, false, 1>(c, a, stream) : launch_dwpw8_instance<4, false, false, 0>(c, a, stream);
  }
  if (pool) return bn ? launch_dwpw8_instance<8, false, true, 1>(c, a, stream) : launch_dwpw8_instance<8, false, true, 0>(c, a, stream);
  return bn ? launch_dwpw8_instance<8, false, false, 1>(c, a, stream) : launch_dwpw8_instance<8, false, false, 0>(c, a, stream);
}

int launch_conv8(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream) {
  if (op.fused_dw >= 0) return launch_dwpw8(c, op, img0, batch, stream);
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

// the fused tail of an INT8 engine (heads_i8.hip.h): ops head_start .. head_start + 2 in one launch
int launch_heads8(spvo_ctx *c, int batch, hipStream_t stream) {
  const Op &pb = c->ops[c->head_start], &db = c->ops[c->head_start + 1];
  const Tensor &ti = c->tensors[pb.in], &te = c->tensors[db.in], &tdet = c->tensors[pb.out], &traw = c->tensors[db.out], &tdesc = c->tensors[c->t_desc];
  const size_t plane = (size_t)ti.hp * ti.wp;
  HeadsArgs8 a;
  a.in_det = (const int8_t *)(ring_ptr(c, ti)) + (size_t)(pb.in_c_off / 16) * plane * 16; a.det_in_per_image = ti.per_image * 4;
  a.in_desc = (const int8_t *)(ring_ptr(c, te)) + (size_t)(db.in_c_off / 16) * plane * 16; a.desc_in_per_image = te.per_image * 4;
  a.in_hp = ti.hp; a.in_wp = ti.wp;
  a.wpack = c->d_heads_w8; a.qm = c->d_heads_qm; a.bias = c->d_heads_b;
  a.det = ring_ptr(c, tdet); a.det_per_image = tdet.per_image;
  a.desc_raw = c->heads_keep_raw ? ring_ptr(c, traw) : nullptr; a.raw_per_image = traw.per_image;
  a.desc = ring_ptr(c, tdesc);
  a.H = ti.H; a.W = ti.W; a.batch = batch;
  const double hbytes = batch * ((double)(pb.cin + db.cin) * ti.H * ti.W + (double)(pb.cout + (a.desc_raw ? 2 : 1) * db.cout) * ti.H * ti.W * 4) + (double)(pb.cout * pb.cin + db.cout * db.cin);
  ScopedStage st(c, stage_id(c, "heads"), (pb.flops_per_image + db.flops_per_image) * batch, hbytes, stream);
  const int ntiles = (batch * ti.H * ti.W + 31) / 32;
  hipLaunchKernelGGL(heads_i8_kernel<>, dim3(ntiles), dim3(HEADS8_THREADS), 0, stream, a);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

void launch_unpad_c16(const Tensor &t, int batch, float *dst, hipStream_t stream) {
  hipLaunchKernelGGL(unpad_c16_kernel<>, dim3((t.W + 63) / 64, t.H, batch * t.ch), dim3(64), 0, stream, (const int8_t *)t.d, dst, t.ch, t.H, t.W, t.hp, t.wp);
}

}  // namespace spvo_int

