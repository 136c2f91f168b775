// spvo_net_f32.hip -- FP32 engines: launchers of the direct (conv_mfma.hip.h) and Winograd (conv_wino2 / conv_wino4.hip.h) convolution kernels and the
// layer executor that walks a plan (replaces the TensorRT enqueue at feature_detection_neural_network.cpp:169).
#include "spvo_internal.hip.h"
#include "conv_mfma.hip.h"
#include "conv_wino2.hip.h"
#include "conv_wino4.hip.h"
#include "heads.hip.h"

namespace spvo_int {

template <int KS, int CK, int WR, int WC, bool POOL, bool RELU, int EPI = 0, int MINW = 1, int TAG = 0>
int launch_conv_instance(spvo_ctx *c, ConvArgs args, hipStream_t stream) {
  using T = ConvTile<KS, CK, WR, WC>;
  auto k = conv_mfma_kernel<KS, CK, WR, WC, POOL, RELU, MINW, 0, EPI, TAG>;
  static int per_cu[64] = {};   // resident workgroups per CU of this instance, per device
  const int dev = c->cfg.device & 63;
  if (!per_cu[dev]) {
    HIP_TRY(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, T::LDS_BYTES));
    int n = 0;
    HIP_TRY(c, hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, (const void *)k, 256, T::LDS_BYTES));
    per_cu[dev] = std::max(n, 1);
  }
  // persistent grid: what the chip holds at once; each workgroup walks tiles with that stride
  const int n_tiles = args.tiles_x * args.tiles_y * args.co_tiles * args.batch;
  const int grid = std::min(n_tiles, c->num_cus * per_cu[dev]);
  hipLaunchKernelGGL(k, dim3(grid), dim3(256), T::LDS_BYTES, stream, args);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

template <int KS, int CK, int WR, int WC, bool POOL, int MINW = 1>
int launch_conv_variant(spvo_ctx *c, const ConvArgs &a, int batch, bool relu, hipStream_t stream, bool dominant = false) {
  using T = ConvTile<KS, CK, WR, WC>;
  ConvArgs args = a;
  args.tiles_x = (a.W + T::TW - 1) / T::TW;
  args.tiles_y = (a.H + T::TH - 1) / T::TH;
  args.batch = batch;
  if constexpr (KS == 3) {
    if (dominant && relu) return launch_conv_instance<KS, CK, WR, WC, POOL, true, 0, MINW, 1>(c, args, stream);   // own kernel name
  }
  return relu ? launch_conv_instance<KS, CK, WR, WC, POOL, true, 0, MINW>(c, args, stream)
              : launch_conv_instance<KS, CK, WR, WC, POOL, false, 0, MINW>(c, args, stream);
}

// MobileNet 1x1 layers: EPI 1 = ReLU, BatchNorm, ReLU (mbv1); EPI 2 = residual add, ReLU (mbv2)
template <int WR, int WC, bool POOL>
int launch_conv_epi(spvo_ctx *c, const ConvArgs &a, int batch, int epi, hipStream_t stream) {
  using T = ConvTile<1, 16, WR, WC>;
  ConvArgs args = a;
  args.tiles_x = (a.W + T::TW - 1) / T::TW;
  args.tiles_y = (a.H + T::TH - 1) / T::TH;
  args.batch = batch;
  return epi == 1 ? launch_conv_instance<1, 16, WR, WC, POOL, true, 1>(c, args, stream)
                  : launch_conv_instance<1, 16, WR, WC, POOL, false, 2>(c, args, stream);
}

// Winograd F(2x2, 3x3) instance of a 3x3 layer (conv_wino2.hip.h): one tile shape, one workgroup of 512 threads per CU (157 KB of LDS)
template <bool POOL, bool RELU, int TAG, bool ODD = false, bool NARROW = false>
int launch_conv_wino_instance(spvo_ctx *c, const ConvArgs &args, hipStream_t stream) {
  static bool ready[64] = {};
  const int dev = c->cfg.device & 63;
  const int n_tiles = args.tiles_x * args.tiles_y * args.co_tiles * args.batch;
  // One workgroup per CU, each walking ceil(n_tiles / grid) tiles.  The grid is the SMALLEST one that keeps that
  // number of rounds: 3330 tiles are 14 rounds on 256 CUs and still 14 rounds on 238, and the 18 CUs left over take the
  // small kernels of the other streams (tail of the previous pair, solver): with all 256 CUs claimed, any of those
  // kernels sitting on a CU when a layer starts keeps that layer's last workgroup waiting for a CU.
  const int rounds = (n_tiles + c->num_cus - 1) / c->num_cus;
  const int grid = (n_tiles + rounds - 1) / rounds;
  auto k = conv_wino2_kernel<POOL, RELU, TAG, ODD, NARROW>;
  if (!ready[dev]) {
    HIP_TRY(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, WINO2_LDS_BYTES));
    ready[dev] = true;
  }
  ConvArgs a2 = args;
  if (rounds < 2 || args.n_chunks < 4) a2.sched = nullptr;   // one tile per workgroup: nothing to hand out; short K loops: see the kernel
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), WINO2_LDS_BYTES, stream, a2);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

int launch_conv_wino_sel(spvo_ctx *c, const ConvArgs &args, bool relu, bool pool, bool dominant, bool narrow, hipStream_t stream) {
  const ConvArgs &a = args;
  if (narrow) {   // 32 output channels per workgroup (layers that would leave CUs idle with 64)
    if (!pool && ((a.H | a.W) & 1)) return relu ? launch_conv_wino_instance<false, true, 0, true, true>(c, args, stream) : launch_conv_wino_instance<false, false, 0, true, true>(c, args, stream);
    if (pool) return relu ? launch_conv_wino_instance<true, true, 0, false, true>(c, args, stream) : launch_conv_wino_instance<true, false, 0, false, true>(c, args, stream);
    return relu ? launch_conv_wino_instance<false, true, 0, false, true>(c, args, stream) : launch_conv_wino_instance<false, false, 0, false, true>(c, args, stream);
  }
  if (!pool && ((a.H | a.W) & 1)) return relu ? launch_conv_wino_instance<false, true, 0, true>(c, args, stream) : launch_conv_wino_instance<false, false, 0, true>(c, args, stream);
  if (dominant && relu) return pool ? launch_conv_wino_instance<true, true, 1, false>(c, args, stream) : launch_conv_wino_instance<false, true, 1, false>(c, args, stream);
  if (pool) return relu ? launch_conv_wino_instance<true, true, 0, false>(c, args, stream) : launch_conv_wino_instance<true, false, 0, false>(c, args, stream);
  return relu ? launch_conv_wino_instance<false, true, 0, false>(c, args, stream) : launch_conv_wino_instance<false, false, 0, false>(c, args, stream);
}

// conv_wino4.hip.h: Winograd F(4x4,3x3), 16 x 32 output tiles, 8 waves, one workgroup per CU
template <bool POOL, bool RELU, int TAG>
int launch_conv_wino4_instance(spvo_ctx *c, const ConvArgs &args, hipStream_t stream) {
  static bool ready[64] = {};
  const int dev = c->cfg.device & 63;
  auto k = conv_wino4_kernel<POOL, RELU, TAG>;
  if (!ready[dev]) {
    HIP_TRY(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, Wino4Tile::LDS_BYTES));
    ready[dev] = true;
  }
  const int n_tiles = args.tiles_x * args.tiles_y * args.co_tiles * args.batch;
  const int rounds = (n_tiles + c->num_cus - 1) / c->num_cus;
  const int grid = (n_tiles + rounds - 1) / rounds;   // the smallest grid that keeps the number of rounds (see launch_conv_wino_instance)
  ConvArgs a2 = args;
  if (rounds < 2 || args.n_chunks < 5) a2.sched = nullptr;   // (the raw-tile cursor runs three items ahead: a tile's successor must be known by the end of its second item)
  hipLaunchKernelGGL(k, dim3(grid), dim3(512), Wino4Tile::LDS_BYTES, stream, a2);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

int launch_conv_wino4(spvo_ctx *c, const ConvArgs &a, int batch, bool relu, bool pool, bool dominant, hipStream_t stream) {
  ConvArgs args = a;
  args.tiles_x = (a.W + Wino4Tile::TW - 1) / Wino4Tile::TW;
  args.tiles_y = (a.H + Wino4Tile::TH - 1) / Wino4Tile::TH;
  args.batch = batch;
  if (pool && ((a.H | a.W) & 1)) return fail(c, SPVO_ERR_INVALID, "F(4x4) kernel: pooled layer with odd size");   // (the plan loader keeps those on the other kernels)
  if (dominant && relu) return pool ? launch_conv_wino4_instance<true, true, 1>(c, args, stream) : launch_conv_wino4_instance<false, true, 1>(c, args, stream);
  if (pool) return relu ? launch_conv_wino4_instance<true, true, 0>(c, args, stream) : launch_conv_wino4_instance<true, false, 0>(c, args, stream);
  return relu ? launch_conv_wino4_instance<false, true, 0>(c, args, stream) : launch_conv_wino4_instance<false, false, 0>(c, args, stream);
}

int launch_conv_wino(spvo_ctx *c, const ConvArgs &a, int batch, bool relu, bool pool, bool dominant, bool narrow, hipStream_t stream) {
  ConvArgs args = a;
  args.tiles_x = (a.W + WinoTile::TW - 1) / WinoTile::TW;
  args.tiles_y = (a.H + WinoTile::TH - 1) / WinoTile::TH;
  args.batch = batch;
  return launch_conv_wino_sel(c, args, relu, pool, dominant, narrow, stream);
}

// images [img0, img0 + batch) of the tensors, on `stream`
int launch_conv(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream) {
  const Tensor &ti = c->tensors[op.in];
  const Tensor &to = c->tensors[op.out];
  const float *tin = (ti.dr[c->cur_ring] ? ti.dr[c->cur_ring] : ti.d) + (size_t)img0 * ti.per_image;
  float *tout = (to.dr[c->cur_ring] ? to.dr[c->cur_ring] : to.d) + (size_t)img0 * to.per_image;
  const bool relu = op.flags & FLAG_RELU, pool = op.flags & FLAG_POOL;
  const int epi = (op.flags & FLAG_BN) ? 1 : (op.flags & FLAG_ADD) ? 2 : 0;
  if (op.type == OP_DWCONV) {
    dim3 grid((ti.W + 255) / 256, (ti.H + 3) / 4, batch * op.cout);
    if (relu)
      hipLaunchKernelGGL(dwconv3x3_kernel<true>, grid, dim3(256), 0, stream, tin, tout, op.d_w, op.d_b, op.cout, ti.H, ti.W, ti.hp, ti.wp);
    else
      hipLaunchKernelGGL(dwconv3x3_kernel<false>, grid, dim3(256), 0, stream, tin, tout, op.d_w, op.d_b, op.cout, ti.H, ti.W, ti.hp, ti.wp);
    HIP_TRY(c, hipGetLastError());
    return SPVO_OK;
  }
  if (op.cin == 1) {
    dim3 grid((ti.W + 63) / 64, (ti.H + 3) / 4, batch);
#define SPVO_FIRST(KS, RELU) hipLaunchKernelGGL((conv_first_kernel<KS, RELU>), grid, dim3(256), 0, stream, tin, tout, op.d_w, op.d_b, \
                                                op.d_bn_scale, op.d_bn_shift, ti.H, ti.W, ti.hp, ti.wp, to.ch, op.out_c_off, op.cout, 0)
    if (op.ks == 3 && !op.d_bn_scale) {   // 4 pixels per thread: 16-byte stores
      dim3 g4((ti.W + 255) / 256, (ti.H + 3) / 4, batch);
      if (relu) hipLaunchKernelGGL(conv_first4_kernel<true>, g4, dim3(256), 0, stream, tin, tout, op.d_w, op.d_b, ti.H, ti.W, ti.hp, ti.wp, to.ch, op.out_c_off, op.cout);
      else hipLaunchKernelGGL(conv_first4_kernel<false>, g4, dim3(256), 0, stream, tin, tout, op.d_w, op.d_b, ti.H, ti.W, ti.hp, ti.wp, to.ch, op.out_c_off, op.cout);
    } else if (op.ks == 3) { if (relu) SPVO_FIRST(3, true); else SPVO_FIRST(3, false); }
    else            { if (relu) SPVO_FIRST(1, true); else SPVO_FIRST(1, false); }
#undef SPVO_FIRST
    HIP_TRY(c, hipGetLastError());
    return SPVO_OK;
  }
  ConvArgs a;
  a.in = tin; a.out = tout; a.wpack = op.d_w; a.bias = op.d_b;
  a.H = ti.H; a.W = ti.W;
  a.in_hp = ti.hp; a.in_wp = ti.wp; a.in_ctot = ti.ch; a.in_coff = op.in_c_off;
  a.out_hp = to.hp; a.out_wp = to.wp; a.out_ctot = to.ch; a.out_coff = op.out_c_off;
  a.cout = op.cout; a.n_chunks = op.n_chunks; a.co_tiles = op.co_tiles;
  a.tiles_x = a.tiles_y = 0;
  a.batch = batch;
  a.sched = op.d_sched;
  if (op.wino4) return launch_conv_wino4(c, a, batch, relu, pool, op.dominant, stream);
  if (op.wino) return launch_conv_wino(c, a, batch, relu, pool, op.dominant, op.wino_narrow, stream);
  const int key = op.ks * 10000 + op.ck * 100 + op.wr * 20 + op.wc * 2 + (pool ? 1 : 0);   // ks, ck, wr, wc, pool
  if (epi) {
    a.bn_scale = op.d_bn_scale; a.bn_shift = op.d_bn_shift;
    if (epi == 2) a.residual = ring_ptr(c, c->tensors[op.residual]) + (size_t)img0 * c->tensors[op.residual].per_image;
    switch (key) {
      case 11644: return launch_conv_epi<2, 2, false>(c, a, batch, epi, stream);
      case 11624: return launch_conv_epi<1, 2, false>(c, a, batch, epi, stream);
      case 11622: return launch_conv_epi<1, 1, false>(c, a, batch, epi, stream);
      case 11645: return launch_conv_epi<2, 2, true>(c, a, batch, epi, stream);
      case 11643: return launch_conv_epi<2, 1, true>(c, a, batch, epi, stream);
      default: return fail(c, SPVO_ERR_INVALID, "no conv kernel variant for key %d with epilogue %d", key, epi);
    }
  }
  switch (key) {   // third template argument of the launch helper = waves per SIMD the register budget allows
    case 30844: return launch_conv_variant<3, 8, 2, 2, false, 1>(c, a, batch, relu, stream, op.dominant);
    case 30842: return launch_conv_variant<3, 8, 2, 1, false, 2>(c, a, batch, relu, stream, op.dominant);
    case 30824: return launch_conv_variant<3, 8, 1, 2, false, 2>(c, a, batch, relu, stream, op.dominant);
    case 30822: return launch_conv_variant<3, 8, 1, 1, false, 4>(c, a, batch, relu, stream, op.dominant);
    case 30845: return launch_conv_variant<3, 8, 2, 2, true, 1>(c, a, batch, relu, stream, op.dominant);
    case 30843: return launch_conv_variant<3, 8, 2, 1, true, 2>(c, a, batch, relu, stream, op.dominant);
    case 11644: return launch_conv_variant<1, 16, 2, 2, false>(c, a, batch, relu, stream);
    case 11624: return launch_conv_variant<1, 16, 1, 2, false>(c, a, batch, relu, stream);
    case 11622: return launch_conv_variant<1, 16, 1, 1, false>(c, a, batch, relu, stream);
    case 11645: return launch_conv_variant<1, 16, 2, 2, true>(c, a, batch, relu, stream);
    case 11643: return launch_conv_variant<1, 16, 2, 1, true>(c, a, batch, relu, stream);
    default: return fail(c, SPVO_ERR_INVALID, "no conv kernel variant for key %d", key);
  }
}

int launch_op(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream) {
  if (op.merged || op.fused_away) return SPVO_OK;
  const Tensor &ti = c->tensors[op.in];
  const Tensor &to = c->tensors[op.out];
  if (op.type == OP_CONV || op.type == OP_DWCONV) {
    // algorithmic HBM bytes of the launch: the input channels it reads, the output channels it writes (after the
    // fused pooling), its weights once -- interiors only, in the element size of the engine's tensors
    auto esz = [](const Tensor &t) { return t.i8 ? 1.0 : (t.f16 || t.s3) ? (t.s3 ? 6.0 : 2.0) : 4.0; };
    const double wsz = c->int8 ? 1.0 : (c->fp16 ? 2.0 : (c->s3 ? 6.0 : 4.0));
    const double bytes = batch * ((double)op.cin * ti.H * ti.W * esz(ti) + (double)op.cout * to.H * to.W * esz(to)) +
                         (double)op.cout * (op.type == OP_DWCONV ? 1 : op.cin) * op.ks * op.ks * wsz;
    if (op.fused_dw >= 0) {   // INT8 MobileNet block in one launch (spvo_net_i8.hip): what the depthwise op (and the stem) would have done counts here
      const Op &dw = c->ops[op.fused_dw];
      const Tensor &t0 = c->tensors[op.fused_stem ? c->ops[0].in : dw.in];
      double fl = op.flops_per_image + dw.flops_per_image, by = (double)t0.ch * t0.H * t0.W * esz(t0) + (double)op.cout * to.H * to.W * esz(to);
      if (op.fused_stem) fl += c->ops[0].flops_per_image + c->ops[1].flops_per_image;
      ScopedStage st(c, op.stage, fl * batch, batch * by + (double)op.cout * op.cin + 9.0 * dw.cout, stream);
      return launch_conv8(c, op, img0, batch, stream);
    }
    ScopedStage st(c, op.stage, op.flops_per_image * batch, bytes, stream);
    return c->int8 ? launch_conv8(c, op, img0, batch, stream) : c->fp16 ? launch_conv16(c, op, img0, batch, stream)
           : c->s3 ? launch_conv_s3(c, op, img0, batch, stream) : launch_conv(c, op, img0, batch, stream);
  }
  const float *tin = (ti.dr[c->cur_ring] ? ti.dr[c->cur_ring] : ti.d) + (size_t)img0 * ti.per_image;
  float *tout = (to.dr[c->cur_ring] ? to.dr[c->cur_ring] : to.d) + (size_t)img0 * to.per_image;
  // (max-pool / L2 normalisation: read one tensor, write one -- in the element sizes of the engine's tensors)
  auto tsz = [](const Tensor &t) { return t.i8 ? 1.0 : t.f16 ? 2.0 : t.s3 ? 6.0 : 4.0; };
  ScopedStage st(c, op.stage, 0, batch * ((double)ti.ch * ti.H * ti.W * tsz(ti) + (double)to.ch * to.H * to.W * tsz(to)), stream);
  if (op.type == OP_MAXPOOL && ti.f16) {
    return launch_maxpool_f16(c, ti, to, tin, tout, batch, stream);
  } else if (op.type == OP_MAXPOOL) {
    dim3 grid((to.W + 63) / 64, (to.H + 3) / 4, batch * to.ch);
    hipLaunchKernelGGL(maxpool2_kernel<>, grid, dim3(256), 0, stream, tin, tout, to.ch, to.H, to.W, ti.hp, ti.wp, to.hp, to.wp);
  } else if (op.type == OP_L2NORM) {
    // 16 pixels per block: 3-4 blocks per CU hide each other's latency (measured 15 us vs 21 us with 32 pixels)
    hipLaunchKernelGGL((l2norm_nhwc_kernel<256, 16>), dim3((ti.W + 15) / 16, ti.H, batch), dim3(256), 0, stream, tin, tout, ti.H, ti.W, ti.hp, ti.wp);
  }
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

// Both images of a stereo pair go through every layer in ONE launch (they are independent, the batch index is part
// of the tile id).  Per-image streams for the small layers were measured and gave nothing: two persistent
// kernels do not backfill each other's ragged ends.
// ops [first, last) on `stream`
// the fused tail of the graph (heads.hip.h): ops head_start .. head_start + 2 in one launch
int launch_heads(spvo_ctx *c, int batch, hipStream_t stream) {
  const Op &pb = c->ops[c->head_start], &db = c->ops[c->head_start + 1];
  const Tensor &ti = c->tensors[pb.in], &te = c->tensors[db.in], &tdet = c->tensors[pb.out], &traw = c->tensors[db.out], &tdesc = c->tensors[c->t_desc];
  static bool ready[64] = {};
  const int dev = c->cfg.device & 63;
  if (!ready[dev]) {
    HIP_TRY(c, hipFuncSetAttribute((const void *)heads_fused_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, HEADS_LDS_BYTES));
    HIP_TRY(c, hipFuncSetAttribute((const void *)heads_fused_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, HEADS_LDS_BYTES));
    ready[dev] = true;
  }
  const bool f16 = ti.f16;   // an FP16 engine: C8 fp16 activations in, the same fp32 arithmetic and fp32 bindings out
  const size_t plane = (size_t)ti.hp * ti.wp;
  HeadsArgs a;
  a.in_det = ring_ptr(c, ti) + (size_t)pb.in_c_off * plane / (f16 ? 2 : 1); a.det_in_per_image = ti.per_image;
  a.in_desc = ring_ptr(c, te) + (size_t)db.in_c_off * plane / (f16 ? 2 : 1); a.desc_in_per_image = te.per_image;
  a.in_hp = ti.hp; a.in_wp = ti.wp;
  a.wpack = c->d_heads_w;
  a.det = ring_ptr(c, tdet); a.det_per_image = tdet.per_image;
  a.desc_raw = c->heads_keep_raw ? ring_ptr(c, traw) : nullptr; a.raw_per_image = traw.per_image;
  a.desc = ring_ptr(c, tdesc);
  a.H = ti.H; a.W = ti.W; a.batch = batch;
  const double hbytes = batch * ((double)(pb.cin + db.cin) * ti.H * ti.W * (f16 ? 2 : 4) + (double)(pb.cout + (a.desc_raw ? 2 : 1) * db.cout) * ti.H * ti.W * 4) + (double)(pb.cout * pb.cin + db.cout * db.cin) * 4;
  ScopedStage st(c, stage_id(c, "heads"), (pb.flops_per_image + db.flops_per_image) * batch, hbytes, stream);
  // one workgroup per CU: each takes an equal share of the 16-pixel tiles of all images (heads.hip.h)
  const int ntiles = (batch * ti.H * ti.W + 15) / 16;
  const dim3 grid(std::min(c->num_cus, (ntiles + 1) / 2));
  if (f16) hipLaunchKernelGGL(heads_fused_kernel<true>, grid, dim3(HEADS_THREADS), HEADS_LDS_BYTES, stream, a);
  else hipLaunchKernelGGL(heads_fused_kernel<false>, grid, dim3(HEADS_THREADS), HEADS_LDS_BYTES, stream, a);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

int run_ops(spvo_ctx *c, int batch, size_t first, size_t last, hipStream_t stream) {
  for (size_t i = first; i < last && i < c->ops.size(); ++i) {
    if (c->heads_fused && i == c->head_start && last >= c->head_start + 3) {
      int rc = c->int8 ? launch_heads8(c, batch, stream) : launch_heads(c, batch, stream);
      if (rc) return rc;
      i += 2;
      continue;
    }
    int rc = launch_op(c, c->ops[i], 0, batch, stream);
    if (rc) return rc;
  }
  return SPVO_OK;
}

int run_network(spvo_ctx *c, int batch) {
  ScopedStage net(c, stage_id(c, "net"));
  // the synchronous entry points expose every tensor (spvo_debug_tensor), the un-normalised descriptor planes among them: the fused heads write
  // them as well here (tuning "heads_keep_raw" = 0: measurement scripts time the launch as a submission runs it, without them)
  c->heads_keep_raw = tuning("heads_keep_raw", 1) != 0;
  int rc = run_ops(c, batch, 0, c->ops.size(), c->stream);
  c->heads_keep_raw = false;
  if (rc) return rc;
  c->last_batch = batch;
  return SPVO_OK;
}

}  // namespace spvo_int

