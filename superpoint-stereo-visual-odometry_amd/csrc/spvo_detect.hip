// spvo_detect.hip -- preprocess (K0), heat map / NMS / descriptor sampling (K7-K11), the detector submissions (addStereoImagePair,
// feature_detection_neural_network.cpp:449-498), spvo_forward / spvo_debug_tensor, and the ORB detector of the classic front end.
#include "spvo_internal.hip.h"
#include "conv_mfma.hip.h"
#include "post.hip.h"
#include "conv_first_pre.hip.h"
#include "orb.hip.h"

namespace spvo_int {

// ---------------------------------------------------------------- resize tables
void linear_coeffs(int dst, int src, std::vector<int> &idx, std::vector<int> &a0, std::vector<int> &a1) {
  // OpenCV resize.cpp, INTER_LINEAR, 8-bit: float32 fractional part, 11-bit coefficients
  const double scale = (double)src / (double)dst;
  idx.resize(dst); a0.resize(dst); a1.resize(dst);
  for (int d = 0; d < dst; ++d) {
    float f = (float)((d + 0.5) * scale - 0.5);
    int s = (int)std::floor(f);
    f -= (float)s;
    if (s < 0) { s = 0; f = 0.f; }
    if (s >= src - 1) { s = src - 1; f = 0.f; }
    idx[d] = s;
    a0[d] = (int)std::nearbyint((1.f - f) * 2048.f);
    a1[d] = (int)std::nearbyint(f * 2048.f);
  }
}

struct CropGeom { int row_off, col_off, crop_rows, crop_cols; float scale; };

CropGeom crop_geometry(int rows, int cols, int net_h, int net_w) {
  // base.cpp:75-119, float32 arithmetic and int truncation as written there
  CropGeom g{0, 0, rows, cols, 1.f};
  const float real = (float)cols / (float)rows;
  const float expected = (float)net_w / (float)net_h;
  if (expected > real) {
    g.crop_rows = (int)((float)cols / expected);
    g.row_off = (rows - g.crop_rows) / 2;
  } else if (expected < real) {
    g.crop_cols = (int)((float)rows * expected);
    g.col_off = (cols - g.crop_cols) / 2;
  }
  g.scale = (float)net_w / (float)g.crop_cols;
  return g;
}

void fix_projection(double P[12], const CropGeom &g, int rows, int cols, int bug_compat) {
  if (bug_compat) {
    // base.cpp:95,111: at<float>(r, 2) on a CV_64F matrix = low 32 bits of P[r][1]
    float lo;
    if (g.crop_rows != rows) {
      std::memcpy(&lo, (char *)&P[4 + 1], 4);
      lo -= (float)g.row_off;
      std::memcpy((char *)&P[4 + 1], &lo, 4);
    } else if (g.crop_cols != cols) {
      std::memcpy(&lo, (char *)&P[0 + 1], 4);
      lo -= (float)g.col_off;
      std::memcpy((char *)&P[0 + 1], &lo, 4);
    }
  } else {
    if (g.crop_rows != rows) P[4 + 2] -= (double)(float)g.row_off;
    else if (g.crop_cols != cols) P[0 + 2] -= (double)(float)g.col_off;
  }
  for (int k = 0; k < 8; ++k) P[k] *= (double)g.scale;  // base.cpp:120
}

int ensure_tables(spvo_ctx *c, const CropGeom &g) {
  if (c->tab_rows == g.crop_rows && c->tab_cols == g.crop_cols) return SPVO_OK;
  std::vector<int> xi, xa0, xa1, yi, yb0, yb1;
  linear_coeffs(c->W, g.crop_cols, xi, xa0, xa1);
  linear_coeffs(c->H, g.crop_rows, yi, yb0, yb1);
  std::vector<int> all;
  all.insert(all.end(), xi.begin(), xi.end());
  all.insert(all.end(), xa0.begin(), xa0.end());
  all.insert(all.end(), xa1.begin(), xa1.end());
  all.insert(all.end(), yi.begin(), yi.end());
  all.insert(all.end(), yb0.begin(), yb0.end());
  all.insert(all.end(), yb1.begin(), yb1.end());
  HIP_TRY(c, hipMemcpyAsync(c->d_tab, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));  // `all` is a stack-lifetime buffer
  c->tab_rows = g.crop_rows;
  c->tab_cols = g.crop_cols;
  return SPVO_OK;
}

// one launch for `count` images (d_src0, d_src1) into slots slot0, slot0 + 1 of the resized-image buffer and the input tensor
// slot0: the first image of the network's input planes the pair goes to (and of c->d_resized, the stand-alone entries' resized images);
// resized_dst: a submission's own buffer for its two resized images instead (never offset)
int launch_preprocess(spvo_ctx *c, const uint8_t *d_src0, const uint8_t *d_src1, int count, int rows, int cols, size_t stride, const CropGeom &g, int slot0,
                      uint8_t *resized_dst = nullptr) {
  int rc = ensure_tables(c, g);
  if (rc) return rc;
  ResizeTab tab;
  tab.xi = c->d_tab; tab.xa0 = c->d_tab + c->W; tab.xa1 = c->d_tab + 2 * c->W;
  tab.yi = c->d_tab + 3 * c->W; tab.yb0 = tab.yi + c->H; tab.yb1 = tab.yi + 2 * c->H;
  const Tensor &tin = c->tensors[c->t_input];
  const int identity = (g.crop_rows == c->H && g.crop_cols == c->W) ? 1 : 0;
  dim3 grid((c->W + 63) / 64, (c->H + 3) / 4, count);
  hipLaunchKernelGGL(preprocess_kernel, grid, dim3(256), 0, c->stream, d_src0, d_src1, stride, rows, cols, g.row_off, g.col_off, g.crop_rows, g.crop_cols, tab, c->H, c->W,
                     resized_dst ? resized_dst : c->d_resized + (size_t)(slot0 & 1) * c->H * c->W, tin.d + (size_t)slot0 * tin.per_image, tin.per_image, tin.hp, tin.wp, identity);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

// the same, inside the first layer's launch (conv_first_pre.hip.h): the images of a group of `n` submissions (two each) -> resized u8 images,
// fp32 input planes and the first layer's output planes of the group's set.  The tables are the context's (ensure_tables at submission).
int launch_first_pre(spvo_ctx *c, PendingDetect *const *mem, int n, hipStream_t stream) {
  const Op &op = c->ops[0];
  const Tensor &ti = c->tensors[op.in], &to = c->tensors[op.out];
  const CropGeomS &g = mem[0]->g;
  FirstPreArgs a;
  for (int k = 0; k < 4; ++k) { a.src[k] = nullptr; a.out_u8[k] = nullptr; }
  const size_t hw = (size_t)c->H * c->W;
  for (int m = 0; m < n; ++m)
    for (int i = 0; i < 2; ++i) {
      a.src[2 * m + i] = mem[m]->src[i];
      // (without a buffer of its own a submission's resized images go where the stand-alone entry points keep theirs: the last submission's stay)
      a.out_u8[2 * m + i] = mem[m]->res_dst ? mem[m]->res_dst + i * hw : (m == n - 1 ? c->d_resized + i * hw : nullptr);
    }
  a.stride = mem[0]->stride;
  a.row_off = g.row_off; a.col_off = g.col_off; a.crop_rows = g.crop_rows; a.crop_cols = g.crop_cols;
  a.identity = (g.crop_rows == c->H && g.crop_cols == c->W) ? 1 : 0;
  a.tab.xi = c->d_tab; a.tab.xa0 = c->d_tab + c->W; a.tab.xa1 = c->d_tab + 2 * c->W;
  a.tab.yi = c->d_tab + 3 * c->W; a.tab.yb0 = a.tab.yi + c->H; a.tab.yb1 = a.tab.yi + 2 * c->H;
  a.in_plane = ti.d; a.in_per_image = ti.per_image;
  a.out = to.dr[c->cur_ring] ? to.dr[c->cur_ring] : to.d;
  a.w = op.d_w; a.bias = op.d_b;
  a.H = ti.H; a.W = ti.W; a.hp = ti.hp; a.wp = ti.wp; a.out_ctot = to.ch; a.out_coff = op.out_c_off; a.cout = op.cout;
  const int batch = 2 * n;
  ScopedStage st(c, op.stage, op.flops_per_image * batch, 0.0, stream);
  const dim3 grid((ti.W + 255) / 256, (ti.H + 3) / 4, batch);
  if (op.flags & FLAG_RELU) hipLaunchKernelGGL(conv_first4_pre_kernel<true>, grid, dim3(256), 0, stream, a);
  else hipLaunchKernelGGL(conv_first4_pre_kernel<false>, grid, dim3(256), 0, stream, a);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

// ---------------------------------------------------------------- NMS pipeline
constexpr int NMS_INNER = 4;
constexpr int NMS_GRID = 128;

// NMS counter blocks rotate through RING sets with the submissions: the last NMS kernel of one
// submission zeroes the block of the next one, so the steady state needs no memset.
// `set` < RING: a detector submission's buffers and counters; set == RING: the stand-alone entry
// points (buffers of set 0, counters of their own so that the submissions' blocks stay zeroed).
NmsPair nms_pair(spvo_ctx *c, int set) {
  NmsPair p;
  for (int i = 0; i < 2; ++i) {
    p.b[i] = c->nms_r[set % RING][i].b;
    p.b[i].counters = c->d_counters_all + (size_t)(set * 2 + i) * NMS_COUNTER_INTS;
  }
  return p;
}

// `n_launch` round launches + collect + rank + write for `nimg` images; the last kernel writes the counters
// into the set's pinned mirror.  Launch 0 of a batch never exits early.
int launch_nms_rounds(spvo_ctx *c, int nimg, const NmsPair &np, int set, int n_launch, int *zero_next, bool redo = false) {
  hipStream_t st = c->post;
  const float *heat = c->d_heat_r[set % RING];
  const int collect = redo ? 0 : 1;   // first batch: survivors are listed as they are decided; continuation: the list was cleared, re-collect everything
  // a submission's first batch: n_launch - 1 round launches, then the one-workgroup-per-image kernel that finishes the stragglers
  // (nms_finish_kernel); the host's continuation (redo): round launches only
  const int n_round = (!redo && n_launch > 1) ? n_launch - 1 : n_launch;
  for (int l = 0; l < n_round; ++l) {
    if (c->cfg.dist_thresh == 4)
      hipLaunchKernelGGL((nms_round_kernel<NMS_INNER, 4>), dim3(NMS_GRID, nimg), dim3(256), 0, st, heat, c->H, c->W, 4, np, l, c->cfg.border_remove, c->surv_cap, collect);
    else
      hipLaunchKernelGGL((nms_round_kernel<NMS_INNER, 0>), dim3(NMS_GRID, nimg), dim3(256), 0, st, heat, c->H, c->W, c->cfg.dist_thresh, np, l, c->cfg.border_remove, c->surv_cap, collect);
  }
  if (n_round < n_launch) {
    if (c->cfg.dist_thresh == 4)
      hipLaunchKernelGGL((nms_finish_kernel<4>), dim3(nimg), dim3(NMS_FIN_THREADS), 0, st, heat, c->H, c->W, 4, np, n_round, c->cfg.border_remove, c->surv_cap);
    else
      hipLaunchKernelGGL((nms_finish_kernel<0>), dim3(nimg), dim3(NMS_FIN_THREADS), 0, st, heat, c->H, c->W, c->cfg.dist_thresh, np, n_round, c->cfg.border_remove, c->surv_cap);
  }
  if (redo) hipLaunchKernelGGL(nms_collect_kernel, dim3(NMS_GRID, nimg), dim3(256), 0, st, heat, c->H, c->W, c->cfg.border_remove, c->surv_cap, np);
  hipLaunchKernelGGL(nms_rank_kernel, dim3(128, nimg), dim3(256), 0, st, c->surv_cap, np);
  hipLaunchKernelGGL(nms_write_kernel, dim3(32, nimg), dim3(256), 0, st, c->H, c->cfg.max_keypoints, c->surv_cap, np, zero_next, c->h_counters_r[set % RING]);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

// processOneHeatmap for images [0, nimg).  With a submission go three round launches of 4 in-kernel rounds (trained weights' heat
// maps settle in 2-3; a launch whose predecessor left nothing undecided exits at once) and the finishing kernel, one workgroup per
// image that iterates over whatever is still undecided until nothing is (post.hip.h).  Adversarial maps (e.g. a constant image: one
// decision chain across the whole picture) are continued by the host in batches of round launches -- every launch decides at least
// the best undecided candidate, so the loop terminates -- but that continuation queues behind everything on the tail stream and
// synchronises: in a pipelined loop it costs more than a trunk, which is what the finishing kernel is for.
// nms_settle runs after the caller's wait and reports whether it had to redo work (the caller then re-runs what depends on the keypoints).

// more rounds for the (rare) submissions whose first batch left candidates undecided
int nms_settle(spvo_ctx *c, int nimg, const NmsPair &np, int set, bool *redone) {
  int last = c->nms_first;
  *redone = false;
  const int *hc = c->h_counters_r[set % RING];
  for (;;) {
    bool pending = false;
    for (int i = 0; i < nimg; ++i) pending |= hc[i * NMS_COUNTER_INTS + 8 + last - 1] != 0;
    if (!pending) break;
    if (!*redone) c->stages[stage_id(c, "nms_redo")].calls += 1;   // host-driven continuations: counted even with profiling off (tests, diagnostics, bench.py)
    *redone = true;
    last = NMS_MAX_LAUNCH;
    for (int i = 0; i < nimg; ++i)   // keep n_cand, clear the rest of the block
      HIP_TRY(c, hipMemsetAsync(np.b[i].counters + 1, 0, (NMS_COUNTER_INTS - 1) * sizeof(int), c->post));
    int rc = launch_nms_rounds(c, nimg, np, set, last, nullptr, true);
    if (rc) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->post));
  }
  for (int i = 0; i < nimg; ++i)
    if (hc[i * NMS_COUNTER_INTS + 3]) return fail(c, SPVO_ERR_CAPACITY, "NMS survivor buffer overflow");
  return SPVO_OK;
}

// stand-alone entry (heat map already in d_heat): threshold + rounds, synchronous
int run_nms(spvo_ctx *c, int nimg) {
  const NmsPair np = nms_pair(c, RING);   // its own counter set: the submissions' blocks stay clean
  for (int i = 0; i < nimg; ++i) HIP_TRY(c, hipMemsetAsync(np.b[i].counters, 0, NMS_COUNTER_INTS * sizeof(int), c->stream));
  dim3 grid((c->W + 63) / 64, (c->H + 3) / 4, nimg);
  hipLaunchKernelGGL(nms_threshold_kernel, grid, dim3(256), 0, c->stream, c->d_heat, c->H, c->W, c->cfg.conf_thresh, np);
  int rc = launch_nms_rounds(c, nimg, np, RING, c->nms_first, nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  bool redone;
  return nms_settle(c, nimg, np, RING, &redone);
}

}  // namespace spvo_int

// ===========================================================================
extern "C" {

int spvo_preprocess(spvo_ctx *c, const uint8_t *img, int rows, int cols, size_t stride, double P[12], uint8_t *resized_u8) {
  if (!c || !img || !P || rows <= 0 || cols <= 0 || stride < (size_t)cols) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t bytes = (size_t)(rows - 1) * stride + cols;   // what is the caller's of a strided view: not the last row's padding
  if (bytes > c->img_cap) {
    for (int i = 0; i < 2; ++i) { if (c->d_img[i]) (void)hipFree(c->d_img[i]); c->d_img[i] = nullptr; }
    for (int i = 0; i < 2; ++i) { int rc = dev_alloc(c, &c->d_img[i], bytes, false); if (rc) return rc; }
    c->img_cap = bytes;
  }
  const CropGeom g = crop_geometry(rows, cols, c->H, c->W);
  HIP_TRY(c, hipMemcpyAsync(c->d_img[0], img, bytes, hipMemcpyHostToDevice, c->stream));
  int rc = launch_preprocess(c, c->d_img[0], c->d_img[0], 1, rows, cols, stride, g, 0);
  if (rc) return rc;
  fix_projection(P, g, rows, cols, c->cfg.bug_compat_p);
  if (resized_u8) HIP_TRY(c, hipMemcpyAsync(resized_u8, c->d_resized, (size_t)c->H * c->W, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SPVO_OK;
}

int spvo_forward(spvo_ctx *c, const float *input, int batch, float *det, float *desc_nhwc) {
  if (!c || !input) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  if (batch < 1 || batch > c->B) return fail(c, SPVO_ERR_INVALID, "batch %d out of range", batch);
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t hw = (size_t)c->H * c->W;
  const Tensor &tin = c->tensors[c->t_input];
  HIP_TRY(c, hipMemcpyAsync(c->d_dense_in, input, batch * hw * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(pad_input_kernel, dim3((c->W + 63) / 64, (c->H + 3) / 4, batch), dim3(256), 0, c->stream, c->d_dense_in, tin.d, c->H, c->W, tin.hp, tin.wp);
  int rc = run_network(c, batch);
  if (rc) return rc;
  const Tensor &td = c->tensors[c->t_det];
  if (det) {
    hipLaunchKernelGGL(unpad_kernel, dim3((td.W + 63) / 64, (td.H + 3) / 4, batch * 65), dim3(256), 0, c->stream, td.d, c->d_det_dense, 65, td.H, td.W, td.hp, td.wp);
    HIP_TRY(c, hipMemcpyAsync(det, c->d_det_dense, (size_t)batch * 65 * td.H * td.W * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  if (desc_nhwc) {
    const Tensor &ts = c->tensors[c->t_desc];
    HIP_TRY(c, hipMemcpyAsync(desc_nhwc, ts.d, (size_t)batch * ts.per_image * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SPVO_OK;
}

int spvo_debug_tensor(spvo_ctx *c, int tensor_id, int batch, float *out, size_t out_floats) {
  if (!c || !out) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  if (tensor_id < 0 || tensor_id >= (int)c->tensors.size() || batch < 1 || batch > c->B) return fail(c, SPVO_ERR_INVALID, "bad tensor id / batch");
  const Tensor &t = c->tensors[tensor_id];
  const size_t need = (size_t)batch * t.ch * t.H * t.W;
  if (out_floats < need) return fail(c, SPVO_ERR_CAPACITY, "need %zu floats", need);
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  if (t.nhwc) {
    HIP_TRY(c, hipMemcpy(out, t.d, need * sizeof(float), hipMemcpyDeviceToHost));
    return SPVO_OK;
  }
  float *tmp = nullptr;
  HIP_TRY(c, hipMalloc((void **)&tmp, need * sizeof(float)));
  if (t.i8) launch_unpad_c16(t, batch, tmp, c->stream);
  else if (t.s3) launch_unpad_s3(t, batch, tmp, c->stream);
  else if (t.f16) launch_unpad_c8(t, batch, tmp, c->stream);
  else hipLaunchKernelGGL(unpad_kernel, dim3((t.W + 63) / 64, (t.H + 3) / 4, batch * t.ch), dim3(256), 0, c->stream, t.d, tmp, t.ch, t.H, t.W, t.hp, t.wp);
  hipError_t e = hipMemcpyAsync(out, tmp, need * sizeof(float), hipMemcpyDeviceToHost, c->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
  (void)hipFree(tmp);
  if (e != hipSuccess) return fail(c, SPVO_ERR_DEVICE, "debug copy failed: %s", hipGetErrorString(e));
  return SPVO_OK;
}

int spvo_heatmap(spvo_ctx *c, const float *det, float *heat) {
  if (!c || !det || !heat) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipMemcpyAsync(c->d_det_dense, det, (size_t)65 * c->Hc * c->Wc * sizeof(float), hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(heatmap_kernel<false>, dim3((c->Wc + 63) / 64, (c->Hc + 3) / 4, 1), dim3(256), 0, c->stream, c->d_det_dense, c->d_heat, c->Hc, c->Wc, 0, 0);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(heat, c->d_heat, (size_t)c->H * c->W * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SPVO_OK;
}

int spvo_nms(spvo_ctx *c, const float *heat, int32_t *xy, int *n) {
  if (!c || !heat || !xy || !n) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  HIP_TRY(c, hipMemcpyAsync(c->d_heat, heat, (size_t)c->H * c->W * sizeof(float), hipMemcpyHostToDevice, c->stream));
  int rc = run_nms(c, 1);
  if (rc) return rc;
  *n = c->h_counters[2];
  HIP_TRY(c, hipMemcpy(xy, c->nms[0].b.out_xy, (size_t)(*n) * 2 * sizeof(int), hipMemcpyDeviceToHost));
  return SPVO_OK;
}

int spvo_sample_descriptors(spvo_ctx *c, const float *desc_nhwc, const int32_t *xy, int n, float *out) {
  if (!c || !desc_nhwc || (n > 0 && (!xy || !out))) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  if (n < 0 || n > c->cfg.max_keypoints) return fail(c, SPVO_ERR_CAPACITY, "n = %d exceeds max_keypoints", n);
  if (n == 0) return SPVO_OK;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const Tensor &ts = c->tensors[c->t_desc];
  HIP_TRY(c, hipMemcpyAsync(ts.d, desc_nhwc, ts.per_image * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_xy_tmp, xy, (size_t)n * 2 * sizeof(int), hipMemcpyHostToDevice, c->stream));
  SampleJobs sj;
  sj.j[0] = SampleJob{ts.d, c->d_xy_tmp, nullptr, n, c->d_desc_tmp, nullptr, nullptr, nullptr, nullptr, nullptr};
  sj.j[1] = sj.j[0];
  hipLaunchKernelGGL(sample_desc_kernel, dim3((n + 3) / 4, 1), dim3(256), 0, c->stream, sj, c->H, c->W, c->Hc, c->Wc);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->d_desc_tmp, (size_t)n * 256 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SPVO_OK;
}

// ring: the submission's own set (keypoint mirror); tring / img0: whose network outputs hold this pair's descriptor maps, and where
static int enqueue_sample(spvo_ctx *c, const int slots[2], const NmsPair &np, int ring, int tring, int img0) {
  const Tensor &ts = c->tensors[c->t_desc];
  const float *desc = (ts.dr[tring] ? ts.dr[tring] : ts.d) + (size_t)img0 * ts.per_image;
  ScopedStage ss(c, stage_id(c, "sample"));
  const int cap = c->cfg.max_keypoints;
  // the keypoints as floats go straight into the set's pinned mirror (8 bytes per keypoint), not through a staging buffer and a copy
  float *stage = c->h_xy_r[ring];
  SampleJobs sj;
  for (int i = 0; i < 2; ++i) {
    FeatureSlot &s = c->slots[slots[i]];
    // the keypoint count is read from the NMS counters on the device: no host round trip
    sj.j[i] = SampleJob{desc + (size_t)i * ts.per_image, np.b[i].out_xy, (const int *)(np.b[i].counters + 2), 0, s.d_desc, s.d_sqn,
                        stage + (size_t)i * cap * 2, s.d_xy, s.d_n, nullptr};
  }
  hipLaunchKernelGGL(sample_desc_kernel, dim3((cap + 3) / 4, 2), dim3(256), 0, c->post, sj, c->H, c->W, c->Hc, c->Wc);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

// descriptors of a host-image submission -> the set's pinned mirror, on `st` (a copy kernel: posted PCIe writes, no SDMA engine)
static int enqueue_desc_mirror(spvo_ctx *c, const int slots[2], int ring, hipStream_t st) {
  const int cap = c->cfg.max_keypoints;
  MirrorDescJob mj;
  for (int i = 0; i < 2; ++i) {
    const FeatureSlot &s = c->slots[slots[i]];
    mj.src[i] = s.d_desc; mj.n[i] = s.d_n; mj.dst[i] = c->h_desc_r[ring] + (size_t)i * cap * 256;
  }
  hipLaunchKernelGGL(mirror_desc_kernel, dim3(32, 2), dim3(256), 0, st, mj);
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

static int enqueue_prematch(spvo_ctx *c, int slot_l, int slot_r, int prev_l, int ring) {
  const int cap = c->cfg.max_keypoints;
  const int partner[2] = {slot_r, prev_l};
  MatchReq req[2];
  int nj = 0;
  for (int k = 0; k < 2; ++k) {
    MatchCache &mc = c->mcache[ring][k];
    mc.valid = false;
    if (partner[k] < 0) continue;
    FeatureSlot &a = c->slots[slot_l], &b = c->slots[partner[k]];
    req[nj] = MatchReq{a.d_desc, b.d_desc, cap, cap, a.d_n, b.d_n, a.d_sqn, b.d_sqn};
    MatchCache &dst = c->mcache[ring][nj];   // job nj's result lands in cache entry nj
    dst.slot_a = slot_l; dst.slot_b = partner[k];
    dst.selector = c->pm_selector; dst.cross = c->pm_cross; dst.ratio = c->pm_ratio;
    dst.valid = true;   // generations are stamped after the slots' counts are known
    ++nj;
  }
  if (nj == 0) return SPVO_OK;
  return enqueue_matches(c, req, nj, c->pm_selector, c->pm_cross, c->pm_ratio, c->h_match_out[ring]);
}

// Submission = network on `stream`, then the tail (heat map + NMS, sampling, the two matches and
// their writes to pinned memory) on `stream_t` behind an event.  Up to MAX_INFLIGHT submissions may
// be queued: the tail of one overlaps with the network of the next, whose kernels leave CUs idle at
// their ragged ends.  Every buffer a tail touches belongs to the submission's set (RING of them), so
// a later submission -- or the rare host-driven NMS redo of an earlier one -- never meets it.
//
// Trunk pairing (spvo_set_trunk_pairing, round 4): a submission's images are preprocessed at once, but its network may be HELD
// until the next submission arrives and then run for both pairs in ONE set of launches (four images per layer): every layer pays
// its launch, its first loads and its last stores once per two pairs, and 912 tiles spread better over 256 CUs than twice 456 --
// 678 instead of 738 us per pair for the forward pass (tools/fwd_batch.py).  A pair is only held while an earlier trunk is still
// queued or running (holding costs nothing then); spvo_detect_wait on a held pair launches it alone.  Results are independent of
// the grouping: the kernels were selected for two images at engine load and every tile is computed the same way wherever it runs
// (tests/test_gpu_host.py::test_prefetch_pipeline_is_transparent, depths 3 and 4).
static int ensure_host_sets(spvo_ctx *c, size_t image_bytes);
static int launch_group(spvo_ctx *c, bool from_submit = false);
static int launch_group_body(spvo_ctx *c);

// host_l / host_r != NULL: the images are in HOST memory -- they are staged through the set's pinned buffers and copied to the
// device on the network stream (d_l, d_r are then ignored); extras: see PendingDetect
static int detect_submit(spvo_ctx *c, const uint8_t *d_l, const uint8_t *d_r, int rows, int cols, size_t stride, int slot_l, int slot_r,
                         const uint8_t *host_l = nullptr, const uint8_t *host_r = nullptr, int extras = 0) {
  if ((int)c->pendq.size() >= MAX_INFLIGHT) return fail(c, SPVO_ERR_STATE, "%d detector submissions are already in flight", MAX_INFLIGHT);
  if (slot_l < 0 || slot_l >= N_SLOTS || slot_r < 0 || slot_r >= N_SLOTS || slot_l == slot_r) return fail(c, SPVO_ERR_INVALID, "bad feature slots %d, %d", slot_l, slot_r);
  for (const auto &q : c->pendq)
    if (q.slot_l == slot_l || q.slot_r == slot_l || q.slot_l == slot_r || q.slot_r == slot_r || q.prev_l == slot_l || q.prev_l == slot_r)
      return fail(c, SPVO_ERR_STATE, "feature slots %d, %d are used by a submission in flight", slot_l, slot_r);
  if (c->cfg.max_batch != 2) return fail(c, SPVO_ERR_INVALID, "max_batch == 1 detect path is not built yet; use max_batch = 2");
  const CropGeom g = crop_geometry(rows, cols, c->H, c->W);
  const uint8_t *srcs[2] = {d_l, d_r};
  const int slots[2] = {slot_l, slot_r};
  if (!host_l && (!d_l || !d_r)) return fail(c, SPVO_ERR_INVALID, "null image");
  // temporal partner = the left slot of the previous submission, if it survives this one
  int prev_l = c->last_slot_l;
  if (prev_l == slot_l || prev_l == slot_r || (prev_l >= 0 && !c->slots[prev_l].filled)) prev_l = -1;
  if (host_l || extras) {
    const int rc0 = ensure_host_sets(c, (size_t)rows * stride);
    if (rc0) return rc0;
  }
  const int ring = (int)(c->submit_count++ % RING);
  for (auto &mc : c->mcache[ring]) mc.valid = false;
  // The bulk results a host-image submission takes back (`extras`) are WRITTEN into the set's pinned mirrors by kernels (posted PCIe
  // writes), never copied behind events: a device-to-host copy waiting for its event occupies an SDMA queue, and the NEXT pair's image
  // upload queued on the same engine waits with it (round 3: bench.py's look-ahead leg 0.95 ms per frame, the same calls from
  // tools/sync_leg.py 0.81, depending on the process's copy history).
  if (host_l) {   // pageable -> pinned (host copy), pinned -> device (ONE DMA on the network stream): the caller's buffers are free on return
    const size_t bytes = (size_t)(rows - 1) * stride + cols;   // what is the caller's of a strided view: not the last row's padding
    std::memcpy(c->h_img_r[ring], host_l, bytes);
    std::memcpy(c->h_img_r[ring] + c->img_cap_r, host_r, bytes);
    // both images in one copy (the staging buffers of a set are contiguous, left then right): two copies were 17 + 16 us with 9 us
    // between them on the synchronous path's critical path (profiles/r04_sync_timeline.log)
    // ... on the SOLVER's stream when the pair's trunk will only queue behind another one: the copy then runs at once, beside the
    // running trunk, instead of between two trunks on the network stream, which only waits for its event (look-ahead leg 1313 ->
    // 1332-1343 frames/s; no device-to-host copy shares that engine any more: the bulk results leave through copy kernels).
    // Tuning "upload_side" = 0: always on the network stream.
    const bool up_side = tuning("upload_side", 1) != 0 && c->last_launch_ring >= 0 && hipEventQuery(c->ev_net[c->last_launch_ring]) == hipErrorNotReady;
    HIP_TRY(c, hipMemcpyAsync(c->d_img_r[ring], c->h_img_r[ring], c->img_cap_r + bytes, hipMemcpyHostToDevice, up_side ? c->stream2 : c->stream));
    if (up_side) {
      HIP_TRY(c, hipEventRecord(c->ev_up[ring], c->stream2));
      HIP_TRY(c, hipStreamWaitEvent(c->stream, c->ev_up[ring], 0));
    }
    srcs[0] = c->d_img_r[ring];
    srcs[1] = c->d_img_r[ring] + c->img_cap_r;
  }
  // ---- phase A, at once: the pair's images into the network's input planes (images 2 * position in the group, + 1).  Same stream as
  // the trunks: it runs behind the trunk that is using the input planes now.
  // (preprocess fused into the first layer, conv_first_pre.hip.h: one launch reads the images of the whole group with ONE crop geometry and the
  // context's tables -- a held submission of another geometry goes first, alone)
  if (c->pre_fused && c->held) {
    const PendingDetect &h = c->pendq.back();
    if (h.rows != rows || h.cols != cols || h.stride != stride || h.g.crop_rows != g.crop_rows || h.g.crop_cols != g.crop_cols || h.g.row_off != g.row_off || h.g.col_off != g.col_off) {
      int rc = launch_group(c, false);
      if (rc) return rc;
    }
  }
  const int pos = c->held;      // 0, or 1 when a held submission is waiting for a partner
  c->post = c->stream;
  if (c->pre_fused) {
    int rc = ensure_tables(c, g);
    if (rc) return rc;
  } else {
    ScopedStage sp(c, stage_id(c, "preprocess"));
    // the resized u8 images (what nn.cpp:154 pushes to images_dq) stay in device memory here: one byte per thread into pinned host
    // memory made this kernel 31 us instead of 8, in front of the whole network
    int rc = launch_preprocess(c, srcs[0], srcs[1], 2, rows, cols, stride, g, 2 * pos, (extras & 1) ? c->d_resized_r[ring] : nullptr);
    if (rc) return rc;
  }
  for (int i = 0; i < 2; ++i) c->slots[slots[i]].filled = true;
  c->last_slot_l = slot_l;
  PendingDetect pd;
  pd.g = CropGeomS{g.row_off, g.col_off, g.crop_rows, g.crop_cols, g.scale};
  pd.rows = rows; pd.cols = cols;
  pd.slot_l = slot_l; pd.slot_r = slot_r; pd.prev_l = prev_l; pd.ring = ring; pd.extras = extras; pd.early_res = (extras & 1) != 0;
  pd.launched = false; pd.img0 = 2 * pos; pd.tring = ring;
  if (c->pre_fused) { pd.pre_pending = true; pd.src[0] = srcs[0]; pd.src[1] = srcs[1]; pd.res_dst = (extras & 1) ? c->d_resized_r[ring] : nullptr; pd.stride = stride; }
  c->pendq.push_back(pd);
  c->held += 1;
  // ---- phases B and C now, unless the pair may wait for a partner: pairing is on, it is the first of its group, and an earlier
  // trunk is still queued or running (so nothing idles while it waits)
  const bool earlier_trunk_pending = c->last_launch_ring >= 0 && hipEventQuery(c->ev_net[c->last_launch_ring]) == hipErrorNotReady;
  // (tuning "pair_always" = 0: only while an earlier trunk is queued or running.  Holding the first pair of a group unconditionally costs
  // nothing where the GPU bounds the step -- the stream is never idle there -- and where the HOST does (FP16 / INT8 engines: ~35 launches of
  // ~4 us per pair against 0.1 ms of network) it halves the trunk's launches per pair: config 3 4050-4200 -> 4230-4520 frames/s, the
  // spread of its blocks 7-9 % -> 1.5-2 %)
  if (c->pair_trunks && c->held == 1 && (earlier_trunk_pending || c->pair_always)) return SPVO_OK;
  return launch_group(c, true);
}

extern "C++" {
namespace spvo_int {
int release_held_if_idle(spvo_ctx *c) {
  if (c->held != 1 || c->pair_always || (c->last_launch_ring >= 0 && hipEventQuery(c->ev_net[c->last_launch_ring]) == hipErrorNotReady)) return SPVO_OK;
  return launch_group(c);
}
}  // namespace spvo_int
}

// Launches the held pairs and keeps the queue consistent when that fails (a HIP error in a layer launch, the NMS rounds, an event
// record): the members' events were never recorded, so nothing may wait for them.  A member the caller has been told about (its submit
// returned SPVO_OK earlier) stays queued, marked `failed`: spvo_detect_wait / _collect takes it off the queue and returns the error, so the
// host's queue of accepted pairs and this one stay in step.  The member whose submit call is failing right now (`from_submit`) leaves
// the queue: that caller never counted it.
static int launch_group(spvo_ctx *c, bool from_submit) {
  const int n = c->held;
  if (n <= 0) return SPVO_OK;
  const int rc = launch_group_body(c);
  if (rc == SPVO_OK) return rc;
  const std::string why = c->error;
  seg_abort(c);   // (a launch segment the failing path left open: its kernels were never launched)
  c->held = 0;
  c->post = c->stream;
  c->cur_ring = 0;
  for (int m = 0; m < n && m < (int)c->pendq.size(); ++m) {
    PendingDetect &pd = c->pendq[c->pendq.size() - 1 - m];
    pd.failed = true;
    pd.launched = false;
    c->slots[pd.slot_l].filled = c->slots[pd.slot_r].filled = false;
    if (c->last_slot_l == pd.slot_l) c->last_slot_l = -1;
  }
  if (from_submit && !c->pendq.empty()) c->pendq.pop_back();
  c->error = why;
  return rc;
}

// phases B (the trunk of the held pairs: one or two, 2 or 4 images per launch) and C (each pair's tail)
static int launch_group_body(spvo_ctx *c) {
  const int n = c->held;
  if (n <= 0) return SPVO_OK;
  if (n > 2 || (int)c->pendq.size() < n) return fail(c, SPVO_ERR_STATE, "internal: %d held submissions, %zu in flight", n, c->pendq.size());
  c->held = 0;
  PendingDetect *mem[2] = {&c->pendq[c->pendq.size() - n], n == 2 ? &c->pendq[c->pendq.size() - 1] : nullptr};
  const int tring = mem[0]->ring;      // the set whose network outputs (det, desc, head inputs) hold all images of the group
  const int batch = 2 * n;
  const Tensor &td = c->tensors[c->t_det];
  // ---- everything below is enqueued without a host round trip
  c->cur_ring = tring;
  c->post = c->stream;
  // tuning "trunk_timing" = 1 (diagnostic): how long the network stream works per trunk launch and how long it stands idle between two,
  // from timing events at both ends of the trunk (printed every 200 launches)
  // -- the switch is read when the context is created (spvo_create), its timing events and sums belong to the context
  TrunkDiag &td_ = c->tdiag;
  const bool trunk_timing = c->trunk_timing != 0;
  constexpr int TT = TrunkDiag::TT;   // ring of timing events: deeper than the launches that can be in flight
  hipEvent_t *tt_b = td_.b, *tt_e = td_.e, *tt_tb = td_.tb, *tt_te = td_.te;   // trunk begin / end (network stream), tail begin / end (tail stream)
  double &tt_tail = td_.tail, &tt_lag = td_.lag;
  int *tt_np = td_.np;
  long &tt_n = td_.n;
  double &tt_busy = td_.busy, &tt_idle = td_.idle, &tt_pairs = td_.pairs;
  const int trace_lo = c->trunk_timing;   // > 1: one line per launch from that launch on (80 of them), with the host clock
  if (c->inject_launch_failure > 0 && ++c->launch_count == c->inject_launch_failure)   // tests of the error path (tuning "inject_launch_failure")
    return fail(c, SPVO_ERR_DEVICE, "injected launch failure (diagnostic switch)");
  if (trunk_timing) {
    const double tnow = diag_now_us();
    const bool found_idle = c->last_launch_ring >= 0 && hipEventQuery(c->ev_net[c->last_launch_ring]) == hipSuccess;
    if (found_idle) ++g_diag.late;   // the trunk before this one is done already: the stream is idle
    if (trace_lo > 1 && g_diag.launches + 1 >= trace_lo && g_diag.launches + 1 < trace_lo + 80)
      std::fprintf(stderr, "T %.0f launch %ld: %d pairs, stream %s, submissions so far %u, in flight %zu\n", tnow, g_diag.launches + 1, n, found_idle ? "IDLE" : "busy", c->submit_count, c->pendq.size());
    if (++g_diag.launches > 100 && g_diag.iv_printed < 16 && (found_idle || g_diag.iv_printed % 4 != 0)) {   // an idle launch and the three behind it
      std::fprintf(stderr, "[spvo]   launch %ld (%d pairs, stream %s): %.0f us since the previous launch, of which the host waited %.0f us for features, %.0f us for matches, %.0f us for the solver; %zu submissions in flight\n",
                   g_diag.launches, n, found_idle ? "IDLE" : "busy", tnow - g_diag.t_last_submit, g_diag.iv_tail, g_diag.iv_match, g_diag.iv_solve, c->pendq.size());
      ++g_diag.iv_printed;
    }
    g_diag.iv_tail = g_diag.iv_match = g_diag.iv_solve = 0;
    g_diag.depth_sum += (int)c->pendq.size();
    if (g_diag.t_last_submit > 0) g_diag.max_interval = std::max(g_diag.max_interval, tnow - g_diag.t_last_submit);
    g_diag.t_last_submit = tnow;
    if (tt_n == 0)
      for (int r = 0; r < TT; ++r) { (void)hipEventCreate(&tt_b[r]); (void)hipEventCreate(&tt_e[r]); (void)hipEventCreate(&tt_tb[r]); (void)hipEventCreate(&tt_te[r]); }
    if (tt_n >= TT) {   // the launches before those that may be in flight are complete: ring slots (n-8) and (n-9)
      const int r2 = (int)((tt_n - 8) % TT), r3 = (int)((tt_n - 9) % TT);
      float busy = 0, idle = 0;
      int &tt_late = td_.late;
      float &tt_max = td_.max_idle;
      if (hipEventElapsedTime(&busy, tt_b[r2], tt_e[r2]) == hipSuccess && hipEventElapsedTime(&idle, tt_e[r3], tt_b[r2]) == hipSuccess) {
        tt_busy += busy; tt_idle += idle;
        tt_late += idle > 0.05f ? 1 : 0;
        tt_max = std::max(tt_max, idle);
      }
      std::string &tt_pat = td_.pat;
      tt_pat += (char)('0' + tt_np[r2]);
      if (idle > 0.05f) tt_pat += idle > 0.3f ? 'I' : 'i';
      if (tt_n % 200 == 0) { std::fprintf(stderr, "[spvo]   pairs per launch (i / I: the stream stood idle > 50 / > 300 us in front of it): %s\n", tt_pat.c_str()); tt_pat.clear(); }
      if (trace_lo > 1) {   // device-side times of launch (tt_n - 8), relative to the first trace line's moment
        hipEvent_t &base = td_.base;
        double &base_host = td_.base_host;
        if (!base && g_diag.launches >= trace_lo - 8) { (void)hipEventCreate(&base); (void)hipEventRecord(base, c->stream_t); (void)hipEventSynchronize(base); base_host = diag_now_us(); }
        float b0 = 0, e0 = 0, tb0 = 0, te0 = 0;
        if (base && g_diag.launches - 8 >= trace_lo && g_diag.launches - 8 < trace_lo + 80 && hipEventElapsedTime(&b0, base, tt_b[r2]) == hipSuccess &&
            hipEventElapsedTime(&e0, base, tt_e[r2]) == hipSuccess && hipEventElapsedTime(&tb0, base, tt_tb[r2]) == hipSuccess && hipEventElapsedTime(&te0, base, tt_te[r2]) == hipSuccess)
          std::fprintf(stderr, "G launch %ld (%d pairs): trunk %.0f .. %.0f, tail %.0f .. %.0f (host clock)\n", g_diag.launches - 8, tt_np[r2], base_host + b0 * 1e3, base_host + e0 * 1e3,
                       base_host + tb0 * 1e3, base_host + te0 * 1e3);
      }
      float tail = 0, lag = 0;
      if (hipEventElapsedTime(&tail, tt_tb[r2], tt_te[r2]) == hipSuccess && hipEventElapsedTime(&lag, tt_e[r2], tt_te[r2]) == hipSuccess) { tt_tail += tail; tt_lag += lag; }
      if (tt_n % 200 == 0) {
        std::fprintf(stderr, "[spvo] trunk timing over 200 launches (%.0f pairs): network stream busy %.1f us, idle %.1f us per launch (%d gaps above 50 us, longest %.0f us)\n",
                     tt_pairs, tt_busy * 1e3 / 200, tt_idle * 1e3 / 200, tt_late, tt_max * 1e3);
        std::fprintf(stderr, "[spvo]   NMS continuations driven by the host so far: %lld\n", c->stages[stage_id(c, "nms_redo")].calls);
        std::fprintf(stderr, "[spvo]   tail stream: %.1f us per launch from its first kernel to its last, which ends %.1f us behind the trunk\n", tt_tail * 1e3 / 200, tt_lag * 1e3 / 200);
        tt_tail = tt_lag = 0;
        std::fprintf(stderr, "[spvo]   host: longest interval between launches %.0f us, longest wait for a tail %.0f us, for a solve %.0f us, matches not served from the cache %d; "
                             "launches that found the network stream idle %d, mean submissions in flight at launch %.2f\n",
                     g_diag.max_interval, g_diag.max_tail_wait, g_diag.max_solve_wait, g_diag.match_miss, g_diag.late, g_diag.depth_sum / 200.0);
        g_diag.max_interval = g_diag.max_tail_wait = g_diag.max_solve_wait = 0; g_diag.match_miss = 0; g_diag.late = 0; g_diag.depth_sum = 0;
        tt_busy = tt_idle = tt_pairs = 0; tt_late = 0; tt_max = 0;
      }
    }
    tt_pairs += n;
    tt_np[tt_n % TT] = n;
    (void)hipEventRecord(tt_b[tt_n % TT], c->stream);
  }
  hipEvent_t det_e0 = nullptr;
  const bool prof_detect = c->prof && (c->prof_only < 0 || c->prof_only == stage_id(c, "detect"));
  if (prof_detect) { det_e0 = get_event(c); (void)hipEventRecord(det_e0, c->stream); }
  int rc;
  const int hon = tuning("heads_on_net", -1);
  const bool heads_on_net = hon < 0 ? c->heads_on_net : hon != 0;
  bool any_res0 = false;
  for (int m = 0; m < n; ++m) any_res0 = any_res0 || mem[m]->early_res;
  const long long gen = ((long long)c->plan_gen << 40) ^ ((long long)c->alloc_gen << 20) ^ (long long)tuning_generation();   // engine, buffers, switches
  {
    ScopedStage net(c, stage_id(c, "net"));
    // launch segment T: the group's trunk (and its heads where they stay on the network stream) -- not for a group whose first layer also
    // preprocesses (its arguments are the caller's image pointers) or whose resized images leave through the tail stream in between
    const bool seg_t = !mem[0]->pre_pending && !any_res0 &&
                       seg_begin(c, &c->seg_T[tring][n - 1], seg_key({1, tring, batch, gen, heads_on_net ? 1 : 0, (long long)c->head_start}), c->stream);
    (void)seg_t;
    rc = mem[0]->pre_pending ? launch_first_pre(c, mem, n, c->stream) : run_ops(c, batch, 0, std::min<size_t>(1, c->head_start), c->stream);
    for (int m = 0; m < n; ++m) mem[m]->pre_pending = false;
    // The resized images leave for their sets' pinned mirrors UNDER the network: a copy kernel (16 bytes per lane, no SDMA engine involved)
    // on the TAIL stream behind the FIRST layer -- beside it (conv1a is bound by its 217 MB of stores) the copy made that layer 54 us
    // instead of 37 -- i.e. beside conv1b, which leaves 12 CUs free and does not notice
    bool any_res = false;
    for (int m = 0; m < n; ++m) any_res = any_res || mem[m]->early_res;
    if (!rc && any_res) {
      HIP_TRY(c, hipEventRecord(c->ev_pre[tring], c->stream));
      HIP_TRY(c, hipStreamWaitEvent(c->stream_t, c->ev_pre[tring], 0));
      const size_t n16 = ((size_t)2 * c->H * c->W + 15) / 16;   // (the buffers are allocated in multiples of 256 bytes)
      for (int m = 0; m < n; ++m) {
        if (!mem[m]->early_res) continue;
        hipLaunchKernelGGL(mirror_copy_kernel, dim3((unsigned)std::min<size_t>((n16 + 255) / 256, 64)), dim3(256), 0, c->stream_t,
                           reinterpret_cast<const uint4 *>(c->d_resized_r[mem[m]->ring]), reinterpret_cast<uint4 *>(c->h_resized_r[mem[m]->ring]), n16);
        HIP_TRY(c, hipGetLastError());
        HIP_TRY(c, hipEventRecord(c->ev_res[mem[m]->ring], c->stream_t));
      }
    }
    if (!rc) rc = run_ops(c, batch, std::min<size_t>(1, c->head_start), c->head_start, c->stream);
  }
  if (rc) { (void)seg_end(c); c->cur_ring = 0; return rc; }
  c->last_batch = batch;
  // The heads (2.2 GFLOP: the one heavy piece behind the trunk): on the network stream, in front of the next pair's trunk, when the
  // trunk is made of persistent one-workgroup-per-CU launches (VGG fp32: beside the next pair's conv1b they would have 12 CUs, the
  // tail would finish late and the network stream idle 50-70 us per pair: 1257-1265 against 1308 frames/s); on the tail stream
  // otherwise, where the overlap pays (sp_squeeze fp32 1286 against 1248 frames/s, INT8 sp_mbv1 2360 against 2237).  The plan
  // loader decides (spvo_ctx::heads_on_net); tuning "heads_on_net" = 0 / 1 overrides (measurements).
  if (heads_on_net) {
    rc = run_ops(c, batch, c->head_start, c->ops.size(), c->stream);
    if (rc) { (void)seg_end(c); c->cur_ring = 0; return rc; }
  }
  if ((rc = seg_end(c))) { c->cur_ring = 0; return rc; }   // segment T goes out here: one graph launch, or its kernels one by one
  if (trunk_timing) (void)hipEventRecord(tt_e[tt_n % TT], c->stream);
  for (int m = 0; m < n; ++m) HIP_TRY(c, hipEventRecord(c->ev_net[mem[m]->ring], c->stream));
  c->last_launch_ring = mem[n - 1]->ring;
  // Tail streams: one, or two that consecutive submissions alternate between by the parity of their set (tuning "tail_streams" = 2: for
  // engines whose network is shorter than a pair's chain of ~10 dependent tail kernels -- FP16 / INT8 -- that chain, one pair after the
  // other on one stream, is the frame loop's cycle time: config 3 5670 -> 6490 frames/s).  It pays only when the second stream gets a
  // hardware queue of its own: the runtime deals a process's streams onto GPU_MAX_HW_QUEUES (default 4) queues, and a fifth stream that
  // shares one with the network or the first tail stream makes things worse (5300).  Opt-in, with GPU_MAX_HW_QUEUES=8 in the process's
  // environment; FP32 engines lose with it (more small kernels beside the trunk's persistent launches).  What crosses
  // from one submission's tail to the next one's: the features the temporal match reads (ev_feat of the submission before) and the clean
  // NMS counter block, which a tail hands to the next submission ON ITS OWN STREAM (two sets ahead with two streams).
  hipStream_t tstreams[2] = {c->stream_t, c->tail_streams == 2 ? c->stream_tb : c->stream_t};
  for (int m = 0; m < n; ++m) mem[m]->ts = c->tail_streams == 2 ? (mem[m]->ring & 1) : 0;
  hipStream_t ts0 = tstreams[mem[0]->ts];
  HIP_TRY(c, hipStreamWaitEvent(ts0, c->ev_net[tring], 0));
  if (n == 2 && mem[1]->ts != mem[0]->ts) HIP_TRY(c, hipStreamWaitEvent(tstreams[mem[1]->ts], c->ev_net[tring], 0));
  if (trunk_timing) (void)hipEventRecord(tt_tb[tt_n % TT], ts0);
  c->post = ts0;
  if (!heads_on_net) {
    seg_begin(c, &c->seg_H[tring][n - 1], seg_key({2, tring, batch, gen, (long long)c->head_start}), ts0);   // launch segment H: the heads
    rc = run_ops(c, batch, c->head_start, c->ops.size(), ts0);   // heads: on the (first pair's) tail stream, reading this group's ring buffers
    { const int rce = seg_end(c); if (!rc) rc = rce; }
    if (!rc && n == 2 && mem[1]->ts != mem[0]->ts) {
      HIP_TRY(c, hipEventRecord(c->ev_heads[tring], ts0));
      HIP_TRY(c, hipStreamWaitEvent(tstreams[mem[1]->ts], c->ev_heads[tring], 0));
    }
  }
  c->cur_ring = 0;
  if (rc) { c->post = c->stream; return rc; }
  // ---- phase C: each pair's tail, in submission order (the second pair's temporal match reads the first pair's features)
  for (int m = 0; m < n && !rc; ++m) {
    PendingDetect &pd = *mem[m];
    const int ring = pd.ring, slots[2] = {pd.slot_l, pd.slot_r};
    pd.tring = tring;
    const NmsPair np = nms_pair(c, ring);
    hipStream_t tsm = tstreams[pd.ts];
    c->post = tsm;
    c->ms_set = pd.ts;
    // launch segment A: heat map, NMS rounds + finish, rank, write, sampling -- eight dependent kernels up to ev_feat
    seg_begin(c, &c->seg_A[ring], seg_key({3, ring, tring, pd.img0, pd.slot_l, pd.slot_r, gen, c->tail_streams, c->nms_first, c->cfg.max_keypoints}), tsm);
    {
      // heat map + threshold + candidate list in one kernel; the counter block of this set was
      // zeroed by the previous submission's last NMS kernel (or at allocation)
      ScopedStage sh(c, stage_id(c, "heatmap"));
      hipLaunchKernelGGL(heatmap_nms_kernel, dim3((c->Wc + 63) / 64, (c->Hc + 3) / 4, 2), dim3(256), 0, c->post, td.dr[tring] + (size_t)pd.img0 * td.per_image, c->d_heat_r[ring],
                         c->Hc, c->Wc, td.hp, td.wp, c->cfg.conf_thresh, np);
      HIP_TRY(c, hipGetLastError());
    }
    {
      ScopedStage sn(c, stage_id(c, "nms"));
      rc = launch_nms_rounds(c, 2, np, ring, c->nms_first, c->d_counters_all + (size_t)(((ring + c->tail_streams) % RING) * 2) * NMS_COUNTER_INTS);
    }
    if (!rc) rc = enqueue_sample(c, slots, np, ring, tring, pd.img0);
    { const int rce = seg_end(c); if (!rc) rc = rce; }
    // Keypoints, counts and descriptors are final here: spvo_detect_wait / _collect waits for THIS point (ev_feat); the matches enqueued
    // behind it are waited for where they are asked for (spvo_match_slots, ev_tail).
    if (!rc) rc = hipEventRecord(c->ev_feat[ring], tsm) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "hipEventRecord failed");
    if (!rc && c->prematch) {
      // (two tail streams: the temporal partner's features come from the submission before, on the other stream)
      if (c->tail_streams == 2 && pd.prev_l >= 0) HIP_TRY(c, hipStreamWaitEvent(tsm, c->ev_feat[(ring + RING - 1) % RING], 0));
      // launch segment B: the pair's two matches (distance GEMM + merge; the fp8 shortlist's conversions and re-rank)
      unsigned ratio_bits;
      std::memcpy(&ratio_bits, &c->pm_ratio, 4);
      seg_begin(c, &c->seg_B[ring], seg_key({4, ring, pd.slot_l, pd.slot_r, pd.prev_l, gen, c->pm_selector, c->pm_cross, (long long)ratio_bits, c->match_fp8 ? 1 : 0, pd.ts}), tsm);
      rc = enqueue_prematch(c, pd.slot_l, pd.slot_r, pd.prev_l, ring);
      { const int rce = seg_end(c); if (!rc) rc = rce; }
    }
    if (!rc && prof_detect) {   // "detect" spans both streams: first kernel on `stream` .. last kernel on the tail stream
      hipEvent_t e1 = get_event(c);
      (void)hipEventRecord(e1, tsm);
      c->pending.push_back({stage_id(c, "detect"), det_e0, e1});
      if (m + 1 < n) { det_e0 = get_event(c); (void)hipEventRecord(det_e0, tstreams[mem[m + 1]->ts]); }   // (an event is timed once)
    }
    if (!rc) rc = (hipEventRecord(c->ev_tail[ring], tsm) == hipSuccess) ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "hipEventRecord failed");
    // The descriptors a host-image submission takes back (extras bit 1: 2 x 1 MB) leave for the set's pinned mirror BEHIND the
    // matches, on the tail stream (a stream of their own had them share a hardware queue with the network stream in processes that had
    // created and destroyed contexts before -- the runtime deals streams onto four queues -- and bench.py's look-ahead leg fell from
    // 1230 to 1070 frames/s while the same calls from tools/sync_leg.py ran at 1260): written by the sampling kernel itself they
    // made it 40 us instead of 4 in front of ev_feat; beside the matches the copy kernel (44 us of PCIe writes) made the distance GEMM
    // 55 us instead of 20.  ev_copy = they have arrived (spvo_detect_mirrors_wait; spvo_detect_collect waits for it itself).
    if (!rc && (pd.extras & 2)) {
      rc = enqueue_desc_mirror(c, slots, ring, tsm);
      if (!rc) HIP_TRY(c, hipEventRecord(c->ev_copy[ring], tsm));
    }
    pd.launched = true;
  }
  if (trunk_timing) { (void)hipEventRecord(tt_te[tt_n % TT], tstreams[mem[n - 1]->ts]); ++tt_n; }
  c->post = c->stream;
  c->ms_set = 0;
  return rc;
}

// completes the OLDEST submission
static int detect_wait(spvo_ctx *c, double P_l[12], double P_r[12], spvo_features *out_l, spvo_features *out_r, uint8_t *resized_l, uint8_t *resized_r,
                       spvo_detect_mirrors *mirrors = nullptr) {
  if (c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "no detector submission in flight");
  if (c->pendq.front().failed) {   // its group's launch failed after the submission had been accepted (launch_group): nothing of it is in flight
    c->pendq.pop_front();
    const std::string why = c->error;
    return fail(c, SPVO_ERR_STATE, "the submission's network launch had failed: %s", why.c_str());
  }
  if (!c->pendq.front().launched) {   // a pair that was held for a partner (trunk pairing) and is asked for first: it runs alone
    if (c->held == 0) {   // unlaunched with nothing held: no event of it was ever recorded -- never wait for one
      c->pendq.pop_front();
      return fail(c, SPVO_ERR_STATE, "internal: an unlaunched submission with no group held");
    }
    const int rcl = launch_group(c);
    if (rcl) {            // (launch_group marked it failed; it is the front of the queue)
      if (!c->pendq.empty() && c->pendq.front().failed) c->pendq.pop_front();
      return rcl;
    }
  }
  const PendingDetect pd = c->pendq.front();
  const int slots[2] = {pd.slot_l, pd.slot_r};
  const int cap = c->cfg.max_keypoints;
  uint8_t *res[2] = {resized_l, resized_r};
  spvo_features *outs[2] = {out_l, out_r};
  const bool want_res = resized_l || resized_r, want_desc = (out_l && out_l->desc) || (out_r && out_r->desc);
  // what the submission staged into its own pinned mirrors is simply read there; anything else has to be copied now, from buffers a
  // younger submission may already be rewriting -- refused BEFORE the submission is taken off the queue
  const bool extras = (want_res && !(pd.extras & 1)) || (want_desc && !(pd.extras & 2));
  if (extras && c->pendq.size() > 1) return fail(c, SPVO_ERR_STATE, "resized images / host descriptors can only be fetched with one submission in flight (or request them at spvo_detect_submit)");
  c->pendq.pop_front();
  hipStream_t tsw = (pd.ts && c->stream_tb) ? c->stream_tb : c->stream_t;   // the submission's own tail stream
  c->post = tsw;
  c->ms_set = pd.ts;
  auto copy_extras = [&]() -> int {
    if (!(pd.extras & 1))
      for (int i = 0; i < 2; ++i)
        if (res[i]) HIP_TRY(c, hipMemcpyAsync(res[i], c->d_resized + (size_t)i * c->H * c->W, (size_t)c->H * c->W, hipMemcpyDeviceToHost, c->post));
    // descriptors: copy the full slot (1000 x 256 floats); rows >= n are stale
    if (!(pd.extras & 2))
      for (int i = 0; i < 2; ++i)
        if (outs[i] && outs[i]->desc) HIP_TRY(c, hipMemcpyAsync(outs[i]->desc, c->slots[slots[i]].d_desc, (size_t)cap * 256 * sizeof(float), hipMemcpyDeviceToHost, c->post));
    return SPVO_OK;
  };
  int rc = SPVO_OK;
  if (extras) {
    if ((rc = copy_extras())) { c->post = c->stream; c->ms_set = 0; return rc; }
    rc = hipStreamSynchronize(tsw) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "stream synchronisation failed");
  } else {
    // only this submission's tail: a younger one may be queued behind it on both streams
    const double tw0 = diag_now_us();
    rc = wait_event(c->ev_feat[pd.ring]) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "event synchronisation failed");
    g_diag.max_tail_wait = std::max(g_diag.max_tail_wait, diag_now_us() - tw0);
    g_diag.iv_tail += diag_now_us() - tw0;
  }
  if (!rc && pd.early_res) rc = wait_event(c->ev_res[pd.ring]) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "event synchronisation failed");   // the resized images: they left under the network (copy kernel on the tail stream)
  bool redone = false;
  const NmsPair np = nms_pair(c, pd.ring);
  if (!rc) rc = nms_settle(c, 2, np, pd.ring, &redone);
  if (!rc && (redone || pd.rematch)) {   // rare: keypoints changed after the first batch -> redo what depends on them
    if (pd.rematch) c->stages[stage_id(c, "rematch")].calls += 1;
    if (redone && (pd.extras & 2)) (void)wait_event(c->ev_copy[pd.ring]);   // the mirror of the superseded descriptors has landed: the new one goes on top
    if (redone) rc = enqueue_sample(c, slots, np, pd.ring, pd.tring, pd.img0);
    if (!rc && redone && (pd.extras & 2)) rc = enqueue_desc_mirror(c, slots, pd.ring, c->post);
    if (!rc && c->prematch) rc = enqueue_prematch(c, pd.slot_l, pd.slot_r, pd.prev_l, pd.ring);
    if (!rc && extras) rc = copy_extras();
    if (!rc) rc = hipStreamSynchronize(tsw) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "stream synchronisation failed");
    if (redone)
      for (auto &q : c->pendq)
        if (q.prev_l == pd.slot_l) q.rematch = true;   // it matched against keypoints that have just been replaced
  }
  c->post = c->stream;
  c->ms_set = 0;
  if (rc) return rc;
  const int *hc = c->h_counters_r[pd.ring];
  for (int i = 0; i < 2; ++i) {
    FeatureSlot &s = c->slots[slots[i]];
    s.n = hc[i * NMS_COUNTER_INTS + 2];
    s.gen += 1;
    if (outs[i]) {
      outs[i]->n = s.n;
      if (outs[i]->xy && s.n > 0) std::memcpy(outs[i]->xy, c->h_xy_r[pd.ring] + (size_t)i * cap * 2, (size_t)s.n * 2 * sizeof(float));
      if (outs[i]->desc && (pd.extras & 2) && s.n > 0) {
        (void)wait_event(c->ev_copy[pd.ring]);   // the descriptors' mirror (copy kernel behind the matches)
        std::memcpy(outs[i]->desc, c->h_desc_r[pd.ring] + (size_t)i * cap * 256, (size_t)s.n * 256 * sizeof(float));
      }
    }
    if (res[i] && (pd.extras & 1)) std::memcpy(res[i], c->h_resized_r[pd.ring] + (size_t)i * c->H * c->W, (size_t)c->H * c->W);
  }
  for (auto &mc : c->mcache[pd.ring])
    if (mc.valid) { mc.gen_a = c->slots[mc.slot_a].gen; mc.gen_b = c->slots[mc.slot_b].gen; }
  if (mirrors)
    for (int i = 0; i < 2; ++i) {
      mirrors->n[i] = c->slots[slots[i]].n;
      mirrors->xy[i] = c->h_xy_r[pd.ring] + (size_t)i * cap * 2;
      mirrors->desc[i] = (pd.extras & 2) ? c->h_desc_r[pd.ring] + (size_t)i * cap * 256 : nullptr;
      mirrors->resized[i] = (pd.extras & 1) ? c->h_resized_r[pd.ring] + (size_t)i * c->H * c->W : nullptr;
      mirrors->token = pd.ring;
    }
  const CropGeom g{pd.g.row_off, pd.g.col_off, pd.g.crop_rows, pd.g.crop_cols, pd.g.scale};
  if (P_l) fix_projection(P_l, g, pd.rows, pd.cols, c->cfg.bug_compat_p);
  if (P_r) fix_projection(P_r, g, pd.rows, pd.cols, c->cfg.bug_compat_p);
  return SPVO_OK;
}

// buffers of the host-image submissions: allocated on first use, grown when a larger image arrives (never while submissions are in flight)
static int ensure_host_sets(spvo_ctx *c, size_t image_bytes) {
  const size_t hw2 = (size_t)2 * c->H * c->W, desc = (size_t)2 * c->cfg.max_keypoints * 256;
  if (!c->host_sets_ready) {   // (a flag of its own: a failure half-way must not look like "allocated" to the next call)
    ++c->alloc_gen;   // (pinned buffers recorded launch segments may point to: a new generation of segment keys)
    for (int r = 0; r < RING; ++r) {
      if (!c->d_resized_r[r]) { int rc = dev_alloc(c, &c->d_resized_r[r], hw2, false); if (rc) return rc; }
      if (!c->h_resized_r[r]) HIP_TRY(c, hipHostMalloc((void **)&c->h_resized_r[r], hw2));
      if (!c->h_desc_r[r]) HIP_TRY(c, hipHostMalloc((void **)&c->h_desc_r[r], desc * sizeof(float)));
    }
    c->host_sets_ready = true;
  }
  if (image_bytes > c->img_cap_r) {
    ++c->alloc_gen;
    if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "the image size grew while submissions are in flight");
    HIP_TRY(c, hipDeviceSynchronize());
    for (int r = 0; r < RING; ++r) {
      if (c->d_img_r[r]) (void)hipFree(c->d_img_r[r]);
      if (c->h_img_r[r]) (void)hipHostFree(c->h_img_r[r]);
      c->d_img_r[r] = c->h_img_r[r] = nullptr;
    }
    c->img_cap_r = 0;
    for (int r = 0; r < RING; ++r) {
      int rc = dev_alloc(c, &c->d_img_r[r], 2 * image_bytes, false);
      if (rc) return rc;
      HIP_TRY(c, hipHostMalloc((void **)&c->h_img_r[r], 2 * image_bytes));
    }
    c->img_cap_r = image_bytes;
  }
  return SPVO_OK;
}

static int detect_common(spvo_ctx *c, const uint8_t *d_l, const uint8_t *d_r, int rows, int cols, size_t stride, double P_l[12], double P_r[12],
                         int slot_l, int slot_r, spvo_features *out_l, spvo_features *out_r, uint8_t *resized_l, uint8_t *resized_r) {
  int rc = detect_submit(c, d_l, d_r, rows, cols, stride, slot_l, slot_r);
  if (rc) return rc;
  return detect_wait(c, P_l, P_r, out_l, out_r, resized_l, resized_r);
}

int spvo_detect(spvo_ctx *c, const uint8_t *img_l, const uint8_t *img_r, int rows, int cols, size_t stride, double P_l[12], double P_r[12],
                int slot_l, int slot_r, spvo_features *out_l, spvo_features *out_r, uint8_t *resized_l, uint8_t *resized_r) {
  if (!c || !img_l || !img_r || !P_l || !P_r || rows <= 0 || cols <= 0 || stride < (size_t)cols) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t bytes = (size_t)(rows - 1) * stride + cols;   // what is the caller's of a strided view: not the last row's padding
  if (bytes > c->img_cap) {
    for (int i = 0; i < 2; ++i) { if (c->d_img[i]) (void)hipFree(c->d_img[i]); c->d_img[i] = nullptr; }
    for (int i = 0; i < 2; ++i) { int rc = dev_alloc(c, &c->d_img[i], bytes, false); if (rc) return rc; }
    c->img_cap = bytes;
  }
  HIP_TRY(c, hipMemcpyAsync(c->d_img[0], img_l, bytes, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_img[1], img_r, bytes, hipMemcpyHostToDevice, c->stream));
  return detect_common(c, c->d_img[0], c->d_img[1], rows, cols, stride, P_l, P_r, slot_l, slot_r, out_l, out_r, resized_l, resized_r);
}

int spvo_detect_dev(spvo_ctx *c, const void *d_img_l, const void *d_img_r, int rows, int cols, size_t stride, double P_l[12], double P_r[12],
                    int slot_l, int slot_r, spvo_features *out_l, spvo_features *out_r) {
  if (!c || !d_img_l || !d_img_r || !P_l || !P_r || rows <= 0 || cols <= 0 || stride < (size_t)cols) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  return detect_common(c, (const uint8_t *)d_img_l, (const uint8_t *)d_img_r, rows, cols, stride, P_l, P_r, slot_l, slot_r, out_l, out_r, nullptr, nullptr);
}

int spvo_detect_dev_submit(spvo_ctx *c, const void *d_img_l, const void *d_img_r, int rows, int cols, size_t stride, int slot_l, int slot_r) {
  if (!c || !d_img_l || !d_img_r || rows <= 0 || cols <= 0 || stride < (size_t)cols) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  return detect_submit(c, (const uint8_t *)d_img_l, (const uint8_t *)d_img_r, rows, cols, stride, slot_l, slot_r);
}

int spvo_detect_wait(spvo_ctx *c, double P_l[12], double P_r[12], spvo_features *out_l, spvo_features *out_r) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  return detect_wait(c, P_l, P_r, out_l, out_r, nullptr, nullptr);
}

int spvo_detect_submit(spvo_ctx *c, const uint8_t *img_l, const uint8_t *img_r, int rows, int cols, size_t stride, int slot_l, int slot_r, int extras) {
  if (!c || !img_l || !img_r || rows <= 0 || cols <= 0 || stride < (size_t)cols || (extras & ~3)) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  return detect_submit(c, nullptr, nullptr, rows, cols, stride, slot_l, slot_r, img_l, img_r, extras);
}

int spvo_detect_collect(spvo_ctx *c, double P_l[12], double P_r[12], spvo_features *out_l, spvo_features *out_r, uint8_t *resized_l, uint8_t *resized_r) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  return detect_wait(c, P_l, P_r, out_l, out_r, resized_l, resized_r);
}

int spvo_set_trunk_pairing(spvo_ctx *c, int on) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  c->pair_trunks = on != 0;
  if (!c->pair_trunks && c->held) return launch_group(c);   // nobody is left waiting for a partner that will not be paired
  return SPVO_OK;
}

int spvo_detect_mirrors_wait(spvo_ctx *c, const spvo_detect_mirrors *m) {
  if (!c || !m || m->token < 0 || m->token >= RING) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (!m->desc[0] && !m->desc[1]) return SPVO_OK;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  return wait_event(c->ev_copy[m->token]) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "event synchronisation failed");
}

int spvo_detect_collect_mirrors(spvo_ctx *c, double P_l[12], double P_r[12], spvo_detect_mirrors *out) {
  if (!c || !out) return fail(c, SPVO_ERR_INVALID, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  std::memset(out, 0, sizeof *out);
  return detect_wait(c, P_l, P_r, nullptr, nullptr, nullptr, nullptr, out);
}

// ---------------------------------------------------------------- ORB (classic front end, orb.hip.h)
namespace {
uint32_t host_hash32(uint32_t x) { x ^= x >> 16; x *= 0x7FEB352Du; x ^= x >> 15; x *= 0x846CA68Bu; x ^= x >> 16; return x; }
// the 256 test pairs: isotropic Gaussian of the original BRIEF (sigma = patch / 5), fixed seed, rounded, kept inside the patch
// (the same construction as oracle/cpu/orb_cpu.inc; tests/test_gpu_orb.py compares the two tables)
void orb_host_tables(std::vector<float> &pattern, float taps[7], std::vector<signed char> &disc) {
  constexpr int PATCH = 31, HALF = ORB_HALF;
  pattern.resize(1024);
  uint32_t state = 0x9E3779B9u;
  auto uni = [&]() { state = host_hash32(state + 0x6D2B79F5u); return ((state >> 8) + 0.5f) / 16777216.0f; };
  auto gauss = [&]() { const float u1 = uni(), u2 = uni(); return std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2); };
  for (int i = 0; i < 1024; ++i) {
    float v = gauss() * (PATCH / 5.0f);
    v = std::min(std::max(v, -(float)(HALF - 2)), (float)(HALF - 2));
    pattern[i] = std::round(v);
  }
  float sum = 0;
  for (int i = 0; i < 7; ++i) { taps[i] = std::exp(-0.5f * (i - 3) * (i - 3) / 4.0f); sum += taps[i]; }
  for (int i = 0; i < 7; ++i) taps[i] /= sum;
  disc.clear();
  for (int dy = -HALF; dy <= HALF; ++dy) {
    const int lim = (int)std::floor(std::sqrt((double)HALF * HALF - dy * dy));
    for (int dx = -lim; dx <= lim; ++dx) { disc.push_back((signed char)dx); disc.push_back((signed char)dy); }
  }
}
}  // namespace

int spvo_orb_tables(float *pattern, float *taps) {
  std::vector<float> p;
  std::vector<signed char> d;
  float t[7];
  orb_host_tables(p, t, d);
  if (pattern) std::memcpy(pattern, p.data(), 1024 * sizeof(float));
  if (taps) std::memcpy(taps, t, sizeof t);
  return SPVO_OK;
}

int spvo_orb_detect(spvo_ctx *c, const uint8_t *img, int rows, int cols, size_t stride, int nfeatures, spvo_orb_keypoint *kps, uint8_t *desc, int cap, int *n_out) {
  if (!c || !img || !n_out || rows <= 0 || cols <= 0 || stride < (size_t)cols || nfeatures <= 0 || cap < 0 || (cap > 0 && (!kps || !desc)))
    return fail(c, SPVO_ERR_INVALID, "bad argument");
  static_assert(sizeof(spvo_orb_keypoint) == sizeof(OrbKeypoint), "keypoint records differ");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  *n_out = 0;
  hipStream_t st = c->stream2;
  auto &o = c->orb;
  // ---- level geometry and per-level quota (the reference's parameters: 8 levels, scale 1.2)
  constexpr float SCALE = 1.2f;
  int ph[ORB_LEVELS], pw[ORB_LEVELS], want[ORB_LEVELS];
  float lscale[ORB_LEVELS];
  size_t off[ORB_LEVELS + 1];
  {
    float scale = 1.f;
    const float f = 1.0f / SCALE;
    float n_level = nfeatures * (1 - f) / (1 - std::pow(f, (float)ORB_LEVELS));
    int assigned = 0;
    off[0] = 0;
    for (int l = 0; l < ORB_LEVELS; ++l, scale *= SCALE) {
      ph[l] = (int)std::lround(rows / scale); pw[l] = (int)std::lround(cols / scale);
      lscale[l] = scale;
      want[l] = l == ORB_LEVELS - 1 ? std::max(nfeatures - assigned, 0) : (int)std::lround(n_level);
      assigned += want[l];
      n_level *= f;
      off[l + 1] = off[l] + (((size_t)ph[l] * pw[l] + 255) & ~(size_t)255);
    }
  }
  const int surv_cap = (rows / 2 + 1) * (cols / 2 + 1);   // 3x3 suppression: at most one survivor per 2x2 block
  const int kp_cap = nfeatures;
  // what THIS image needs: the pyramid (all levels side by side), one key / rank entry per possible survivor of every level, the
  // resize tables of levels 1..7.  All three depend on rows and cols separately (a 100 x 1500 image needs longer tables than a
  // 400 x 400 one although it has fewer pixels), so each is compared with what is allocated.
  size_t need_keys = 0, need_tab = 0;
  for (int l = 0; l < ORB_LEVELS; ++l) {
    need_keys += (size_t)std::min(surv_cap, (ph[l] / 2 + 1) * (pw[l] / 2 + 1));
    if (l > 0) need_tab += (size_t)3 * (pw[l] + ph[l]);
  }
  const size_t need_pyr = off[ORB_LEVELS] + 256;
  if (need_pyr > o.pyr_cap || need_keys > o.key_cap || need_tab > o.tab_cap || kp_cap > o.kp_cap) {
    HIP_TRY(c, hipStreamSynchronize(st));
    for (void *p : {(void *)o.im, (void *)o.score, (void *)o.blur, (void *)o.tmp, (void *)o.keys, (void *)o.rank, (void *)o.out_xy, (void *)o.counters, (void *)o.tab,
                    (void *)o.kps, (void *)o.desc}) if (p) (void)hipFree(p);
    o.im = o.score = o.blur = nullptr; o.tmp = nullptr; o.keys = nullptr; o.rank = o.out_xy = o.counters = o.tab = nullptr; o.kps = nullptr; o.desc = nullptr;
    const size_t pyr = std::max(need_pyr, o.pyr_cap), kall = std::max(need_keys, o.key_cap), tabn = std::max(need_tab, o.tab_cap);
    const int kpn = std::max(kp_cap, o.kp_cap);
    o.pyr_cap = o.key_cap = o.tab_cap = 0; o.kp_cap = 0;   // a failed allocation below leaves a context that spvo_destroy and a later call can still handle
    int rc;
    if ((rc = dev_alloc(c, &o.im, pyr)) || (rc = dev_alloc(c, &o.score, pyr)) || (rc = dev_alloc(c, &o.blur, pyr)) || (rc = dev_alloc(c, &o.tmp, pyr)) ||
        (rc = dev_alloc(c, &o.keys, kall)) || (rc = dev_alloc(c, &o.rank, kall)) || (rc = dev_alloc(c, &o.out_xy, 2 * kall)) ||
        (rc = dev_alloc(c, &o.counters, (size_t)ORB_LEVELS * NMS_COUNTER_INTS)) || (rc = dev_alloc(c, &o.tab, tabn)) || (rc = dev_alloc(c, &o.kps, kpn)) ||
        (rc = dev_alloc(c, &o.desc, (size_t)kpn * 32)))
      return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (dev_alloc clears on the network stream)
    o.pyr_cap = pyr; o.key_cap = kall; o.tab_cap = tabn; o.kp_cap = kpn;
    o.tab_rows = o.tab_cols = 0;
  }
  // resize tables of all levels, one upload per image size
  size_t toff[ORB_LEVELS] = {0};
  {
    size_t t = 0;
    for (int l = 1; l < ORB_LEVELS; ++l) { toff[l] = t; t += (size_t)3 * (pw[l] + ph[l]); }
    if (o.tab_rows != rows || o.tab_cols != cols) {
      std::vector<int> all, xi, xa0, xa1, yi, yb0, yb1;
      for (int l = 1; l < ORB_LEVELS; ++l) {
        linear_coeffs(pw[l], pw[l - 1], xi, xa0, xa1);
        linear_coeffs(ph[l], ph[l - 1], yi, yb0, yb1);
        for (auto *v : {&xi, &xa0, &xa1, &yi, &yb0, &yb1}) all.insert(all.end(), v->begin(), v->end());
      }
      HIP_TRY(c, hipStreamSynchronize(st));
      HIP_TRY(c, hipMemcpy(o.tab, all.data(), all.size() * sizeof(int), hipMemcpyHostToDevice));
      o.tab_rows = rows; o.tab_cols = cols;
    }
  }
  if (!o.pattern) {
    std::vector<float> pat;
    std::vector<signed char> disc;
    float taps[7];
    orb_host_tables(pat, taps, disc);
    int rc;
    if ((rc = dev_alloc(c, &o.pattern, 1024)) || (rc = dev_alloc(c, &o.taps, 8)) || (rc = dev_alloc(c, &o.disc, disc.size()))) return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    HIP_TRY(c, hipMemcpy(o.pattern, pat.data(), 1024 * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(o.taps, taps, 7 * 4, hipMemcpyHostToDevice));
    HIP_TRY(c, hipMemcpy(o.disc, disc.data(), disc.size(), hipMemcpyHostToDevice));
  }
  // of a strided view (a cv::Mat ROI) only (rows - 1) * stride + cols bytes are the caller's: the last row's padding may lie
  // beyond the end of the parent allocation
  const size_t src_bytes = (size_t)(rows - 1) * stride + cols;
  if (src_bytes > o.src_cap) {
    HIP_TRY(c, hipStreamSynchronize(st));
    if (o.src) (void)hipFree(o.src);
    o.src = nullptr; o.src_cap = 0;
    int rc = dev_alloc(c, &o.src, src_bytes, false);
    if (rc) return rc;
    o.src_cap = src_bytes;
  }
  HIP_TRY(c, hipMemcpyAsync(o.src, img, src_bytes, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpy2DAsync(o.im, cols, o.src, stride, cols, rows, hipMemcpyDeviceToDevice, st));   // level 0: the image, rows packed
  // the whole image is enqueued without a host round trip: the pyramid level by level, then every stage once for all levels;
  // one counter block per level, a level's keypoints land behind those of the levels below (orb_describe_kernel sums their counts)
  HIP_TRY(c, hipMemsetAsync(o.counters, 0, (size_t)ORB_LEVELS * NMS_COUNTER_INTS * sizeof(int), st));
  OrbLevels lv;
  size_t koff = 0;
  int want_max = 0;
  for (int l = 0; l < ORB_LEVELS; ++l) {
    OrbLevel &L = lv.l[l];
    const int lcap = std::min(surv_cap, (ph[l] / 2 + 1) * (pw[l] / 2 + 1));
    L.im = o.im + off[l]; L.score = o.score + off[l]; L.blur = o.blur + off[l]; L.tmp = o.tmp + off[l];
    L.keys = o.keys + koff; L.rank = o.rank + koff; L.out_xy = o.out_xy + 2 * koff; L.counters = o.counters + l * NMS_COUNTER_INTS;
    L.h = ph[l]; L.w = pw[l]; L.cap = lcap; L.scale = lscale[l];
    L.want = (ph[l] <= 2 * ORB_EDGE + 2 || pw[l] <= 2 * ORB_EDGE + 2) ? 0 : want[l];
    want_max = std::max(want_max, L.want);
    koff += lcap;
    if (l > 0) hipLaunchKernelGGL(orb_resize_kernel, dim3((pw[l] + 63) / 64, (ph[l] + 3) / 4), dim3(256), 0, st, o.im + off[l - 1], ph[l - 1], pw[l - 1], pw[l - 1], L.im, ph[l], pw[l],
                                  o.tab + toff[l]);
  }
  if (want_max > 0) {
    const dim3 grid((cols + 63) / 64, (rows + 3) / 4, ORB_LEVELS);
    hipLaunchKernelGGL(orb_fast_kernel, grid, dim3(256), 0, st, lv, ORB_FAST_T);
    hipLaunchKernelGGL(orb_collect_kernel, grid, dim3(256), 0, st, lv);
    hipLaunchKernelGGL(orb_rank_kernel, dim3(128, ORB_LEVELS), dim3(256), 0, st, lv);
    hipLaunchKernelGGL(orb_write_kernel, dim3(32, ORB_LEVELS), dim3(256), 0, st, lv);
    hipLaunchKernelGGL(orb_blur_h_kernel, grid, dim3(256), 0, st, lv, o.taps);
    hipLaunchKernelGGL(orb_blur_v_kernel, grid, dim3(256), 0, st, lv, o.taps);
    hipLaunchKernelGGL(orb_describe_kernel, dim3((want_max + 3) / 4, ORB_LEVELS), dim3(256), 0, st, lv, o.disc, o.pattern, o.kps, o.desc, kp_cap);
  }
  HIP_TRY(c, hipGetLastError());
  int cnt[ORB_LEVELS * NMS_COUNTER_INTS];
  HIP_TRY(c, hipMemcpyAsync(cnt, o.counters, sizeof cnt, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  int base = 0;
  for (int l = 0; l < ORB_LEVELS; ++l) {
    if (cnt[l * NMS_COUNTER_INTS + 3]) return fail(c, SPVO_ERR_CAPACITY, "ORB: corner buffer overflow at level %d", l);
    base += cnt[l * NMS_COUNTER_INTS + 2];
  }
  base = std::min(base, kp_cap);
  *n_out = base;
  const int ncopy = std::min(base, cap);
  if (ncopy > 0) {
    HIP_TRY(c, hipMemcpyAsync(kps, o.kps, (size_t)ncopy * sizeof(OrbKeypoint), hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipMemcpyAsync(desc, o.desc, (size_t)ncopy * 32, hipMemcpyDeviceToHost, st));
    HIP_TRY(c, hipStreamSynchronize(st));
  }
  return SPVO_OK;
}

}  // extern "C"
