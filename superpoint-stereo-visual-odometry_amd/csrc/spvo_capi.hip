// spvo_capi.hip -- the extern "C" shim declared in include/spvo.h: context,
// weight loading / repacking, the network executor and one entry point per
// reference stage.  Everything heavy is a hand-written gfx950 kernel from the
// headers next to this file; this file only allocates, launches and copies.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>
#include <deque>
#include <time.h>

#include "../../include/spvo.h"
#include "conv_mfma.hip.h"
#include "match.hip.h"
#include "odometry.hip.h"
#include "orb.hip.h"
#include "conv_f16.hip.h"
#include "conv_bf16x3.hip.h"
#include "conv_wino.hip.h"
#include "conv_wino2.hip.h"
#include "conv_i8.hip.h"
#include "post.hip.h"

using namespace spvo;

namespace {

thread_local std::string g_error;  // for calls without a context

constexpr int RING = 4;          // buffer sets a detector submission owns (network outputs, heat map, NMS state, counters, host mirrors)
constexpr int MAX_INFLIGHT = 3;  // detector submissions that may be queued at once (RING - 1: the set of the pair just completed still serves its matches)
constexpr int N_SLOTS = 10;      // feature slots: 5 stereo pairs (previous, current and three in flight)

struct Tensor {
  int ch = 0, level = 0, H = 0, W = 0, hp = 0, wp = 0;
  bool nhwc = false;  // dense [B][H][W][C] (descriptor map) instead of padded planes
  bool f16 = false;   // FP16 engines: C8 fp16 [C/8][Hp][Wp][8] instead of fp32 planes (per_image still counts floats = 4 bytes)
  bool i8 = false;    // INT8 engines: C16 int8 [C/16][Hp][Wp][16]
  bool s3 = false;    // FP32 engines in split mode: C8x3 bf16 pieces [C/8][3][Hp][Wp][8] (conv_bf16x3.hip.h)
  float scale = 0.f;  // INT8 engines: real value = q * scale (calibrated)
  float *d = nullptr;
  float *dr[RING] = {nullptr, nullptr, nullptr, nullptr};  // network outputs only: one buffer per submission set (d == dr[0])
  size_t per_image = 0;  // floats
};

enum { OP_CONV = 1, OP_MAXPOOL = 2, OP_L2NORM = 3, OP_DWCONV = 4 };
enum { FLAG_RELU = 1, FLAG_POOL = 2, FLAG_BN = 4, FLAG_ADD = 8 };

struct Op {
  int type = 0, in = 0, out = 0, out_c_off = 0, in_c_off = 0, cin = 0, cout = 0, ks = 0, flags = 0;
  int ck = 0, n_chunks = 0, co_tiles = 0, wr = 0, wc = 0;
  int residual = 0;  // tensor added before the last ReLU (FLAG_ADD)
  bool merged = false;     // FP32 engines: this op's output channels are computed by the previous op's launch (sibling layers
                           // that read the same tensor and write adjacent channel ranges of one tensor: convPa + convDa)
  bool wino = false;       // FP32 engines: this 3x3 layer runs the Winograd F(2x2,3x3) kernel (conv_wino.hip.h)
  bool wino_narrow = false;   // ... with 32 instead of 64 output channels per workgroup (layers whose 64-channel tiles would leave CUs idle)
  bool wino2 = false;      // ... its 8-wave form (conv_wino2.hip.h: two waves per SIMD; the default, SPVO_WINO2=0 keeps the 4-wave form)
  bool dominant = false;   // the op with the most FLOPs: launched under its own kernel name (TAG = 1)
  float *d_w = nullptr, *d_b = nullptr, *d_bn_scale = nullptr, *d_bn_shift = nullptr;
  int *d_sched = nullptr;      // Winograd layers (8-wave form): {8 band counters, workgroups done}, zero between launches (SPVO_WINO_DYNAMIC=0: none)
  _Float16 *d_w16 = nullptr;   // FP16 engines: pack_conv_weights_f16()
  int8_t *d_w8 = nullptr;      // INT8 engines: pack_conv_weights_i8()
  unsigned short *d_ws3 = nullptr;   // FP32 engines in split mode: pack_conv_weights_s3()
  int *d_wq32 = nullptr;       // INT8 engines, depthwise: quantised weights [C][9] as int32
  float *d_qm = nullptr;       // INT8 engines: weight scale * input scale per output channel
  float inv_s_out = 0.f, s_res = 0.f;
  double flops_per_image = 0;
  int stage = -1;
};

struct Stage {
  std::string name;
  double total_ms = 0;
  long long calls = 0;
  double flops = 0, bytes = 0;  // algorithmic, per call (last call's value)
};

struct Pending { int stage; hipEvent_t e0, e1; };

struct FeatureSlot {
  int n = 0;
  bool filled = false;      // a submission has written (or is writing) this slot
  int *d_xy = nullptr;      // [cap][2] int
  float *d_xyf = nullptr;   // [cap][2] float
  float *d_desc = nullptr;  // [cap][256]
  int *d_n = nullptr;       // device copy of n (read by kernels enqueued before the host knows n)
  float *d_sqn = nullptr;   // [cap] squared norms of the descriptors (written by the sampler)
  unsigned long long gen = 0;  // bumped whenever the slot is rewritten
};

// matches enqueued together with the detector (spvo_set_prematch); results live in pinned memory
struct MatchCache {
  bool valid = false;
  int slot_a = -1, slot_b = -1, selector = 0, cross = 0;
  float ratio = 0.f;
  unsigned long long gen_a = 0, gen_b = 0;
  int2 *h_out = nullptr;      // pinned [cap] packed {train_idx, distance bits}
};

struct MatchScratch {         // one set per concurrently enqueued match
  float *d_na = nullptr, *d_nb = nullptr, *d_best_d2 = nullptr;
  float *d_dt = nullptr;      // [cap][match_ldt(cap)] approximate squared distances of every pair (K12a -> K12b)
  int *d_best_idx = nullptr;
  unsigned long long *d_train_best = nullptr;
  unsigned char *d_a8 = nullptr, *d_b8 = nullptr;   // fp8 copies of both sides (spvo_set_match_fp8)
  int2 *d_out = nullptr;      // packed result, points into spvo_ctx::d_match_out
};

struct NmsImage {
  NmsBuffers b;
};

}  // namespace

struct CropGeomS { int row_off = 0, col_off = 0, crop_rows = 0, crop_cols = 0; float scale = 1.f; };

struct PendingDetect {           // one spvo_detect*_submit in flight
  CropGeomS g;
  int rows = 0, cols = 0, slot_l = 0, slot_r = 0, prev_l = -1, ring = 0;
  bool rematch = false;          // the temporal partner's keypoints were redone after this submission matched against them
  int extras = 0;                // spvo_detect_submit: bit 0 resized images, bit 1 descriptors travel to the set's pinned mirrors
};


struct spvo_ctx {
  spvo_config cfg;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // fused solve: overlaps with a detector submission in flight
  hipStream_t stream_t = nullptr;  // detector tail (heat map, NMS, sampling, matching): overlaps with the NEXT submission's network
  hipStream_t post = nullptr;      // where post-processing is enqueued right now: `stream`, or `stream_t` for a submission
  std::deque<PendingDetect> pendq;
  int cur_ring = 0;                // set whose network outputs the running forward pass writes
  unsigned submit_count = 0;
  std::string error;
  bool weights = false;
  bool fp16 = false;               // the loaded engine's precision
  bool split_req = false;          // spvo_set_fp32_split / SPVO_FP32_SPLIT: FP32 engines loaded from now on run on the bf16x3 kernels
  bool s3 = false;                 // the loaded FP32 engine runs in split mode
  size_t head_start = 0;           // ops [head_start, end) = the 1x1 heads + L2 norm: a submission runs them on the tail stream
  bool int8 = false;
  int H = 0, W = 0, Hc = 0, Wc = 0, B = 0;
  int num_cus = 256;

  std::vector<Tensor> tensors;
  std::vector<Op> ops;
  int t_input = 0, t_det = 0, t_desc = 0;
  int last_batch = 0;

  // post-processing buffers
  float *d_dense_in = nullptr;   // [B][H][W] staging for spvo_forward
  float *d_det_dense = nullptr;  // [B][65][Hc][Wc]
  float *d_heat = nullptr;       // [B][H][W], inside d_heat_base with a 64-float guard on both sides
  float *d_heat_base = nullptr;
  NmsImage nms[2];
  int surv_cap = 0;
  int *h_counters = nullptr;     // pinned [2][NMS_COUNTER_INTS]
  uint8_t *d_img[2] = {nullptr, nullptr};
  size_t img_cap = 0;
  uint8_t *d_resized = nullptr;  // [2][H][W]
  int *d_tab = nullptr;          // resize tables: xi,xa0,xa1 [W] ; yi,yb0,yb1 [H]
  int tab_rows = -1, tab_cols = -1;
  FeatureSlot slots[N_SLOTS];
  int *d_xy_tmp = nullptr;       // [cap][2] for spvo_sample_descriptors
  float *d_desc_tmp = nullptr;   // [cap][256]
  float *h_xy = nullptr;         // pinned [2][cap][2]
  int last_slot_l = -1;          // left slot of the previous submission (temporal partner)

  // matching scratch
  int match_cap = 0;
  float *d_ma = nullptr, *d_mb = nullptr;
  MatchScratch ms[2];
  int2 *d_match_out = nullptr;   // [2][cap]: both jobs' results leave in one copy
  int2 *h_match_out[RING] = {nullptr, nullptr, nullptr, nullptr};   // pinned [2][cap] per submission set
  int2 *h_match_tmp = nullptr;   // pinned [cap] for the synchronous entry points
  int *d_counters_all = nullptr; // [RING sets + 1 stand-alone set][2 images][NMS_COUNTER_INTS]
  float *d_xy_stage = nullptr;   // [RING][2][cap][2] keypoints of both images as floats: one copy per submission
  MatchCache mcache[RING][2];    // [submission set][stereo, temporal]
  // per submission set (index 0 doubles as the stand-alone entry points' set)
  NmsImage nms_r[RING][2];
  float *d_heat_r[RING] = {nullptr, nullptr, nullptr, nullptr}, *d_heat_base_r[RING] = {nullptr, nullptr, nullptr, nullptr};
  int *h_counters_r[RING] = {nullptr, nullptr, nullptr, nullptr};
  float *h_xy_r[RING] = {nullptr, nullptr, nullptr, nullptr};
  // host-image submissions (spvo_detect_submit): per set, pinned staging + device copies of the two input images, a resized-image
  // buffer of its own and pinned mirrors of the resized images and of the descriptors
  uint8_t *h_img_r[RING] = {nullptr, nullptr, nullptr, nullptr}, *d_img_r[RING] = {nullptr, nullptr, nullptr, nullptr};
  size_t img_cap_r = 0;          // bytes per image in those buffers
  uint8_t *d_resized_r[RING] = {nullptr, nullptr, nullptr, nullptr}, *h_resized_r[RING] = {nullptr, nullptr, nullptr, nullptr};
  float *h_desc_r[RING] = {nullptr, nullptr, nullptr, nullptr};   // [2][cap][256]
  hipEvent_t ev_net[RING] = {nullptr, nullptr, nullptr, nullptr}, ev_tail[RING] = {nullptr, nullptr, nullptr, nullptr};
  bool match_fp8 = false;        // fp8 shortlist GEMM (approximate; spvo_set_match_fp8)
  bool prematch = false;
  int pm_selector = SPVO_SELECT_KNN, pm_cross = 0;
  float pm_ratio = 0.8f;

  // odometry scratch
  int odo_cap = 0, ransac_cap = 0, obs_cap = 0;
  double *d_P = nullptr;         // Pl[12], Pr[12], K[9], prior[6], start[7]
  float *d_pts_a = nullptr, *d_pts_b = nullptr, *d_xyz = nullptr;
  RansacWork rw{};
  ObsDev *d_obs = nullptr;
  RefineOut *d_refine = nullptr;
  // ORB detector / extractor of the classic front end (orb.hip.h): buffers grow on demand
  struct OrbBufs {
    size_t px_cap = 0;        // pixels of level 0 the image buffers are sized for
    int kp_cap = 0;
    uint8_t *im = nullptr, *score = nullptr, *blur = nullptr, *src = nullptr;   // im: all pyramid levels back to back
    float *tmp = nullptr, *pattern = nullptr, *taps = nullptr;
    unsigned long long *keys = nullptr;
    int *rank = nullptr, *out_xy = nullptr, *counters = nullptr, *tab = nullptr;
    signed char *disc = nullptr;
    OrbKeypoint *kps = nullptr;
    uint8_t *desc = nullptr;
    size_t src_cap = 0;
    int tab_rows = 0, tab_cols = 0;   // image size the resize tables in `tab` belong to
  } orb;
  // Hamming matcher (classic front end's binary descriptors): rows padded to 16 words
  int ham_cap = 0;
  uint32_t *d_ham_a = nullptr, *d_ham_b = nullptr;
  int *d_ham_idx = nullptr;
  float *d_ham_dist = nullptr;
  unsigned long long *d_ham_vote = nullptr;
  // fused solve: one packed input, one packed result
  struct SolvePending { bool active = false; int n = 0, refinement_degree = 0; double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0}; } solve_pending;   // spvo_solve_submit .. _wait
  hipEvent_t ev_solve = nullptr;
  int solve_cap = 0;
  char *d_solve_in = nullptr, *h_solve_in = nullptr;    // 64 doubles + 12*cap words
  double *d_solve_res = nullptr, *h_solve_res = nullptr;  // ransac[8] gate[16] refine[12] + pad
  char *d_solve_o = nullptr, *h_solve_o = nullptr;      // xyz [3n] floats, inliers [n] ints
  int *d_ctl = nullptr;

  // profiling
  bool prof = false;
  int prof_only = -1;            // >= 0: only this stage is timed (spvo_profile_only)
  std::vector<Stage> stages;
  std::vector<Pending> pending;
  std::vector<hipEvent_t> free_events;
};

namespace {

// SPVO_TRUNK_TIMING diagnostics: where the host spends its time between two submissions (maxima over the 200 submissions of a report)
struct HostDiag { double t_last_submit = 0, max_interval = 0, max_tail_wait = 0, max_solve_wait = 0; int match_miss = 0, late = 0, depth_sum = 0; } g_diag;
inline double diag_now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }

int fail(spvo_ctx *c, int code, const char *fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->error = buf;
  g_error = buf;
  return code;
}

#define HIP_TRY(c, expr)                                                                     \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(c, SPVO_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                  __FILE__, __LINE__);                                                       \
  } while (0)

template <typename T>
int dev_alloc(spvo_ctx *c, T **p, size_t count, bool zero = true) {
  HIP_TRY(c, hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
  if (zero) HIP_TRY(c, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
  return SPVO_OK;
}

int stage_id(spvo_ctx *c, const std::string &name) {
  for (size_t i = 0; i < c->stages.size(); ++i)
    if (c->stages[i].name == name) return (int)i;
  Stage s;
  s.name = name;
  c->stages.push_back(s);
  return (int)c->stages.size() - 1;
}

// SPVO_SPIN_WAIT=1: the waits of the per-frame path poll their event instead of sleeping in the driver (a sleeping host thread
// pays the wake-up latency of its core at every wait).  Off by default: on the bench box it changed nothing (the waits are
// dominated by GPU time), and a ROS node should not burn a core while it waits.
bool spin_wait_enabled() {
  static const bool on = std::getenv("SPVO_SPIN_WAIT") && std::atoi(std::getenv("SPVO_SPIN_WAIT")) != 0;
  return on;
}
hipError_t wait_event(hipEvent_t ev) {
  if (spin_wait_enabled()) {
    for (long spins = 0; spins < 20000000; ++spins) {   // far longer than any wait of this library; then fall back to the blocking form
      const hipError_t e = hipEventQuery(ev);
      if (e != hipErrorNotReady) return e;
      __builtin_ia32_pause();
    }
  }
  return hipEventSynchronize(ev);
}

hipEvent_t get_event(spvo_ctx *c) {
  if (!c->free_events.empty()) {
    hipEvent_t e = c->free_events.back();
    c->free_events.pop_back();
    return e;
  }
  hipEvent_t e;
  (void)hipEventCreate(&e);
  return e;
}

void resolve_pending(spvo_ctx *c) {
  if (c->pending.empty()) return;
  for (auto &p : c->pending) {
    float ms = 0;
    (void)hipEventSynchronize(p.e1);
    if (hipEventElapsedTime(&ms, p.e0, p.e1) == hipSuccess) {
      c->stages[p.stage].total_ms += ms;
      c->stages[p.stage].calls += 1;
    }
    c->free_events.push_back(p.e0);
    c->free_events.push_back(p.e1);
  }
  c->pending.clear();
}

struct ScopedStage {
  spvo_ctx *c;
  int id = -1;
  hipEvent_t e0 = nullptr;
  hipStream_t st = nullptr;
  ScopedStage(spvo_ctx *ctx, int stage, double flops = 0, double bytes = 0, hipStream_t stream = nullptr) : c(ctx) {
    if (!c->prof || stage < 0 || (c->prof_only >= 0 && stage != c->prof_only)) return;
    id = stage;
    st = stream ? stream : (c->post ? c->post : c->stream);
    if (flops > 0) c->stages[id].flops = flops;
    if (bytes > 0) c->stages[id].bytes = bytes;
    e0 = get_event(c);
    (void)hipEventRecord(e0, st);
  }
  ~ScopedStage() {
    if (id < 0) return;
    hipEvent_t e1 = get_event(c);
    (void)hipEventRecord(e1, st);
    c->pending.push_back({id, e0, e1});
    if (c->pending.size() > 8192) resolve_pending(c);
  }
};

// ---------------------------------------------------------------- conv dispatch
// a tensor's buffer for the submission being enqueued (tensors a tail reads have one per submission set)
inline float *ring_ptr(spvo_ctx *c, const Tensor &t) { return t.dr[c->cur_ring] ? t.dr[c->cur_ring] : t.d; }

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

// Winograd F(2x2, 3x3) instance of a 3x3 layer (conv_wino.hip.h / conv_wino2.hip.h): one tile shape, one workgroup per CU (157 KB of
// LDS); W2 selects the 8-wave form (two waves per SIMD, 512 threads)
template <bool POOL, bool RELU, int TAG, bool ODD = false, bool W2 = false, bool NARROW = false>
int launch_conv_wino_instance(spvo_ctx *c, const ConvArgs &args, hipStream_t stream) {
  static bool ready[64] = {};
  const int dev = c->cfg.device & 63;
  const int n_tiles = args.tiles_x * args.tiles_y * args.co_tiles * args.batch;
  // One workgroup per CU (157 KB of LDS), each walking ceil(n_tiles / grid) tiles.  The grid is the SMALLEST one that keeps that
  // number of rounds: 3330 tiles are 14 rounds on 256 CUs and still 14 rounds on 238, and the 18 CUs left over take the
  // small kernels of the other streams (tail of the previous pair, solver): with all 256 CUs claimed, any of those
  // kernels sitting on a CU when a layer starts keeps that layer's last workgroup waiting for a CU.
  const int rounds = (n_tiles + c->num_cus - 1) / c->num_cus;
  const int grid = (n_tiles + rounds - 1) / rounds;
  if constexpr (W2) {
    auto k = conv_wino2_kernel<POOL, RELU, TAG, ODD, NARROW>;
    if (!ready[dev]) {
      HIP_TRY(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, WINO2_LDS_BYTES));
      ready[dev] = true;
    }
    ConvArgs a2 = args;
    if (rounds < 2 || args.n_chunks < 4) a2.sched = nullptr;   // one tile per workgroup: nothing to hand out; short K loops: see the kernel
    hipLaunchKernelGGL(k, dim3(grid), dim3(512), WINO2_LDS_BYTES, stream, a2);
  } else {
    auto k = conv_wino_kernel<POOL, RELU, TAG, ODD>;
    if (!ready[dev]) {
      HIP_TRY(c, hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, WinoTile::LDS_BYTES));
      ready[dev] = true;
    }
    hipLaunchKernelGGL(k, dim3(grid), dim3(256), WinoTile::LDS_BYTES, stream, args);
  }
  HIP_TRY(c, hipGetLastError());
  return SPVO_OK;
}

template <bool W2>
int launch_conv_wino_sel(spvo_ctx *c, const ConvArgs &args, bool relu, bool pool, bool dominant, bool narrow, hipStream_t stream) {
  const ConvArgs &a = args;
  if constexpr (W2)
    if (narrow) {   // 32 output channels per workgroup (layers that would leave CUs idle with 64)
      if (!pool && ((a.H | a.W) & 1)) return relu ? launch_conv_wino_instance<false, true, 0, true, true, true>(c, args, stream) : launch_conv_wino_instance<false, false, 0, true, true, true>(c, args, stream);
      if (pool) return relu ? launch_conv_wino_instance<true, true, 0, false, true, true>(c, args, stream) : launch_conv_wino_instance<true, false, 0, false, true, true>(c, args, stream);
      return relu ? launch_conv_wino_instance<false, true, 0, false, true, true>(c, args, stream) : launch_conv_wino_instance<false, false, 0, false, true, true>(c, args, stream);
    }
  if (!pool && ((a.H | a.W) & 1)) return relu ? launch_conv_wino_instance<false, true, 0, true, W2>(c, args, stream) : launch_conv_wino_instance<false, false, 0, true, W2>(c, args, stream);
  if (dominant && relu) return pool ? launch_conv_wino_instance<true, true, 1, false, W2>(c, args, stream) : launch_conv_wino_instance<false, true, 1, false, W2>(c, args, stream);
  if (pool) return relu ? launch_conv_wino_instance<true, true, 0, false, W2>(c, args, stream) : launch_conv_wino_instance<true, false, 0, false, W2>(c, args, stream);
  return relu ? launch_conv_wino_instance<false, true, 0, false, W2>(c, args, stream) : launch_conv_wino_instance<false, false, 0, false, W2>(c, args, stream);
}

int launch_conv_wino(spvo_ctx *c, const ConvArgs &a, int batch, bool relu, bool pool, bool dominant, bool w2, bool narrow, hipStream_t stream) {
  ConvArgs args = a;
  args.tiles_x = (a.W + WinoTile::TW - 1) / WinoTile::TW;
  args.tiles_y = (a.H + WinoTile::TH - 1) / WinoTile::TH;
  args.batch = batch;
  return w2 ? launch_conv_wino_sel<true>(c, args, relu, pool, dominant, narrow, stream) : launch_conv_wino_sel<false>(c, args, relu, pool, dominant, false, stream);
}

// Variant choice for a layer.  3x3: every tile variant has a measured rate on perfectly divisible shapes
// (tools/conv_bench sweep, launched back to back; TFLOP/s on 256 CUs) -- the smaller tiles let 2-3 workgroups share
// a CU, whose staging, barriers and output bursts then hide under each other's matrix instructions -- and the
// layer's time is that rate applied to the padded work of the busiest CU: ceil(tiles / CUs) tiles of TH x TW pixels
// (the model reproduces the sweep's times within 4 %).  Chunks of 4 channels (3-4 workgroups per CU) measure a
// further 2-3 % faster in isolation but 4 % SLOWER inside the pipeline, where the chip is shared with the previous
// pair's post-processing and the solver (tools/tune_variants.sh), so they are not offered.
// 1x1: tile by padded work / grid fill.
void choose_variant(int ks, int H, int W, int co_tiles, int batch, bool pool, int num_cus, int *wr, int *wc, int *ck, bool split = false) {
  struct Cand { int wr, wc, ck; double rate; };
  static const Cand k3_f32[] = {{2, 2, 8, 130.0}, {2, 1, 8, 135.0}, {1, 2, 8, 135.0}, {1, 1, 8, 133.0}};
  // split (bf16x3) kernels: the 8x64 tile reads the fewest operands per matrix instruction (14 ds_read_b128 per 24) and
  // measures 4-9 % faster per pixel than the others (conv1b 263 vs 281 us, conv2b 76 vs 83 us)
  static const Cand k3_s3[] = {{2, 2, 8, 108.0}, {2, 1, 8, 100.0}, {1, 2, 8, 97.0}, {1, 1, 8, 97.0}};
  const Cand *k3 = split ? k3_s3 : k3_f32;
  static const Cand k1[] = {{2, 2, 16, 1.0 / 1.00}, {1, 2, 16, 1.0 / 1.03}, {2, 1, 16, 1.0 / 1.03}, {1, 1, 16, 1.0 / 1.06}};
  const Cand *cands = ks == 3 ? k3 : k1;
  const int nc = 4;
  double best = 1e300;
  for (int i = 0; i < nc; ++i) {
    const Cand &v = cands[i];
    if (pool && v.wr != 2) continue;                 // a 2x2 pooling window lives in one wave
    if (ks == 1 && !pool && v.wr == 2 && v.wc == 1) continue;
    const int th = 4 * v.wr, tw = 32 * v.wc;
    const int tx = (W + tw - 1) / tw, ty = (H + th - 1) / th;
    const long tiles = (long)tx * ty * co_tiles * batch;
    double cost;
    if (ks == 3) cost = (double)((tiles + num_cus - 1) / num_cus) * th * tw / v.rate;
    else cost = (double)tx * tw * ty * th / v.rate / std::min(1.0, (double)tiles / num_cus);
    if (cost < best) { best = cost; *wr = v.wr; *wc = v.wc; *ck = v.ck; }
  }
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
                                                op.d_bn_scale, op.d_bn_shift, ti.H, ti.W, ti.hp, ti.wp, to.ch, op.out_c_off, op.cout)
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
  if (op.wino) return launch_conv_wino(c, a, batch, relu, pool, op.dominant, op.wino2, op.wino_narrow, stream);
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

int launch_op(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream) {
  if (op.merged) return SPVO_OK;
  const Tensor &ti = c->tensors[op.in];
  const Tensor &to = c->tensors[op.out];
  if (op.type == OP_CONV || op.type == OP_DWCONV) {
    ScopedStage st(c, op.stage, op.flops_per_image * batch, 0, stream);
    return c->int8 ? launch_conv8(c, op, img0, batch, stream) : c->fp16 ? launch_conv16(c, op, img0, batch, stream)
           : c->s3 ? launch_conv_s3(c, op, img0, batch, stream) : launch_conv(c, op, img0, batch, stream);
  }
  const float *tin = (ti.dr[c->cur_ring] ? ti.dr[c->cur_ring] : ti.d) + (size_t)img0 * ti.per_image;
  float *tout = (to.dr[c->cur_ring] ? to.dr[c->cur_ring] : to.d) + (size_t)img0 * to.per_image;
  ScopedStage st(c, op.stage, 0, 0, stream);
  if (op.type == OP_MAXPOOL && ti.f16) {
    dim3 grid((to.W + 63) / 64, (to.H + 3) / 4, batch * (to.ch / 8));
    hipLaunchKernelGGL(maxpool2_f16_kernel, grid, dim3(256), 0, stream, (const _Float16 *)tin, (_Float16 *)tout, to.H, to.W, ti.hp, ti.wp, to.hp, to.wp);
  } else if (op.type == OP_MAXPOOL) {
    dim3 grid((to.W + 63) / 64, (to.H + 3) / 4, batch * to.ch);
    hipLaunchKernelGGL(maxpool2_kernel, grid, dim3(256), 0, stream, tin, tout, to.ch, to.H, to.W, ti.hp, ti.wp, to.hp, to.wp);
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
int run_ops(spvo_ctx *c, int batch, size_t first, size_t last, hipStream_t stream) {
  for (size_t i = first; i < last && i < c->ops.size(); ++i) {
    int rc = launch_op(c, c->ops[i], 0, batch, stream);
    if (rc) return rc;
  }
  return SPVO_OK;
}

int run_network(spvo_ctx *c, int batch) {
  ScopedStage net(c, stage_id(c, "net"));
  int rc = run_ops(c, batch, 0, c->ops.size(), c->stream);
  if (rc) return rc;
  c->last_batch = batch;
  return SPVO_OK;
}

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
                     (resized_dst ? resized_dst : c->d_resized) + (size_t)slot0 * c->H * c->W, tin.d + (size_t)slot0 * tin.per_image, tin.per_image, tin.hp, tin.wp, identity);
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

// `n_launch` round launches + collect + rank + write for `nimg` images, then the counters travel
// to the host in one copy.  Launch 0 of a batch never exits early.
int launch_nms_rounds(spvo_ctx *c, int nimg, const NmsPair &np, int set, int n_launch, int *zero_next) {
  hipStream_t st = c->post;
  const float *heat = c->d_heat_r[set % RING];
  for (int l = 0; l < n_launch; ++l) {
    if (c->cfg.dist_thresh == 4)
      hipLaunchKernelGGL((nms_round_kernel<NMS_INNER, 4>), dim3(NMS_GRID, nimg), dim3(256), 0, st, heat, c->H, c->W, 4, np, l);
    else
      hipLaunchKernelGGL((nms_round_kernel<NMS_INNER, 0>), dim3(NMS_GRID, nimg), dim3(256), 0, st, heat, c->H, c->W, c->cfg.dist_thresh, np, l);
  }
  hipLaunchKernelGGL(nms_collect_kernel, dim3(NMS_GRID, nimg), dim3(256), 0, st, heat, c->H, c->W, c->cfg.border_remove, c->surv_cap, np);
  hipLaunchKernelGGL(nms_rank_kernel, dim3(128, nimg), dim3(256), 0, st, c->surv_cap, np);
  hipLaunchKernelGGL(nms_write_kernel, dim3(32, nimg), dim3(256), 0, st, c->H, c->cfg.max_keypoints, c->surv_cap, np, zero_next);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(c->h_counters_r[set % RING], np.b[0].counters, (size_t)nimg * NMS_COUNTER_INTS * sizeof(int), hipMemcpyDeviceToHost, st));
  return SPVO_OK;
}

// processOneHeatmap for images [0, nimg).  Real heat maps settle in 2-3 launches of 4 in-kernel
// rounds; adversarial ones (e.g. a constant image: one decision chain across the whole picture)
// simply take more batches -- every launch decides at least the best undecided candidate, so the
// loop terminates.  nms_enqueue only submits; nms_settle runs after the caller's sync and returns
// 1 if it had to redo work (the caller then re-runs what depends on the keypoints).
constexpr int NMS_FIRST = 3;

// more rounds for the (rare) submissions whose first batch left candidates undecided
int nms_settle(spvo_ctx *c, int nimg, const NmsPair &np, int set, bool *redone) {
  int last = NMS_FIRST;
  *redone = false;
  const int *hc = c->h_counters_r[set % RING];
  for (;;) {
    bool pending = false;
    for (int i = 0; i < nimg; ++i) pending |= hc[i * NMS_COUNTER_INTS + 8 + last - 1] != 0;
    if (!pending) break;
    *redone = true;
    last = NMS_MAX_LAUNCH;
    for (int i = 0; i < nimg; ++i)   // keep n_cand, clear the rest of the block
      HIP_TRY(c, hipMemsetAsync(np.b[i].counters + 1, 0, (NMS_COUNTER_INTS - 1) * sizeof(int), c->post));
    int rc = launch_nms_rounds(c, nimg, np, set, last, nullptr);
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
  int rc = launch_nms_rounds(c, nimg, np, RING, NMS_FIRST, nullptr);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  bool redone;
  return nms_settle(c, nimg, np, RING, &redone);
}

int ensure_match(spvo_ctx *c, int na, int nb) {
  const int need = std::max(na, nb);
  if (need <= c->match_cap) return SPVO_OK;
  const int cap = std::max(need, std::max(c->cfg.max_keypoints, 1024));
  HIP_TRY(c, hipDeviceSynchronize());
  // every pointer is cleared as it is freed and the capacity drops to 0 first: an allocation failure further down leaves a
  // context that spvo_destroy (and a later, smaller request) can still handle
  c->match_cap = 0;
  auto drop = [](auto *&p) { if (p) (void)hipFree(p); p = nullptr; };
  drop(c->d_ma); drop(c->d_mb); drop(c->d_match_out);
  for (auto &m : c->ms) {
    drop(m.d_na); drop(m.d_nb); drop(m.d_best_d2); drop(m.d_dt); drop(m.d_best_idx); drop(m.d_train_best); drop(m.d_a8); drop(m.d_b8);
    m.d_out = nullptr;
  }
  for (auto &p : c->h_match_out) { if (p) (void)hipHostFree(p); p = nullptr; }
  if (c->h_match_tmp) (void)hipHostFree(c->h_match_tmp);
  c->h_match_tmp = nullptr;
  for (auto &set : c->mcache) for (auto &mc : set) { mc.valid = false; mc.h_out = nullptr; }
  int rc;
  if ((rc = dev_alloc(c, &c->d_ma, (size_t)cap * MATCH_D))) return rc;
  if ((rc = dev_alloc(c, &c->d_mb, (size_t)cap * MATCH_D))) return rc;
  if ((rc = dev_alloc(c, &c->d_match_out, (size_t)2 * cap))) return rc;
  for (int k = 0; k < 2; ++k) {
    MatchScratch &m = c->ms[k];
    if ((rc = dev_alloc(c, &m.d_na, cap + 4))) return rc;   // K12b reads the norms four at a time
    if ((rc = dev_alloc(c, &m.d_nb, cap + 4))) return rc;
    if ((rc = dev_alloc(c, &m.d_best_d2, (size_t)cap * 2))) return rc;
    if ((rc = dev_alloc(c, &m.d_dt, (size_t)cap * match_ldt(cap)))) return rc;
    if ((rc = dev_alloc(c, &m.d_best_idx, (size_t)cap * 2))) return rc;
    if ((rc = dev_alloc(c, &m.d_train_best, cap))) return rc;
    if ((rc = dev_alloc(c, &m.d_a8, (size_t)cap * MATCH_D))) return rc;
    if ((rc = dev_alloc(c, &m.d_b8, (size_t)cap * MATCH_D))) return rc;
    m.d_out = c->d_match_out + (size_t)k * cap;
  }
  for (int r = 0; r < RING; ++r) HIP_TRY(c, hipHostMalloc((void **)&c->h_match_out[r], (size_t)2 * cap * sizeof(int2)));
  HIP_TRY(c, hipHostMalloc((void **)&c->h_match_tmp, (size_t)cap * sizeof(int2)));
  for (int par = 0; par < RING; ++par)
    for (int k = 0; k < 2; ++k) {
      c->mcache[par][k].h_out = c->h_match_out[par] + (size_t)k * cap;
      c->mcache[par][k].valid = false;
    }
  c->match_cap = cap;
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SPVO_OK;
}

struct MatchReq {
  const float *dA, *dB;
  int na, nb;                     // counts, or upper bounds when the pointers are set
  const int *na_ptr, *nb_ptr;
  const float *sqA, *sqB;         // squared norms if already known (feature slots), else NULL
};

// Enqueue 1 or 2 matches as ONE set of launches (blockIdx.z / .y = job) and one result copy:
// packed {train_idx, distance bits} for job k lands at host_out + k*match_cap.
int enqueue_matches(spvo_ctx *c, const MatchReq *req_in, int njobs, int selector, int cross_check, float ratio, int2 *host_out) {
  MatchJobs jobs;
  int na_max = 0, nb_max = 0;
  // NN + cross-check is cv::batchDistance's crosscheck: the search runs from the TRAIN rows to the query rows, so the
  // two sides change places for the distance GEMM and the re-rank; match_select_cross_kernel writes one entry per
  // query row again
  const bool swap = selector == SPVO_SELECT_NN && cross_check;
  MatchReq req[2];
  for (int k = 0; k < njobs; ++k) {
    req[k] = req_in[k];
    if (swap) {
      std::swap(req[k].dA, req[k].dB); std::swap(req[k].na, req[k].nb);
      std::swap(req[k].na_ptr, req[k].nb_ptr); std::swap(req[k].sqA, req[k].sqB);
    }
    MatchScratch &m = c->ms[k];
    MatchJob &j = jobs.j[k];
    j.A = req[k].dA; j.B = req[k].dB;
    j.na = req[k].na; j.nb = req[k].nb;
    j.na_ptr = req[k].na_ptr; j.nb_ptr = req[k].nb_ptr;
    j.nA = req[k].sqA ? req[k].sqA : m.d_na;
    j.nB = req[k].sqB ? req[k].sqB : m.d_nb;
    j.dt = m.d_dt; j.best_d2 = m.d_best_d2; j.best_idx = m.d_best_idx; j.train_best = m.d_train_best; j.out = m.d_out;
    j.A8 = j.B8 = nullptr;
    if (c->match_fp8) {
      hipLaunchKernelGGL(desc_to_fp8_kernel, dim3((req[k].na + 3) / 4), dim3(256), 0, c->post, req[k].dA, req[k].na, req[k].na_ptr, m.d_a8);
      hipLaunchKernelGGL(desc_to_fp8_kernel, dim3((req[k].nb + 3) / 4), dim3(256), 0, c->post, req[k].dB, req[k].nb, req[k].nb_ptr, m.d_b8);
      j.A8 = m.d_a8; j.B8 = m.d_b8;
    }
    na_max = std::max(na_max, req[k].na);
    nb_max = std::max(nb_max, req[k].nb);
    if (!req[k].sqA) hipLaunchKernelGGL(row_sqnorm_kernel, dim3((req[k].na + 3) / 4), dim3(256), 0, c->post, req[k].dA, req[k].na, req[k].na_ptr, m.d_na);
    if (!req[k].sqB) hipLaunchKernelGGL(row_sqnorm_kernel, dim3((req[k].nb + 3) / 4), dim3(256), 0, c->post, req[k].dB, req[k].nb, req[k].nb_ptr, m.d_nb);
    if (swap) HIP_TRY(c, hipMemsetAsync(m.d_train_best, 0xFF, (size_t)req[k].nb * sizeof(unsigned long long), c->post));
  }
  if (njobs == 1) jobs.j[1] = jobs.j[0];
  const int groups = (nb_max + MATCH_TT - 1) / MATCH_TT;
  const int ldt = match_ldt(c->match_cap);
  const double fl = 2.0 * na_max * nb_max * MATCH_D * njobs;
  ScopedStage st(c, stage_id(c, "match"), fl, 4.0 * (na_max + nb_max) * MATCH_D * njobs);
  const size_t lds = MATCH_LDS_BYTES;
  static bool attr[64] = {};
  if (!attr[c->cfg.device & 63]) {
    HIP_TRY(c, hipFuncSetAttribute((const void *)match_gemm_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr[c->cfg.device & 63] = true;
  }
  {
    ScopedStage sg(c, stage_id(c, "match_gemm"), fl, 4.0 * (na_max + nb_max) * MATCH_D * njobs);
    if (c->match_fp8) hipLaunchKernelGGL(match_gemm_kernel<true>, dim3(groups, (na_max + MATCH_QT - 1) / MATCH_QT, njobs), dim3(256), MATCH_LDS_BYTES_FP8, c->post, jobs, ldt);
    else hipLaunchKernelGGL(match_gemm_kernel<false>, dim3(groups, (na_max + MATCH_QT - 1) / MATCH_QT, njobs), dim3(256), lds, c->post, jobs, ldt);
  }
  {
    ScopedStage sr(c, stage_id(c, "match_rerank"));
    const float err = c->match_fp8 ? MATCH_ERR_REL_FP8 : MATCH_ERR_REL;
    const dim3 gr((na_max + 3) / 4, njobs);
    // rows of up to 1024 columns stay in registers between the two passes of the re-rank (8 chunks for 2048 columns
    // measured 2.4x SLOWER than the chunked form: 232 registers, 70 KB of LDS)
    if (nb_max <= 1024) hipLaunchKernelGGL(match_rerank_kernel<4>, gr, dim3(256), sizeof(MatchRerankLds<4>), c->post, jobs, ldt, err, selector, cross_check, ratio);
    else hipLaunchKernelGGL(match_rerank_kernel<0>, gr, dim3(256), sizeof(MatchRerankLds<0>), c->post, jobs, ldt, err, selector, cross_check, ratio);
  }
  if (swap) hipLaunchKernelGGL(match_select_cross_kernel, dim3((nb_max + 255) / 256, njobs), dim3(256), 0, c->post, jobs);
  HIP_TRY(c, hipGetLastError());
  // jobs' outputs are adjacent in d_match_out (stride match_cap): one copy
  const size_t count = (njobs == 2) ? (size_t)c->match_cap + req_in[1].na : (size_t)req_in[0].na;
  HIP_TRY(c, hipMemcpyAsync(host_out, c->d_match_out, count * sizeof(int2), hipMemcpyDeviceToHost, c->post));
  return SPVO_OK;
}

void unpack_match(const int2 *packed, int n, int32_t *train_idx, float *distance) {
  for (int i = 0; i < n; ++i) {
    train_idx[i] = packed[i].x;
    std::memcpy(&distance[i], &packed[i].y, sizeof(float));
  }
}

int run_match(spvo_ctx *c, const MatchReq &r, int selector, int cross_check, float ratio, int32_t *train_idx, float *distance) {
  if (r.na == 0) return SPVO_OK;
  if (r.nb == 0) {
    for (int i = 0; i < r.na; ++i) { train_idx[i] = -1; distance[i] = 0.f; }
    return SPVO_OK;
  }
  int rc = enqueue_matches(c, &r, 1, selector, cross_check, ratio, c->h_match_tmp);
  if (rc) return rc;
  HIP_TRY(c, hipStreamSynchronize(c->post));
  unpack_match(c->h_match_tmp, r.na, train_idx, distance);
  return SPVO_OK;
}

int ensure_odometry(spvo_ctx *c, int n, int iterations, int n_obs) {
  int rc;
  if (!c->d_P) {
    if ((rc = dev_alloc(c, &c->d_P, 64))) return rc;
    if ((rc = dev_alloc(c, &c->rw.result, 8))) return rc;
    if ((rc = dev_alloc(c, &c->d_refine, 1))) return rc;
  }
  // a buffer is cleared as it is freed and its capacity drops to 0 before the reallocation: a failure half-way leaves a context
  // that spvo_destroy and a later call can still handle
  auto drop = [](auto *&p) { if (p) (void)hipFree(p); p = nullptr; };
  if (n > c->odo_cap) {
    const int cap = std::max(n, 2048);
    c->odo_cap = 0;
    drop(c->d_pts_a); drop(c->d_pts_b); drop(c->d_xyz); drop(c->rw.inliers);
    if ((rc = dev_alloc(c, &c->d_pts_a, (size_t)cap * 3))) return rc;
    if ((rc = dev_alloc(c, &c->d_pts_b, (size_t)cap * 3))) return rc;
    if ((rc = dev_alloc(c, &c->d_xyz, (size_t)cap * 3))) return rc;
    if ((rc = dev_alloc(c, &c->rw.inliers, cap))) return rc;
    c->odo_cap = cap;
  }
  if (iterations > c->ransac_cap) {
    const int cap = std::max(iterations, 512);
    c->ransac_cap = 0;
    drop(c->rw.counts); drop(c->rw.poses);
    if ((rc = dev_alloc(c, &c->rw.counts, cap))) return rc;
    if ((rc = dev_alloc(c, &c->rw.poses, (size_t)cap * 7))) return rc;
    c->ransac_cap = cap;
  }
  if (n_obs > c->obs_cap) {
    const int cap = std::max(n_obs, 8192);
    c->obs_cap = 0;
    drop(c->d_obs);
    if ((rc = dev_alloc(c, &c->d_obs, cap))) return rc;
    c->obs_cap = cap;
  }
  return SPVO_OK;
}

// post-processing issued by a synchronous entry point while submissions are queued goes behind them
struct PostScope {
  spvo_ctx *c;
  explicit PostScope(spvo_ctx *ctx) : c(ctx) { c->post = c->pendq.empty() ? c->stream : c->stream_t; }
  ~PostScope() { c->post = c->stream; }
};

void free_plan(spvo_ctx *c) {
  for (auto &t : c->tensors) {
    if (t.d) (void)hipFree(t.d);
    for (int r = 1; r < RING; ++r) if (t.dr[r]) (void)hipFree(t.dr[r]);
  }
  for (auto &o : c->ops)
  {
    for (float *p : {o.d_w, o.d_b, o.d_bn_scale, o.d_bn_shift}) if (p) (void)hipFree(p);
    if (o.d_w16) (void)hipFree(o.d_w16);
    if (o.d_w8) (void)hipFree(o.d_w8);
    if (o.d_ws3) (void)hipFree(o.d_ws3);
    if (o.d_wq32) (void)hipFree(o.d_wq32);
    if (o.d_qm) (void)hipFree(o.d_qm);
    if (o.d_sched) (void)hipFree(o.d_sched);
  }
  c->tensors.clear(); c->ops.clear(); c->weights = false; c->fp16 = false; c->int8 = false; c->s3 = false;
}

}  // namespace

// ===========================================================================
extern "C" {

void spvo_default_config(spvo_config *cfg) {
  if (!cfg) return;
  cfg->device = 0;
  cfg->net_height = 360;
  cfg->net_width = 1176;
  cfg->max_batch = 2;
  cfg->conf_thresh = 0.015f;
  cfg->dist_thresh = 4;
  cfg->border_remove = 4;
  cfg->max_keypoints = 1000;
  cfg->bug_compat_p = 1;
}

const char *spvo_last_error(const spvo_ctx *ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }
void spvo_internal_set_error(const char *msg) { g_error = msg ? msg : ""; }   // spvo_comm.hip reports through the same channel

int spvo_create(const spvo_config *cfg, spvo_ctx **out) {
  if (!cfg || !out) return fail(nullptr, SPVO_ERR_INVALID, "null argument");
  *out = nullptr;
  if (cfg->net_height <= 0 || cfg->net_width <= 0 || cfg->net_height % 8 || cfg->net_width % 8)
    return fail(nullptr, SPVO_ERR_INVALID, "net size %dx%d must be positive multiples of 8 (feature_detection.hpp:296)", cfg->net_height, cfg->net_width);
  if (cfg->max_batch != 1 && cfg->max_batch != 2)
    return fail(nullptr, SPVO_ERR_INVALID, "Wrong batch size (%d)", cfg->max_batch);  // nn.cpp:490
  if (cfg->max_keypoints <= 0 || cfg->dist_thresh < 0 || cfg->dist_thresh > NMS_PAD || cfg->border_remove < 0)
    return fail(nullptr, SPVO_ERR_INVALID, "bad post-processing parameters (dist_thresh must be in [0, %d])", NMS_PAD);
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(nullptr, SPVO_ERR_DEVICE, "no HIP device visible: this library has no CPU path");
  if (cfg->device < 0 || cfg->device >= ndev) return fail(nullptr, SPVO_ERR_DEVICE, "device %d out of range (%d visible)", cfg->device, ndev);
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, cfg->device) != hipSuccess) return fail(nullptr, SPVO_ERR_DEVICE, "hipGetDeviceProperties failed");
  if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
    return fail(nullptr, SPVO_ERR_DEVICE, "device %d is %s; the kernels are built for gfx950 only", cfg->device, prop.gcnArchName);
  spvo_ctx *c = new spvo_ctx();
  c->cfg = *cfg;
  c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
  c->H = cfg->net_height; c->W = cfg->net_width; c->Hc = c->H / 8; c->Wc = c->W / 8; c->B = 2;
  // Non-blocking streams: work the caller puts on the NULL stream (a framework's default stream, a blocking hipMemcpy) must not
  // serialise the three streams of the pipeline against each other.  Device pointers handed to the *_dev entry points
  // have to be complete when the call is made (include/spvo.h).
  if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->stream_t, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(nullptr, SPVO_ERR_DEVICE, "cannot create a stream on device %d", cfg->device);
  }
  c->post = c->stream;
  if (const char *e = std::getenv("SPVO_FP32_SPLIT")) c->split_req = std::atoi(e) != 0;
  for (int r = 0; r < RING; ++r)
    if (hipEventCreateWithFlags(&c->ev_net[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_tail[r], hipEventDisableTiming) != hipSuccess ) {
      spvo_destroy(c);
      return fail(nullptr, SPVO_ERR_DEVICE, "cannot create events on device %d", cfg->device);
    }
  int rc = SPVO_OK;
  const size_t hw = (size_t)c->H * c->W;
  const int cap = cfg->max_keypoints;
  // one survivor per (dist+1)^2 cell at most
  const int cell = cfg->dist_thresh + 1;
  c->surv_cap = ((c->H + cell - 1) / cell) * ((c->W + cell - 1) / cell) + 64;
  do {
    if ((rc = dev_alloc(c, &c->d_dense_in, 2 * hw))) break;
    if ((rc = dev_alloc(c, &c->d_det_dense, (size_t)2 * 65 * c->Hc * c->Wc))) break;
    if ((rc = dev_alloc(c, &c->d_counters_all, (size_t)(RING + 1) * 2 * NMS_COUNTER_INTS))) break;   // RING submission sets, stand-alone
    if ((rc = dev_alloc(c, &c->d_xy_stage, (size_t)RING * 2 * cfg->max_keypoints * 2))) break;
    for (int r = 0; r < RING && !rc; ++r) {
      if ((rc = dev_alloc(c, &c->d_heat_base_r[r], 2 * hw + 128))) break;
      c->d_heat_r[r] = c->d_heat_base_r[r] + 64;   // K10 reads aligned float4 rows that may start left of column 0
    }
    if (rc) break;
    c->d_heat_base = c->d_heat_base_r[0];
    c->d_heat = c->d_heat_r[0];
    if ((rc = dev_alloc(c, &c->d_resized, 2 * hw))) break;
    if ((rc = dev_alloc(c, &c->d_tab, (size_t)3 * (c->H + c->W)))) break;
    for (int r = 0; r < RING && !rc; ++r)
      for (int i = 0; i < 2 && !rc; ++i) {
        NmsBuffers &b = c->nms_r[r][i].b;
        if ((rc = dev_alloc(c, &b.state, (size_t)(c->H + 2 * NMS_PAD) * nms_state_pitch(c->W)))) break;
        if ((rc = dev_alloc(c, &b.cand, hw))) break;
        b.counters = nullptr;   // set per submission (nms_pair)
        if ((rc = dev_alloc(c, &b.surv_key, c->surv_cap))) break;
        if ((rc = dev_alloc(c, &b.rank, c->surv_cap))) break;
        if ((rc = dev_alloc(c, &b.out_xy, (size_t)cap * 2))) break;
      }
    if (rc) break;
    for (int i = 0; i < 2; ++i) c->nms[i] = c->nms_r[0][i];   // the stand-alone entry points work in set 0
    for (int i = 0; i < N_SLOTS && !rc; ++i) {
      if ((rc = dev_alloc(c, &c->slots[i].d_xy, (size_t)cap * 2))) break;
      if ((rc = dev_alloc(c, &c->slots[i].d_xyf, (size_t)cap * 2))) break;
      if ((rc = dev_alloc(c, &c->slots[i].d_desc, (size_t)cap * 256))) break;
      if ((rc = dev_alloc(c, &c->slots[i].d_n, 1))) break;
      if ((rc = dev_alloc(c, &c->slots[i].d_sqn, cap + 4))) break;   // K12b reads the norms four at a time
    }
    if (rc) break;
    if ((rc = dev_alloc(c, &c->d_xy_tmp, (size_t)cap * 2))) break;
    if ((rc = dev_alloc(c, &c->d_desc_tmp, (size_t)cap * 256))) break;
    for (int r = 0; r < RING && !rc; ++r)
      if (hipHostMalloc((void **)&c->h_counters_r[r], 2 * NMS_COUNTER_INTS * sizeof(int)) != hipSuccess ||
          hipHostMalloc((void **)&c->h_xy_r[r], (size_t)2 * cap * 2 * sizeof(float)) != hipSuccess) rc = fail(c, SPVO_ERR_DEVICE, "hipHostMalloc failed");
    if (rc) break;
    c->h_counters = c->h_counters_r[0];
    c->h_xy = c->h_xy_r[0];
    if ((rc = ensure_match(c, cap, cap))) break;
  } while (0);
  if (rc) {
    g_error = c->error;
    spvo_destroy(c);
    return rc;
  }
  (void)hipStreamSynchronize(c->stream);
  *out = c;
  return SPVO_OK;
}

void spvo_destroy(spvo_ctx *c) {
  if (!c) return;
  (void)hipSetDevice(c->cfg.device);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  if (c->stream_t) (void)hipStreamSynchronize(c->stream_t);
  if (c->ev_solve) (void)hipEventDestroy(c->ev_solve);
  resolve_pending(c);
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  free_plan(c);
  void *ptrs[] = {c->d_dense_in, c->d_det_dense, c->d_resized, c->d_tab, c->d_img[0], c->d_img[1], c->d_xy_tmp, c->d_desc_tmp,
                  c->d_ma, c->d_mb, c->d_match_out, c->d_counters_all, c->d_xy_stage,
                  c->ms[0].d_na, c->ms[0].d_nb, c->ms[0].d_best_d2, c->ms[0].d_dt, c->ms[0].d_best_idx, c->ms[0].d_train_best, c->ms[0].d_a8, c->ms[0].d_b8,
                  c->ms[1].d_na, c->ms[1].d_nb, c->ms[1].d_best_d2, c->ms[1].d_dt, c->ms[1].d_best_idx, c->ms[1].d_train_best, c->ms[1].d_a8, c->ms[1].d_b8,
                  c->d_P, c->d_pts_a, c->d_pts_b, c->d_xyz, c->rw.counts, c->rw.poses, c->rw.result, c->rw.inliers, c->d_obs, c->d_refine};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (int r = 0; r < RING; ++r) {
    for (int i = 0; i < 2; ++i) {
      NmsBuffers &b = c->nms_r[r][i].b;
      void *q[] = {b.state, b.cand, b.surv_key, b.rank, b.out_xy};
      for (void *p : q) if (p) (void)hipFree(p);
    }
    if (c->d_heat_base_r[r]) (void)hipFree(c->d_heat_base_r[r]);
    if (c->h_counters_r[r]) (void)hipHostFree(c->h_counters_r[r]);
    if (c->h_xy_r[r]) (void)hipHostFree(c->h_xy_r[r]);
    if (c->d_img_r[r]) (void)hipFree(c->d_img_r[r]);
    if (c->h_img_r[r]) (void)hipHostFree(c->h_img_r[r]);
    if (c->d_resized_r[r]) (void)hipFree(c->d_resized_r[r]);
    if (c->h_resized_r[r]) (void)hipHostFree(c->h_resized_r[r]);
    if (c->h_desc_r[r]) (void)hipHostFree(c->h_desc_r[r]);
    for (hipEvent_t e : {c->ev_net[r], c->ev_tail[r]}) if (e) (void)hipEventDestroy(e);
  }
  for (int i = 0; i < N_SLOTS; ++i) {
    void *q[] = {c->slots[i].d_xy, c->slots[i].d_xyf, c->slots[i].d_desc, c->slots[i].d_n, c->slots[i].d_sqn};
    for (void *p : q) if (p) (void)hipFree(p);
  }
  for (void *hp : {(void *)c->h_solve_in, (void *)c->h_solve_res, (void *)c->h_solve_o}) if (hp) (void)hipHostFree(hp);
  for (void *dp : {(void *)c->d_solve_in, (void *)c->d_solve_res, (void *)c->d_solve_o, (void *)c->d_ctl}) if (dp) (void)hipFree(dp);
  for (void *dp : {(void *)c->d_ham_a, (void *)c->d_ham_b, (void *)c->d_ham_idx, (void *)c->d_ham_dist, (void *)c->d_ham_vote}) if (dp) (void)hipFree(dp);
  for (void *dp : {(void *)c->orb.im, (void *)c->orb.score, (void *)c->orb.blur, (void *)c->orb.src, (void *)c->orb.tmp, (void *)c->orb.pattern, (void *)c->orb.taps, (void *)c->orb.keys,
                   (void *)c->orb.rank, (void *)c->orb.out_xy, (void *)c->orb.counters, (void *)c->orb.tab, (void *)c->orb.disc, (void *)c->orb.kps, (void *)c->orb.desc})
    if (dp) (void)hipFree(dp);
  for (auto hp : c->h_match_out) if (hp) (void)hipHostFree(hp);
  if (c->h_match_tmp) (void)hipHostFree(c->h_match_tmp);
  if (c->stream_t) (void)hipStreamDestroy(c->stream_t);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  delete c;
}

int spvo_load_weights(spvo_ctx *c, const char *path) {
  if (!c || !path) return fail(c, SPVO_ERR_INVALID, "null argument");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  FILE *f = std::fopen(path, "rb");
  if (!f) return fail(c, SPVO_ERR_IO, "no such engine file: %s", path);  // nn.cpp:53-55
  std::fseek(f, 0, SEEK_END);
  const long sz = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  std::vector<unsigned char> buf((size_t)std::max(sz, 0L));
  const size_t got = std::fread(buf.data(), 1, buf.size(), f);
  std::fclose(f);
  if (got != buf.size() || buf.size() < 48 || std::memcmp(buf.data(), "SPVW0003", 8) != 0)
    return fail(c, SPVO_ERR_IO, "%s is not a SPVW0003 weight file", path);
  const uint32_t *hdr = (const uint32_t *)(buf.data() + 8);
  const uint32_t nt = hdr[0], no = hdr[1];
  size_t pos = 40;
  if (buf.size() < pos + (size_t)nt * 8 + (size_t)no * 72 + 8) return fail(c, SPVO_ERR_IO, "%s: truncated", path);
  // Everything that can be checked on the file alone is checked BEFORE the loaded plan is dropped: a missing, truncated or
  // corrupt file leaves the engine that was loaded before in place.  Ids are compared as unsigned (0xFFFFFFFF is not -1).
  if (hdr[5] > 2) return fail(c, SPVO_ERR_IO, "%s: unknown precision %u", path, hdr[5]);
  if (nt == 0 || nt > 65536 || no > 65536) return fail(c, SPVO_ERR_IO, "%s: implausible tensor / op count", path);
  if (hdr[2] >= nt || hdr[3] >= nt || hdr[4] >= nt) return fail(c, SPVO_ERR_IO, "%s: bad binding tensor ids", path);
  {
    size_t p = pos;
    for (uint32_t i = 0; i < nt; ++i, p += 8) {
      const uint32_t *r = (const uint32_t *)(buf.data() + p);
      if (r[1] > 3 || ((uint32_t)c->H >> r[1]) << r[1] != (uint32_t)c->H) return fail(c, SPVO_ERR_IO, "%s: bad tensor level", path);
    }
    for (uint32_t i = 0; i < no; ++i, p += 72) {
      const uint32_t *r = (const uint32_t *)(buf.data() + p);
      if (r[1] >= nt || r[2] >= nt || r[8] >= nt) return fail(c, SPVO_ERR_IO, "%s: op %u: bad tensor id", path, i);
    }
    uint64_t n_payload;
    std::memcpy(&n_payload, buf.data() + p, 8);
    p += 8;
    if (n_payload > (buf.size() - p) / 4) return fail(c, SPVO_ERR_IO, "%s: truncated payload", path);
  }
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight");
  free_plan(c);   // from here on a failure (device allocation, unsupported layer) leaves the context without an engine
  c->t_input = (int)hdr[2]; c->t_det = (int)hdr[3]; c->t_desc = (int)hdr[4];
  c->fp16 = hdr[5] == 1;   // engine built for FP16 (engine_generation.py's --fp16; the file name says FP16, nn.cpp:44-49)
  c->int8 = hdr[5] == 2;   // INT8 engine (BASELINE config 5; no counterpart in the reference): calibrated activation scales in the file
  const uint32_t act_scale_off = hdr[6];
  for (uint32_t i = 0; i < nt; ++i) {
    const uint32_t *r = (const uint32_t *)(buf.data() + pos);
    pos += 8;
    Tensor t;
    t.ch = r[0]; t.level = r[1];
    if (t.level > 3 || (c->H >> t.level) << t.level != c->H) return fail(c, SPVO_ERR_IO, "bad tensor level");
    t.H = c->H >> t.level; t.W = c->W >> t.level;
    t.hp = padded_h(t.H); t.wp = padded_w(t.W);
    c->tensors.push_back(t);
  }
  struct Raw { uint32_t v[12]; uint64_t w_off, b_off, bn_off; };
  static_assert(sizeof(Raw) == 72, "op record layout");
  std::vector<Raw> raws(no);
  for (uint32_t i = 0; i < no; ++i) { std::memcpy(&raws[i], buf.data() + pos, 72); pos += 72; }
  uint64_t nfl;
  std::memcpy(&nfl, buf.data() + pos, 8);
  pos += 8;
  const float *payload = (const float *)(buf.data() + pos);   // nfl was checked against the file size above

  for (uint32_t i = 0; i < no; ++i) {
    const Raw &r = raws[i];
    Op op;
    op.type = r.v[0]; op.in = r.v[1]; op.out = r.v[2]; op.out_c_off = r.v[3];
    op.cin = r.v[4] & 0xFFFF; op.in_c_off = r.v[4] >> 16; op.cout = r.v[5]; op.ks = r.v[6]; op.flags = r.v[7];
    op.residual = r.v[8];
    if (op.type == OP_L2NORM) c->tensors[op.out].nhwc = true;
    c->ops.push_back(op);
  }
  if (c->int8) {
    if (act_scale_off + (uint64_t)nt > nfl) return fail(c, SPVO_ERR_IO, "%s: activation scales out of range", path);
    for (uint32_t t = 0; t < nt; ++t) {
      c->tensors[t].scale = payload[act_scale_off + t];
      c->tensors[t].i8 = (c->tensors[t].ch % 16) == 0;   // fewer channels (mbv's stem): an fp32 plane
      if (!(c->tensors[t].scale > 0.f)) return fail(c, SPVO_ERR_IO, "%s: tensor %u has no activation scale", path, t);
    }
    c->tensors[c->t_input].i8 = c->tensors[c->t_det].i8 = c->tensors[c->t_desc].i8 = false;   // fp32 bindings
    for (const auto &op : c->ops) {
      if (op.type == OP_L2NORM) c->tensors[op.in].i8 = false;
      if (op.type == OP_MAXPOOL) return fail(c, SPVO_ERR_IO, "%s: INT8 engines have no stand-alone max-pool (squeeze graph)", path);
    }
  }
  if (c->fp16) {
    // half precision between the fp32 network input and the fp32 outputs (nn.cpp:117): every tensor but the input,
    // output_det, the raw descriptor map and output_desc is C8 fp16
    // (a tensor whose channel count is not a multiple of 8 -- mbv's one-channel stem -- stays an fp32 plane that
    // holds fp16 values)
    for (auto &t : c->tensors) t.f16 = (t.ch % 8) == 0;
    c->tensors[c->t_input].f16 = c->tensors[c->t_det].f16 = c->tensors[c->t_desc].f16 = false;
    for (const auto &op : c->ops)
      if (op.type == OP_L2NORM) c->tensors[op.in].f16 = false;
  }
  c->s3 = c->split_req && !c->fp16 && !c->int8;
  if (c->s3) {
    // split mode: every tensor between the fp32 network input and the fp32 outputs holds bf16 triples
    for (auto &t : c->tensors) t.s3 = true;
    c->tensors[c->t_input].s3 = c->tensors[c->t_det].s3 = c->tensors[c->t_desc].s3 = false;
    for (const auto &op : c->ops) {
      if (op.type == OP_L2NORM) c->tensors[op.in].s3 = false;
      if (op.type != OP_CONV && op.type != OP_L2NORM) return fail(c, SPVO_ERR_IO, "%s: the split-fp32 mode covers convolution + L2-norm graphs (VGG SuperPoint) only", path);
    }
    for (const auto &t : c->tensors) if (t.s3 && (t.ch % 8)) return fail(c, SPVO_ERR_IO, "%s: split-fp32 mode: a %d-channel tensor", path, t.ch);
  }
  // The network's tail end -- the trailing run of unpooled 1x1 convolutions and the L2 normalisation: convPb, convDb + norm --
  // is tiny and launch-bound (64 us for 2.2 GFLOP); a submission runs it on the tail stream, where it fills the CUs the next
  // pair's trunk leaves idle, instead of on the network stream, which is the one that limits the frame rate.
  c->head_start = c->ops.size();
  if (!(std::getenv("SPVO_HEADS_ON_TAIL") && std::atoi(std::getenv("SPVO_HEADS_ON_TAIL")) == 0))
    while (c->head_start > 0) {
      const Op &o = c->ops[c->head_start - 1];
      const bool head = o.type == OP_L2NORM || (o.type == OP_CONV && o.ks == 1 && !(o.flags & FLAG_POOL) && o.cin > 1);
      if (!head) break;
      --c->head_start;
    }
  // allocate activations (padded planes stay zero outside the interior for ever)
  for (size_t ti = 0; ti < c->tensors.size(); ++ti) {
    Tensor &t = c->tensors[ti];
    t.per_image = t.nhwc ? (size_t)t.H * t.W * t.ch : t.s3 ? (size_t)t.ch * t.hp * t.wp * 3 / 2 : (size_t)t.ch * t.hp * t.wp / (t.f16 ? 2 : t.i8 ? 4 : 1);
    int rc = dev_alloc(c, &t.d, t.per_image * c->B);
    if (rc) return rc;
    bool head_input = false;   // read by the head ops, which a submission runs on its tail stream while the next trunk already runs
    for (size_t q = c->head_start; q < c->ops.size(); ++q) head_input |= c->ops[q].in == (int)ti || ((c->ops[q].flags & FLAG_ADD) && c->ops[q].residual == (int)ti);
    if ((int)ti == c->t_det || (int)ti == c->t_desc || head_input) {   // what a submission's tail reads while the next network pass already runs
      t.dr[0] = t.d;
      for (int r = 1; r < RING; ++r)
        if ((rc = dev_alloc(c, &t.dr[r], t.per_image * c->B))) return rc;
    }
  }
  for (uint32_t i = 0; i < no; ++i) {
    Op &op = c->ops[i];
    const Raw &r = raws[i];
    const Tensor &ti = c->tensors[op.in];
    const Tensor &to = c->tensors[op.out];
    char name[64];
    if (op.type == OP_DWCONV) {
      std::snprintf(name, sizeof name, "dwconv:%u", i);
      op.stage = stage_id(c, name);
      if (op.ks != 3 || op.cin != op.cout || op.in_c_off || op.out_c_off || ti.ch != op.cin || to.ch != op.cout || to.level != ti.level ||
          (op.flags & ~FLAG_RELU))
        return fail(c, SPVO_ERR_IO, "op %u: unsupported depthwise convolution", i);
      if (r.w_off + (uint64_t)op.cout * 9 > nfl || r.b_off + op.cout > nfl) return fail(c, SPVO_ERR_IO, "op %u: weights out of range", i);
      op.flops_per_image = 2.0 * ti.H * ti.W * op.cout * 9;
      if (c->fp16 && (!ti.f16 || !to.f16)) return fail(c, SPVO_ERR_IO, "op %u: depthwise convolution of an FP16 engine needs channel counts that are multiples of 8", i);
      if (c->int8) {
        if (!ti.i8 || !to.i8) return fail(c, SPVO_ERR_IO, "op %u: depthwise convolution of an INT8 engine needs channel counts that are multiples of 16", i);
        std::vector<int8_t> wq;
        std::vector<float> ws;
        quantize_conv_weights(payload + r.w_off, op.cout, 9, wq, ws);
        std::vector<int> wq32(wq.begin(), wq.end());
        std::vector<float> qm(op.cout);
        for (int o = 0; o < op.cout; ++o) qm[o] = ws[o] * ti.scale;
        op.inv_s_out = 1.f / to.scale;
        int rc = dev_alloc(c, &op.d_wq32, wq32.size(), false);
        if (rc) return rc;
        if ((rc = dev_alloc(c, &op.d_qm, op.cout, false))) return rc;
        if ((rc = dev_alloc(c, &op.d_b, op.cout, false))) return rc;
        HIP_TRY(c, hipMemcpy(op.d_wq32, wq32.data(), wq32.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_qm, qm.data(), qm.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_b, payload + r.b_off, (size_t)op.cout * 4, hipMemcpyHostToDevice));
        continue;
      }
      std::vector<float> wdw(payload + r.w_off, payload + r.w_off + (size_t)op.cout * 9);
      if (c->fp16) for (auto &q : wdw) q = (float)(_Float16)q;
      int rc = dev_alloc(c, &op.d_w, (size_t)op.cout * 9, false);
      if (rc) return rc;
      if ((rc = dev_alloc(c, &op.d_b, op.cout, false))) return rc;
      HIP_TRY(c, hipMemcpy(op.d_w, wdw.data(), (size_t)op.cout * 9 * 4, hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(op.d_b, payload + r.b_off, (size_t)op.cout * 4, hipMemcpyHostToDevice));
    } else if (op.type == OP_CONV) {
      std::snprintf(name, sizeof name, "conv:%u", i);
      op.stage = stage_id(c, name);
      if (op.merged) continue;
      const int taps = op.ks * op.ks;
      const bool bn = op.flags & FLAG_BN, add = op.flags & FLAG_ADD;
      if ((bn && (add || !(op.flags & FLAG_RELU))) || (add && (op.flags & FLAG_RELU)))
        return fail(c, SPVO_ERR_IO, "op %u: epilogue flags 0x%x are not a graph order this library executes", i, op.flags);
      if ((bn || add) && op.cin > 1 && op.ks != 1) return fail(c, SPVO_ERR_IO, "op %u: BatchNorm / residual epilogues exist for 1x1 convolutions only", i);
      if (add) {
        const Tensor &tr = c->tensors[op.residual];
        if (op.cin == 1 || tr.ch != op.cout || tr.level != ti.level || tr.nhwc || op.residual == op.out)
          return fail(c, SPVO_ERR_IO, "op %u: bad residual tensor", i);
      }
      const int co_pad = ((op.cout + CO_TILE - 1) / CO_TILE) * CO_TILE;
      if (bn) {
        // ONNX BatchNormalization, inference form, folded to one fma: scale = gamma / sqrt(var + eps)
        if (r.bn_off + 4ull * op.cout + 1 > nfl) return fail(c, SPVO_ERR_IO, "op %u: BatchNorm parameters out of range", i);
        const float *q = payload + r.bn_off;
        std::vector<float> sc(co_pad, 0.f), sh(co_pad, 0.f);
        for (int o = 0; o < op.cout; ++o) {
          const double k = (double)q[o] / std::sqrt((double)q[3 * op.cout + o] + (double)q[4 * op.cout]);
          sc[o] = (float)k;
          sh[o] = (float)((double)q[op.cout + o] - (double)q[2 * op.cout + o] * k);
        }
        int rc = dev_alloc(c, &op.d_bn_scale, co_pad, false);
        if (rc) return rc;
        if ((rc = dev_alloc(c, &op.d_bn_shift, co_pad, false))) return rc;
        HIP_TRY(c, hipMemcpy(op.d_bn_scale, sc.data(), co_pad * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_bn_shift, sh.data(), co_pad * 4, hipMemcpyHostToDevice));
      }
      if (op.ks != 1 && op.ks != 3) return fail(c, SPVO_ERR_IO, "op %u: kernel size %d", i, op.ks);
      if (r.w_off + (uint64_t)op.cout * op.cin * taps > nfl || r.b_off + op.cout > nfl) return fail(c, SPVO_ERR_IO, "op %u: weights out of range", i);
      if (op.in_c_off + op.cin > ti.ch || op.out_c_off + op.cout > to.ch) return fail(c, SPVO_ERR_IO, "op %u: channel slice out of range", i);
      const bool pool = op.flags & FLAG_POOL;
      if (to.level != ti.level + (pool ? 1 : 0)) return fail(c, SPVO_ERR_IO, "op %u: level mismatch", i);
      if (pool && ((ti.H | ti.W) & 1)) return fail(c, SPVO_ERR_IO, "op %u: pooling an odd-sized map", i);
      op.flops_per_image = 2.0 * ti.H * ti.W * op.cout * op.cin * taps;
      const float *w = payload + r.w_off;
      const float *b = payload + r.b_off;
      // Sibling 3x3 layers (same input slice, same flags, adjacent output channel ranges of one tensor -- the two heads' first
      // convolutions convPa / convDa write channels 0..255 and 256..511 of one tensor) run as ONE layer with the output
      // channels concatenated: one launch instead of two, and 480 workgroups on 256 CUs instead of twice 240.
      std::vector<float> wcat, bcat;
      if (!c->int8 && !c->fp16 && !c->s3 && op.ks == 3 && op.cin > 1 && !bn && !add && !pool && (op.cout % CO_TILE) == 0 && i + 1 < no &&
          !(std::getenv("SPVO_MERGE_SIBLINGS") && std::atoi(std::getenv("SPVO_MERGE_SIBLINGS")) == 0)) {
        Op &nx = c->ops[i + 1];
        const Raw &rn = raws[i + 1];
        if (nx.type == OP_CONV && nx.in == op.in && nx.in_c_off == op.in_c_off && nx.cin == op.cin && nx.ks == op.ks && nx.flags == op.flags &&
            nx.out == op.out && nx.out_c_off == op.out_c_off + op.cout && nx.out_c_off + nx.cout <= to.ch &&
            rn.w_off + (uint64_t)nx.cout * nx.cin * taps <= nfl && rn.b_off + nx.cout <= nfl) {
          wcat.assign(w, w + (size_t)op.cout * op.cin * taps);
          wcat.insert(wcat.end(), payload + rn.w_off, payload + rn.w_off + (size_t)nx.cout * nx.cin * taps);
          bcat.assign(b, b + op.cout);
          bcat.insert(bcat.end(), payload + rn.b_off, payload + rn.b_off + nx.cout);
          w = wcat.data();
          b = bcat.data();
          op.cout += nx.cout;
          op.flops_per_image = 2.0 * ti.H * ti.W * op.cout * op.cin * taps;
          nx.merged = true;
        }
      }
      if (c->int8) {
        if (ti.i8 != (op.cin != 1)) return fail(c, SPVO_ERR_IO, "op %u: INT8 engine: a %d-channel input tensor stored as %s", i, op.cin, ti.i8 ? "int8" : "fp32");
        op.inv_s_out = to.i8 ? 1.f / to.scale : 0.f;
        if (op.cin == 1) {   // fp32 stem
          if (pool || add || (to.i8 && ((op.out_c_off % 16) || (op.cout % 16)))) return fail(c, SPVO_ERR_IO, "op %u: unsupported single-channel-input layer for INT8", i);
          int rc = dev_alloc(c, &op.d_w, (size_t)op.cout * taps, false);
          if (rc) return rc;
          if ((rc = dev_alloc(c, &op.d_b, op.cout, false))) return rc;
          HIP_TRY(c, hipMemcpy(op.d_w, w, (size_t)op.cout * taps * 4, hipMemcpyHostToDevice));
          HIP_TRY(c, hipMemcpy(op.d_b, b, (size_t)op.cout * 4, hipMemcpyHostToDevice));
          continue;
        }
        const int ckg = (op.ks == 3 || op.cin % 64) ? 2 : 4;   // 32 channels per chunk; 64 for 1x1 layers when they divide
        if (op.cin % (16 * ckg) || op.in_c_off % 16) return fail(c, SPVO_ERR_IO, "op %u: cin %d / channel offset %d do not fit the INT8 chunking (%d)", i, op.cin, op.in_c_off, 16 * ckg);
        if (to.i8 && ((op.cout % 16) || (op.out_c_off % 16))) return fail(c, SPVO_ERR_IO, "op %u: cout %d / channel offset %d are not multiples of 16", i, op.cout, op.out_c_off);
        if (!to.i8 && (pool || bn || add)) return fail(c, SPVO_ERR_IO, "op %u: pooled / BatchNorm / residual layer with an fp32 output", i);
        if (add) {
          const Tensor &tr = c->tensors[op.residual];
          if (!tr.i8 || tr.ch != op.cout) return fail(c, SPVO_ERR_IO, "op %u: residual tensor is not a %d-channel int8 tensor", i, op.cout);
          op.s_res = tr.scale;
        }
        op.ck = 16 * ckg;
        op.n_chunks = op.cin / op.ck;
        op.co_tiles = (op.cout + CO_TILE - 1) / CO_TILE;
        int ck_unused;
        choose_variant(op.ks, ti.H, ti.W, op.co_tiles, c->cfg.max_batch, pool, c->num_cus, &op.wr, &op.wc, &ck_unused);
        std::vector<int8_t> wq;
        std::vector<float> ws;
        quantize_conv_weights(w, op.cout, op.cin * taps, wq, ws);
        const std::vector<int8_t> pk = pack_conv_weights_i8(wq.data(), op.cout, op.cin, op.ks, ckg);
        std::vector<float> qm((size_t)op.co_tiles * CO_TILE, 0.f), bp((size_t)op.co_tiles * CO_TILE, 0.f);
        for (int o = 0; o < op.cout; ++o) { qm[o] = ws[o] * ti.scale; bp[o] = b[o]; }
        int rc = dev_alloc(c, &op.d_w8, pk.size(), false);
        if (rc) return rc;
        if ((rc = dev_alloc(c, &op.d_qm, qm.size(), false))) return rc;
        if ((rc = dev_alloc(c, &op.d_b, bp.size(), false))) return rc;
        HIP_TRY(c, hipMemcpy(op.d_w8, pk.data(), pk.size(), hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_qm, qm.data(), qm.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
        continue;
      }
      if (c->fp16) {
        if (ti.f16 != (op.cin != 1)) return fail(c, SPVO_ERR_IO, "op %u: FP16 engine: a %d-channel input tensor stored as %s", i, op.cin, ti.f16 ? "fp16" : "fp32");
        if (op.cin == 1) {   // fp32 plane in: fp32 arithmetic on fp16-rounded weights, fp16 values out
          if (pool || add || (to.f16 && (op.out_c_off % 8))) return fail(c, SPVO_ERR_IO, "op %u: unsupported single-channel-input layer for FP16", i);
          std::vector<float> wr((size_t)op.cout * taps);
          for (size_t q = 0; q < wr.size(); ++q) wr[q] = (float)(_Float16)w[q];
          int rc = dev_alloc(c, &op.d_w, wr.size(), false);
          if (rc) return rc;
          if ((rc = dev_alloc(c, &op.d_b, op.cout, false))) return rc;
          HIP_TRY(c, hipMemcpy(op.d_w, wr.data(), wr.size() * 4, hipMemcpyHostToDevice));
          HIP_TRY(c, hipMemcpy(op.d_b, b, (size_t)op.cout * 4, hipMemcpyHostToDevice));
          continue;
        }
        const int ckg = (op.ks == 3 || op.cin % 32) ? 2 : 4;   // 16 channels per chunk; 32 for 1x1 layers when they divide
        if (op.cin % (8 * ckg) || op.in_c_off % 8) return fail(c, SPVO_ERR_IO, "op %u: cin %d / channel offset %d do not fit the FP16 chunking (%d)", i, op.cin, op.in_c_off, 8 * ckg);
        if (to.f16 && ((op.cout % 8) || (op.out_c_off % 8))) return fail(c, SPVO_ERR_IO, "op %u: cout %d / channel offset %d are not multiples of 8", i, op.cout, op.out_c_off);
        if ((bn || add) && !to.f16) return fail(c, SPVO_ERR_IO, "op %u: BatchNorm / residual epilogue with an fp32 output", i);
        if (add && (!c->tensors[op.residual].f16 || c->tensors[op.residual].ch != op.cout)) return fail(c, SPVO_ERR_IO, "op %u: residual tensor is not a %d-channel fp16 tensor", i, op.cout);
        if (!to.f16 && pool) return fail(c, SPVO_ERR_IO, "op %u: pooled fp32 output", i);
        op.ck = 8 * ckg;
        op.n_chunks = op.cin / op.ck;
        op.co_tiles = (op.cout + CO_TILE - 1) / CO_TILE;
        int ck_unused;
        choose_variant(op.ks, ti.H, ti.W, op.co_tiles, c->cfg.max_batch, pool, c->num_cus, &op.wr, &op.wc, &ck_unused);
        const std::vector<_Float16> pk = pack_conv_weights_f16(w, b, op.cout, op.cin, op.ks, ckg);
        int rc = dev_alloc(c, &op.d_w16, pk.size(), false);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpy(op.d_w16, pk.data(), pk.size() * sizeof(_Float16), hipMemcpyHostToDevice));
        continue;
      }
      if (c->s3) {
        if (bn || add) return fail(c, SPVO_ERR_IO, "op %u: split-fp32 mode has no BatchNorm / residual epilogue", i);
        if (ti.s3 != (op.cin != 1)) return fail(c, SPVO_ERR_IO, "op %u: split-fp32 mode: unexpected storage of the input tensor", i);
        if (to.s3 && ((op.cout % 8) || (op.out_c_off % 8))) return fail(c, SPVO_ERR_IO, "op %u: cout %d / channel offset %d are not multiples of 8", i, op.cout, op.out_c_off);
        if (!to.s3 && pool) return fail(c, SPVO_ERR_IO, "op %u: pooled fp32 output", i);
        if (op.cin == 1) {
          if (pool || !to.s3) return fail(c, SPVO_ERR_IO, "op %u: unsupported single-channel-input layer for split-fp32 mode", i);
          int rc = dev_alloc(c, &op.d_w, (size_t)op.cout * taps, false);
          if (rc) return rc;
          if ((rc = dev_alloc(c, &op.d_b, op.cout, false))) return rc;
          HIP_TRY(c, hipMemcpy(op.d_w, w, (size_t)op.cout * taps * 4, hipMemcpyHostToDevice));
          HIP_TRY(c, hipMemcpy(op.d_b, b, (size_t)op.cout * 4, hipMemcpyHostToDevice));
          continue;
        }
        const int ckg = op.ks == 3 ? 1 : 2;
        if (op.cin % (8 * ckg) || op.in_c_off % 8) return fail(c, SPVO_ERR_IO, "op %u: cin %d / channel offset %d do not fit the split-fp32 chunking (%d)", i, op.cin, op.in_c_off, 8 * ckg);
        op.ck = 8 * ckg;
        op.n_chunks = op.cin / op.ck;
        op.co_tiles = (op.cout + CO_TILE - 1) / CO_TILE;
        int ck_unused;
        choose_variant(op.ks, ti.H, ti.W, op.co_tiles, c->cfg.max_batch, pool, c->num_cus, &op.wr, &op.wc, &ck_unused, true);
        if (const char *force = std::getenv("SPVO_CONV_FORCE")) {   // tuning aid: "op:wr,wc,ck;op:wr,wc,ck" (ck ignored here)
          for (const char *q = force; q && *q; q = std::strchr(q, ';') ? std::strchr(q, ';') + 1 : nullptr) {
            int oi, wr, wc, ck;
            if (std::sscanf(q, "%d:%d,%d,%d", &oi, &wr, &wc, &ck) == 4 && oi == (int)i) { op.wr = wr; op.wc = wc; }
          }
        }
        const std::vector<unsigned short> pk = pack_conv_weights_s3(w, b, op.cout, op.cin, op.ks, ckg);
        int rc = dev_alloc(c, &op.d_ws3, pk.size(), false);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpy(op.d_ws3, pk.data(), pk.size() * sizeof(unsigned short), hipMemcpyHostToDevice));
        continue;
      }
      if (op.cin == 1) {
        if (pool || add) return fail(c, SPVO_ERR_IO, "op %u: single-channel-input layers have no pooling / residual form", i);
        int rc = dev_alloc(c, &op.d_w, (size_t)op.cout * taps, false);
        if (rc) return rc;
        if ((rc = dev_alloc(c, &op.d_b, op.cout, false))) return rc;
        HIP_TRY(c, hipMemcpy(op.d_w, w, (size_t)op.cout * taps * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_b, b, (size_t)op.cout * 4, hipMemcpyHostToDevice));
        continue;
      }
      op.co_tiles = (op.cout + CO_TILE - 1) / CO_TILE;
      choose_variant(op.ks, ti.H, ti.W, op.co_tiles, c->cfg.max_batch, pool, c->num_cus, &op.wr, &op.wc, &op.ck);
      if (const char *force = std::getenv("SPVO_CONV_FORCE")) {   // tuning aid: "op:wr,wc,ck;op:wr,wc,ck"
        for (const char *q = force; q && *q; q = std::strchr(q, ';') ? std::strchr(q, ';') + 1 : nullptr) {
          int oi, wr, wc, ck;
          if (std::sscanf(q, "%d:%d,%d,%d", &oi, &wr, &wc, &ck) == 4 && oi == (int)i) { op.wr = wr; op.wc = wc; op.ck = ck; }
        }
      }
      // Winograd F(2x2,3x3) for the plain 3x3 layers (no BatchNorm / residual epilogue) whose 8x32 tiles give at least
      // SPVO_WINOGRAD_MIN_TILES workgroups (default: 3/4 of the CUs; below that -- conv4a/4b at 45x147: 120 -- the direct kernel's 4x32 tiles fill the chip better: 47 vs 52 us); a pooled layer needs even sizes (the pooling window is
      // the Winograd tile).  SPVO_WINOGRAD=0 switches it off (A/B measurements, parity debugging).
      {
        const bool wino_on = !(std::getenv("SPVO_WINOGRAD") && std::atoi(std::getenv("SPVO_WINOGRAD")) == 0);
        const long wtiles = (long)((ti.W + WinoTile::TW - 1) / WinoTile::TW) * ((ti.H + WinoTile::TH - 1) / WinoTile::TH) * op.co_tiles * c->cfg.max_batch;
        const long min_tiles = std::getenv("SPVO_WINOGRAD_MIN_TILES") ? std::atol(std::getenv("SPVO_WINOGRAD_MIN_TILES")) : 3 * c->num_cus / 4;
        const bool eligible = wino_on && op.ks == 3 && !bn && !add && (op.cin % WinoTile::CK) == 0 && (!pool || ((ti.H % 2) == 0 && (ti.W % 2) == 0));
        op.wino2 = !(std::getenv("SPVO_WINO2") && std::atoi(std::getenv("SPVO_WINO2")) == 0);
        op.wino = eligible && wtiles >= min_tiles;
        // too few 64-channel tiles (conv4a / conv4b at 45x147: 120 on 256 CUs): 32 channels per workgroup fill the chip, and the
        // layer -- one tile's chain of items per workgroup -- becomes a chain of half-size items (SPVO_WINO_NARROW=0: direct kernel)
        const bool narrow_on = !(std::getenv("SPVO_WINO_NARROW") && std::atoi(std::getenv("SPVO_WINO_NARROW")) == 0);
        if (eligible && !op.wino && op.wino2 && narrow_on && (op.cout % 32) == 0 && 2 * wtiles >= min_tiles) op.wino = op.wino_narrow = true;
      }
      if (op.wino) {
        op.ck = WinoTile::CK;
        op.n_chunks = op.cin / op.ck;
        if (op.wino_narrow) op.co_tiles = op.cout / 32;
        const std::vector<float> pk = op.wino2 ? pack_conv_weights_wino2(w, b, op.cout, op.cin, op.wino_narrow ? 32 : CO_TILE) : pack_conv_weights_wino(w, b, op.cout, op.cin);
        int rc = dev_alloc(c, &op.d_w, pk.size(), false);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpy(op.d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        if (op.wino2 && !(std::getenv("SPVO_WINO_DYNAMIC") && std::atoi(std::getenv("SPVO_WINO_DYNAMIC")) == 0) && (rc = dev_alloc(c, &op.d_sched, 16))) return rc;
        continue;
      }
      if (op.cin % op.ck) return fail(c, SPVO_ERR_IO, "op %u: cin %d is not a multiple of %d", i, op.cin, op.ck);
      op.n_chunks = op.cin / op.ck;
      // repack OIHW + bias -> [co_tile][chunk][(tap, ci) rows + bias row][64]
      const std::vector<float> pk = pack_conv_weights(w, b, op.cout, op.cin, op.ks, op.ck);
      std::vector<float> bp((size_t)op.co_tiles * CO_TILE, 0.f);
      for (int o = 0; o < op.cout; ++o) bp[o] = b[o];
      int rc = dev_alloc(c, &op.d_w, pk.size(), false);
      if (rc) return rc;
      if ((rc = dev_alloc(c, &op.d_b, bp.size(), false))) return rc;
      HIP_TRY(c, hipMemcpy(op.d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(op.d_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
    } else if (op.type == OP_MAXPOOL) {
      std::snprintf(name, sizeof name, "pool:%u", i);
      op.stage = stage_id(c, name);
      if (to.level != ti.level + 1 || to.ch != ti.ch) return fail(c, SPVO_ERR_IO, "op %u: bad pool", i);
    } else if (op.type == OP_L2NORM) {
      std::snprintf(name, sizeof name, "l2norm:%u", i);
      op.stage = stage_id(c, name);
      if (ti.ch != 256 || to.ch != 256 || ti.level != 3) return fail(c, SPVO_ERR_IO, "op %u: descriptor tail must be 256 channels at 1/8", i);
    } else {
      return fail(c, SPVO_ERR_IO, "op %u: unknown type %d", i, op.type);
    }
  }
  {   // mark the dominant layer
    Op *best = nullptr;
    for (auto &o : c->ops) if (o.type == OP_CONV && (!best || o.flops_per_image > best->flops_per_image)) best = &o;
    if (best) best->dominant = true;
  }
  const Tensor &td = c->tensors[c->t_det];
  const Tensor &ts = c->tensors[c->t_desc];
  if (td.ch != 65 || td.level != 3 || td.nhwc || !ts.nhwc || c->tensors[c->t_input].ch != 1 || c->tensors[c->t_input].level != 0)
    return fail(c, SPVO_ERR_IO, "%s: unexpected output tensors", path);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  c->weights = true;
  return SPVO_OK;
}

int spvo_set_fp32_split(spvo_ctx *c, int enable) {
  if (!c) return SPVO_ERR_INVALID;
  c->split_req = enable != 0;
  return SPVO_OK;
}

int spvo_engine_precision(const spvo_ctx *c) {
  if (!c || !c->weights) return SPVO_ERR_STATE;
  return c->int8 ? 2 : c->fp16 ? 1 : 0;
}

int spvo_preprocess(spvo_ctx *c, const uint8_t *img, int rows, int cols, size_t stride, double P[12], uint8_t *resized_u8) {
  if (!c || !img || !P || rows <= 0 || cols <= 0 || stride < (size_t)cols) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (!c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "detector submissions are in flight: complete them with spvo_detect_wait first");
  if (!c->weights) return fail(c, SPVO_ERR_STATE, "no weights loaded");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const size_t bytes = (size_t)rows * stride;
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
  if (t.i8) hipLaunchKernelGGL(unpad_c16_kernel, dim3((t.W + 63) / 64, t.H, batch * t.ch), dim3(64), 0, c->stream, (const int8_t *)t.d, tmp, t.ch, t.H, t.W, t.hp, t.wp);
  else if (t.s3) hipLaunchKernelGGL(unpad_s3_kernel, dim3((t.W + 63) / 64, t.H, batch * t.ch), dim3(64), 0, c->stream, (const unsigned short *)t.d, tmp, t.ch, t.H, t.W, t.hp, t.wp);
  else if (t.f16) hipLaunchKernelGGL(unpad_c8_kernel, dim3((t.W + 63) / 64, t.H, batch * t.ch), dim3(64), 0, c->stream, (const _Float16 *)t.d, tmp, t.ch, t.H, t.W, t.hp, t.wp);
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
  sj.j[0] = SampleJob{ts.d, c->d_xy_tmp, nullptr, n, c->d_desc_tmp, nullptr, nullptr, nullptr, nullptr};
  sj.j[1] = sj.j[0];
  hipLaunchKernelGGL(sample_desc_kernel, dim3((n + 3) / 4, 1), dim3(256), 0, c->stream, sj, c->H, c->W, c->Hc, c->Wc);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(out, c->d_desc_tmp, (size_t)n * 256 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SPVO_OK;
}

static int enqueue_sample(spvo_ctx *c, const int slots[2], const NmsPair &np, int ring) {
  const Tensor &ts = c->tensors[c->t_desc];
  const float *desc = ts.dr[ring] ? ts.dr[ring] : ts.d;
  ScopedStage ss(c, stage_id(c, "sample"));
  const int cap = c->cfg.max_keypoints;
  float *stage = c->d_xy_stage + (size_t)ring * 2 * cap * 2;
  SampleJobs sj;
  for (int i = 0; i < 2; ++i) {
    FeatureSlot &s = c->slots[slots[i]];
    // the keypoint count is read from the NMS counters on the device: no host round trip
    sj.j[i] = SampleJob{desc + (size_t)i * ts.per_image, np.b[i].out_xy, (const int *)(np.b[i].counters + 2), 0, s.d_desc, s.d_sqn,
                        stage + (size_t)i * cap * 2, s.d_xy, s.d_n};
  }
  hipLaunchKernelGGL(sample_desc_kernel, dim3((cap + 3) / 4, 2), dim3(256), 0, c->post, sj, c->H, c->W, c->Hc, c->Wc);
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(c->h_xy_r[ring], stage, (size_t)2 * cap * 2 * sizeof(float), hipMemcpyDeviceToHost, c->post));
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
// their copies to pinned memory) on `stream_t` behind an event.  Up to MAX_INFLIGHT submissions may
// be queued: the tail of one overlaps with the network of the next, whose kernels leave CUs idle at
// their ragged ends.  Every buffer a tail touches belongs to the submission's set (RING of them), so
// a later submission -- or the rare host-driven NMS redo of an earlier one -- never meets it.
static int ensure_host_sets(spvo_ctx *c, size_t image_bytes);

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
  const Tensor &td = c->tensors[c->t_det];
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
  if (host_l) {   // pageable -> pinned (host copy), pinned -> device (DMA on the network stream): the caller's buffers are free on return
    const size_t bytes = (size_t)rows * stride;
    std::memcpy(c->h_img_r[ring], host_l, bytes);
    std::memcpy(c->h_img_r[ring] + c->img_cap_r, host_r, bytes);
    HIP_TRY(c, hipMemcpyAsync(c->d_img_r[ring], c->h_img_r[ring], bytes, hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_img_r[ring] + c->img_cap_r, c->h_img_r[ring] + c->img_cap_r, bytes, hipMemcpyHostToDevice, c->stream));
    srcs[0] = c->d_img_r[ring];
    srcs[1] = c->d_img_r[ring] + c->img_cap_r;
  }
  // ---- everything below is enqueued without a host round trip
  c->cur_ring = ring;
  c->post = c->stream;
  // SPVO_TRUNK_TIMING=1 (diagnostic): how long the network stream works per submission and how long it stands idle between two
  // submissions, from timing events at both ends of the trunk (printed every 200 submissions)
  static const bool trunk_timing = std::getenv("SPVO_TRUNK_TIMING") != nullptr;
  constexpr int TT = 8;   // ring of timing events: deeper than the submissions that can be in flight
  static hipEvent_t tt_b[TT], tt_e[TT];
  static long tt_n = 0;
  static double tt_busy = 0, tt_idle = 0;
  if (trunk_timing) {
    const double tnow = diag_now_us();
    if (c->submit_count > 1 && hipEventQuery(c->ev_net[(c->submit_count - 2) % RING]) == hipSuccess) ++g_diag.late;   // the trunk before this one is done already: the stream is idle
    g_diag.depth_sum += (int)c->pendq.size();
    if (g_diag.t_last_submit > 0) g_diag.max_interval = std::max(g_diag.max_interval, tnow - g_diag.t_last_submit);
    g_diag.t_last_submit = tnow;
    if (tt_n == 0)
      for (int r = 0; r < TT; ++r) { (void)hipEventCreate(&tt_b[r]); (void)hipEventCreate(&tt_e[r]); }
    if (tt_n >= TT) {   // the submissions before those that may be in flight are complete: ring slots (n-4) and (n-5)
      const int r2 = (int)((tt_n - 4) % TT), r3 = (int)((tt_n - 5) % TT);
      float busy = 0, idle = 0;
      static int tt_late = 0;
      static float tt_max = 0;
      if (hipEventElapsedTime(&busy, tt_b[r2], tt_e[r2]) == hipSuccess && hipEventElapsedTime(&idle, tt_e[r3], tt_b[r2]) == hipSuccess) {
        tt_busy += busy; tt_idle += idle;
        tt_late += idle > 0.05f ? 1 : 0;
        tt_max = std::max(tt_max, idle);
      }
      if (tt_n % 200 == 0) {
        std::fprintf(stderr, "[spvo] trunk timing over 200 submissions: network stream busy %.1f us, idle %.1f us per submission (%d gaps above 50 us, longest %.0f us)\n",
                     tt_busy * 1e3 / 200, tt_idle * 1e3 / 200, tt_late, tt_max * 1e3);
        std::fprintf(stderr, "[spvo]   host: longest interval between submissions %.0f us, longest wait for a tail %.0f us, for a solve %.0f us, matches not served from the cache %d; "
                             "submissions that found the network stream idle %d, mean submissions in flight at submit %.2f\n",
                     g_diag.max_interval, g_diag.max_tail_wait, g_diag.max_solve_wait, g_diag.match_miss, g_diag.late, g_diag.depth_sum / 200.0);
        g_diag.max_interval = g_diag.max_tail_wait = g_diag.max_solve_wait = 0; g_diag.match_miss = 0; g_diag.late = 0; g_diag.depth_sum = 0;
        tt_busy = tt_idle = 0; tt_late = 0; tt_max = 0;
      }
    }
    (void)hipEventRecord(tt_b[tt_n % TT], c->stream);
  }
  hipEvent_t det_e0 = nullptr;
  const bool prof_detect = c->prof && (c->prof_only < 0 || c->prof_only == stage_id(c, "detect"));
  if (prof_detect) { det_e0 = get_event(c); (void)hipEventRecord(det_e0, c->stream); }
  {
    ScopedStage sp(c, stage_id(c, "preprocess"));
    int rc = launch_preprocess(c, srcs[0], srcs[1], 2, rows, cols, stride, g, 0, (extras & 1) ? c->d_resized_r[ring] : nullptr);
    if (rc) return rc;
  }
  int rc;
  {
    ScopedStage net(c, stage_id(c, "net"));
    rc = run_ops(c, 2, 0, c->head_start, c->stream);
  }
  if (rc) { c->cur_ring = 0; return rc; }
  c->last_batch = 2;
  if (trunk_timing) { (void)hipEventRecord(tt_e[tt_n % 8], c->stream); ++tt_n; }
  HIP_TRY(c, hipEventRecord(c->ev_net[ring], c->stream));
  HIP_TRY(c, hipStreamWaitEvent(c->stream_t, c->ev_net[ring], 0));
  c->post = c->stream_t;
  rc = run_ops(c, 2, c->head_start, c->ops.size(), c->stream_t);   // heads: on the tail stream, reading this submission's ring buffers
  c->cur_ring = 0;
  if (rc) { c->post = c->stream; return rc; }
  const NmsPair np = nms_pair(c, ring);
  {
    // heat map + threshold + candidate list in one kernel; the counter block of this set was
    // zeroed by the previous submission's last NMS kernel (or at allocation)
    ScopedStage sh(c, stage_id(c, "heatmap"));
    hipLaunchKernelGGL(heatmap_nms_kernel, dim3((c->Wc + 63) / 64, (c->Hc + 3) / 4, 2), dim3(256), 0, c->post, td.dr[ring], c->d_heat_r[ring], c->Hc, c->Wc, td.hp, td.wp,
                       c->cfg.conf_thresh, np);
    HIP_TRY(c, hipGetLastError());
  }
  {
    ScopedStage sn(c, stage_id(c, "nms"));
    rc = launch_nms_rounds(c, 2, np, ring, NMS_FIRST, c->d_counters_all + (size_t)(((ring + 1) % RING) * 2) * NMS_COUNTER_INTS);
  }
  if (!rc) rc = enqueue_sample(c, slots, np, ring);
  if (!rc && c->prematch) rc = enqueue_prematch(c, slot_l, slot_r, prev_l, ring);
  if (!rc && (extras & 1))   // resized images (what nn.cpp:154 pushes to images_dq) -> the set's pinned mirror
    rc = hipMemcpyAsync(c->h_resized_r[ring], c->d_resized_r[ring], (size_t)2 * c->H * c->W, hipMemcpyDeviceToHost, c->stream_t) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "hipMemcpyAsync failed");
  if (!rc && (extras & 2)) {   // descriptors of both images: whole slots (the counts are not known on the host yet; rows >= n are stale)
    const size_t per = (size_t)c->cfg.max_keypoints * 256;
    for (int i = 0; i < 2 && !rc; ++i)
      rc = hipMemcpyAsync(c->h_desc_r[ring] + i * per, c->slots[slots[i]].d_desc, per * sizeof(float), hipMemcpyDeviceToHost, c->stream_t) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "hipMemcpyAsync failed");
  }
  if (!rc && prof_detect) {   // "detect" spans both streams: first kernel on `stream` .. last copy on `stream_t`
    hipEvent_t e1 = get_event(c);
    (void)hipEventRecord(e1, c->stream_t);
    c->pending.push_back({stage_id(c, "detect"), det_e0, e1});
  }
  if (!rc) rc = (hipEventRecord(c->ev_tail[ring], c->stream_t) == hipSuccess) ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "hipEventRecord failed");
  c->post = c->stream;
  if (rc) return rc;
  for (int i = 0; i < 2; ++i) c->slots[slots[i]].filled = true;
  c->last_slot_l = slot_l;
  PendingDetect pd;
  pd.g = CropGeomS{g.row_off, g.col_off, g.crop_rows, g.crop_cols, g.scale};
  pd.rows = rows; pd.cols = cols;
  pd.slot_l = slot_l; pd.slot_r = slot_r; pd.prev_l = prev_l; pd.ring = ring; pd.extras = extras;
  c->pendq.push_back(pd);
  return SPVO_OK;
}

// completes the OLDEST submission
static int detect_wait(spvo_ctx *c, double P_l[12], double P_r[12], spvo_features *out_l, spvo_features *out_r, uint8_t *resized_l, uint8_t *resized_r) {
  if (c->pendq.empty()) return fail(c, SPVO_ERR_STATE, "no detector submission in flight");
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
  c->post = c->stream_t;
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
  auto restage = [&]() -> int {   // after an NMS redo the set's mirrors are refreshed too
    if (pd.extras & 2)
      for (int i = 0; i < 2; ++i)
        HIP_TRY(c, hipMemcpyAsync(c->h_desc_r[pd.ring] + (size_t)i * cap * 256, c->slots[slots[i]].d_desc, (size_t)cap * 256 * sizeof(float), hipMemcpyDeviceToHost, c->post));
    return SPVO_OK;
  };
  int rc = SPVO_OK;
  if (extras) {
    if ((rc = copy_extras())) { c->post = c->stream; return rc; }
    rc = hipStreamSynchronize(c->stream_t) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "stream synchronisation failed");
  } else {
    // only this submission's tail: a younger one may be queued behind it on both streams
    const double tw0 = diag_now_us();
    rc = wait_event(c->ev_tail[pd.ring]) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "event synchronisation failed");
    g_diag.max_tail_wait = std::max(g_diag.max_tail_wait, diag_now_us() - tw0);
  }
  bool redone = false;
  const NmsPair np = nms_pair(c, pd.ring);
  if (!rc) rc = nms_settle(c, 2, np, pd.ring, &redone);
  if (!rc && (redone || pd.rematch)) {   // rare: keypoints changed after the first batch -> redo what depends on them
    if (redone) c->stages[stage_id(c, "nms_redo")].calls += 1;       // counted even with profiling off (tests, diagnostics)
    if (pd.rematch) c->stages[stage_id(c, "rematch")].calls += 1;
    if (redone) rc = enqueue_sample(c, slots, np, pd.ring);
    if (!rc && c->prematch) rc = enqueue_prematch(c, pd.slot_l, pd.slot_r, pd.prev_l, pd.ring);
    if (!rc && extras) rc = copy_extras();
    if (!rc && redone) rc = restage();
    if (!rc) rc = hipStreamSynchronize(c->stream_t) == hipSuccess ? SPVO_OK : fail(c, SPVO_ERR_DEVICE, "stream synchronisation failed");
    if (redone)
      for (auto &q : c->pendq)
        if (q.prev_l == pd.slot_l) q.rematch = true;   // it matched against keypoints that have just been replaced
  }
  c->post = c->stream;
  if (rc) return rc;
  const int *hc = c->h_counters_r[pd.ring];
  for (int i = 0; i < 2; ++i) {
    FeatureSlot &s = c->slots[slots[i]];
    s.n = hc[i * NMS_COUNTER_INTS + 2];
    s.gen += 1;
    if (outs[i]) {
      outs[i]->n = s.n;
      if (outs[i]->xy && s.n > 0) std::memcpy(outs[i]->xy, c->h_xy_r[pd.ring] + (size_t)i * cap * 2, (size_t)s.n * 2 * sizeof(float));
      if (outs[i]->desc && (pd.extras & 2) && s.n > 0) std::memcpy(outs[i]->desc, c->h_desc_r[pd.ring] + (size_t)i * cap * 256, (size_t)s.n * 256 * sizeof(float));
    }
    if (res[i] && (pd.extras & 1)) std::memcpy(res[i], c->h_resized_r[pd.ring] + (size_t)i * c->H * c->W, (size_t)c->H * c->W);
  }
  for (auto &mc : c->mcache[pd.ring])
    if (mc.valid) { mc.gen_a = c->slots[mc.slot_a].gen; mc.gen_b = c->slots[mc.slot_b].gen; }
  const CropGeom g{pd.g.row_off, pd.g.col_off, pd.g.crop_rows, pd.g.crop_cols, pd.g.scale};
  if (P_l) fix_projection(P_l, g, pd.rows, pd.cols, c->cfg.bug_compat_p);
  if (P_r) fix_projection(P_r, g, pd.rows, pd.cols, c->cfg.bug_compat_p);
  return SPVO_OK;
}

// buffers of the host-image submissions: allocated on first use, grown when a larger image arrives (never while submissions are in flight)
static int ensure_host_sets(spvo_ctx *c, size_t image_bytes) {
  const size_t hw2 = (size_t)2 * c->H * c->W, desc = (size_t)2 * c->cfg.max_keypoints * 256;
  if (!c->d_resized_r[0]) {
    for (int r = 0; r < RING; ++r) {
      int rc = dev_alloc(c, &c->d_resized_r[r], hw2, false);
      if (rc) return rc;
      HIP_TRY(c, hipHostMalloc((void **)&c->h_resized_r[r], hw2));
      HIP_TRY(c, hipHostMalloc((void **)&c->h_desc_r[r], desc * sizeof(float)));
    }
  }
  if (image_bytes > c->img_cap_r) {
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
  const size_t bytes = (size_t)rows * stride;
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

int spvo_match(spvo_ctx *c, const float *desc_a, int na, const float *desc_b, int nb, int selector, int cross_check, float ratio, int32_t *train_idx, float *distance) {
  if (!c || na < 0 || nb < 0 || (na > 0 && (!desc_a || !train_idx || !distance)) || (nb > 0 && !desc_b)) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (selector != SPVO_SELECT_NN && selector != SPVO_SELECT_KNN) return fail(c, SPVO_ERR_INVALID, "bad selector");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc = ensure_match(c, na, nb);
  if (rc) return rc;
  PostScope ps(c);   // behind the queued tails: they share the matcher's scratch
  if (na) HIP_TRY(c, hipMemcpyAsync(c->d_ma, desc_a, (size_t)na * MATCH_D * sizeof(float), hipMemcpyHostToDevice, c->post));
  if (nb) HIP_TRY(c, hipMemcpyAsync(c->d_mb, desc_b, (size_t)nb * MATCH_D * sizeof(float), hipMemcpyHostToDevice, c->post));
  return run_match(c, MatchReq{c->d_ma, c->d_mb, na, nb, nullptr, nullptr, nullptr, nullptr}, selector, cross_check ? 1 : 0, ratio, train_idx, distance);
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
  const size_t px0 = (size_t)rows * cols;
  const int surv_cap = (rows / 2 + 1) * (cols / 2 + 1);   // 3x3 suppression: at most one survivor per 2x2 block
  const int kp_cap = nfeatures;
  if (px0 > o.px_cap || kp_cap > o.kp_cap) {
    HIP_TRY(c, hipStreamSynchronize(st));
    for (void *p : {(void *)o.im, (void *)o.score, (void *)o.blur, (void *)o.tmp, (void *)o.keys, (void *)o.rank, (void *)o.out_xy, (void *)o.counters, (void *)o.tab,
                    (void *)o.kps, (void *)o.desc}) if (p) (void)hipFree(p);
    o.im = o.score = o.blur = nullptr; o.tmp = nullptr; o.keys = nullptr; o.rank = o.out_xy = o.counters = o.tab = nullptr; o.kps = nullptr; o.desc = nullptr;
    o.px_cap = 0; o.kp_cap = 0;
    int rc;
    const size_t pyr = off[ORB_LEVELS] + 256, kall = (size_t)5 * surv_cap;   // all levels side by side (sum of 1 / 1.44^l < 3.3)
    if ((rc = dev_alloc(c, &o.im, pyr)) || (rc = dev_alloc(c, &o.score, pyr)) || (rc = dev_alloc(c, &o.blur, pyr)) || (rc = dev_alloc(c, &o.tmp, pyr)) ||
        (rc = dev_alloc(c, &o.keys, kall)) || (rc = dev_alloc(c, &o.rank, kall)) || (rc = dev_alloc(c, &o.out_xy, 2 * kall)) ||
        (rc = dev_alloc(c, &o.counters, (size_t)ORB_LEVELS * NMS_COUNTER_INTS)) || (rc = dev_alloc(c, &o.tab, (size_t)16 * (rows + cols))) || (rc = dev_alloc(c, &o.kps, kp_cap)) ||
        (rc = dev_alloc(c, &o.desc, (size_t)kp_cap * 32)))
      return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (dev_alloc clears on the network stream)
    o.px_cap = px0; o.kp_cap = kp_cap;
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
  if ((size_t)rows * stride > o.src_cap) {
    HIP_TRY(c, hipStreamSynchronize(st));
    if (o.src) (void)hipFree(o.src);
    o.src = nullptr; o.src_cap = 0;
    int rc = dev_alloc(c, &o.src, (size_t)rows * stride, false);
    if (rc) return rc;
    o.src_cap = (size_t)rows * stride;
  }
  HIP_TRY(c, hipMemcpyAsync(o.src, img, (size_t)rows * stride, hipMemcpyHostToDevice, st));
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

// cv::BFMatcher(NORM_HAMMING): binary descriptors of `desc_bytes` bytes per row (ORB 32, BRISK 64, AKAZE 61), see match.hip.h K12h
int spvo_match_hamming(spvo_ctx *c, const uint8_t *desc_a, int na, const uint8_t *desc_b, int nb, int desc_bytes, int selector, int cross_check, float ratio,
                       int32_t *train_idx, float *distance) {
  if (!c || na < 0 || nb < 0 || (na > 0 && (!desc_a || !train_idx || !distance)) || (nb > 0 && !desc_b)) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (selector != SPVO_SELECT_NN && selector != SPVO_SELECT_KNN) return fail(c, SPVO_ERR_INVALID, "bad selector");
  if (desc_bytes <= 0 || desc_bytes > 64) return fail(c, SPVO_ERR_INVALID, "binary descriptors of 1 .. 64 bytes are supported (got %d)", desc_bytes);
  if (na == 0) return SPVO_OK;
  if (nb == 0) {
    for (int i = 0; i < na; ++i) { train_idx[i] = -1; distance[i] = 0.f; }
    return SPVO_OK;
  }
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  const int nw = desc_bytes <= 32 ? 8 : 16;
  const int need = std::max(na, nb);
  if (need > c->ham_cap) {
    HIP_TRY(c, hipStreamSynchronize(c->stream2));
    for (void *p : {(void *)c->d_ham_a, (void *)c->d_ham_b, (void *)c->d_ham_idx, (void *)c->d_ham_dist, (void *)c->d_ham_vote}) if (p) (void)hipFree(p);
    c->d_ham_a = c->d_ham_b = nullptr; c->d_ham_idx = nullptr; c->d_ham_dist = nullptr; c->d_ham_vote = nullptr;
    c->ham_cap = 0;
    const int cap = std::max(need, 2048);
    int rc;
    if ((rc = dev_alloc(c, &c->d_ham_a, (size_t)cap * 16)) || (rc = dev_alloc(c, &c->d_ham_b, (size_t)cap * 16)) || (rc = dev_alloc(c, &c->d_ham_idx, cap)) ||
        (rc = dev_alloc(c, &c->d_ham_dist, cap)) || (rc = dev_alloc(c, &c->d_ham_vote, cap)))
      return rc;
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // (dev_alloc clears on the network stream)
    c->ham_cap = cap;
  }
  // rows zero-padded to nw words: padding bits are equal on both sides and add nothing to a distance
  std::vector<uint32_t> pa((size_t)na * nw, 0u), pb((size_t)nb * nw, 0u);
  for (int i = 0; i < na; ++i) std::memcpy(&pa[(size_t)i * nw], desc_a + (size_t)i * desc_bytes, desc_bytes);
  for (int i = 0; i < nb; ++i) std::memcpy(&pb[(size_t)i * nw], desc_b + (size_t)i * desc_bytes, desc_bytes);
  hipStream_t st = c->stream2;   // the solver's stream: overlaps detector submissions in flight
  HIP_TRY(c, hipMemcpyAsync(c->d_ham_a, pa.data(), pa.size() * 4, hipMemcpyHostToDevice, st));
  HIP_TRY(c, hipMemcpyAsync(c->d_ham_b, pb.data(), pb.size() * 4, hipMemcpyHostToDevice, st));
  const bool cross = cross_check && selector == SPVO_SELECT_NN;   // BFMatcher's crossCheck is off for knnMatch (base.cpp:27-28)
  auto launch = [&](const uint32_t *A, int n_a, const uint32_t *B, int n_b, int mode) {
    if (nw == 8) hipLaunchKernelGGL(match_hamming_kernel<8>, dim3((n_a + 3) / 4), dim3(256), 0, st, A, n_a, B, n_b, mode, ratio, c->d_ham_idx, c->d_ham_dist, c->d_ham_vote);
    else hipLaunchKernelGGL(match_hamming_kernel<16>, dim3((n_a + 3) / 4), dim3(256), 0, st, A, n_a, B, n_b, mode, ratio, c->d_ham_idx, c->d_ham_dist, c->d_ham_vote);
  };
  if (cross) {
    HIP_TRY(c, hipMemsetAsync(c->d_ham_vote, 0xFF, (size_t)na * sizeof(unsigned long long), st));
    launch(c->d_ham_b, nb, c->d_ham_a, na, 2);   // every train row votes for its nearest query row
    hipLaunchKernelGGL(match_hamming_cross_kernel, dim3((na + 255) / 256), dim3(256), 0, st, c->d_ham_vote, na, c->d_ham_idx, c->d_ham_dist);
  } else {
    launch(c->d_ham_a, na, c->d_ham_b, nb, selector == SPVO_SELECT_KNN ? 1 : 0);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(train_idx, c->d_ham_idx, (size_t)na * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipMemcpyAsync(distance, c->d_ham_dist, (size_t)na * 4, hipMemcpyDeviceToHost, st));
  HIP_TRY(c, hipStreamSynchronize(st));
  return SPVO_OK;
}

int spvo_match_slots(spvo_ctx *c, int slot_a, int slot_b, int selector, int cross_check, float ratio, int32_t *train_idx, float *distance) {
  if (!c || slot_a < 0 || slot_a >= N_SLOTS || slot_b < 0 || slot_b >= N_SLOTS) return fail(c, SPVO_ERR_INVALID, "bad slot");
  if (selector != SPVO_SELECT_NN && selector != SPVO_SELECT_KNN) return fail(c, SPVO_ERR_INVALID, "bad selector");
  const FeatureSlot &a = c->slots[slot_a], &b = c->slots[slot_b];
  if (a.n > 0 && (!train_idx || !distance)) return fail(c, SPVO_ERR_INVALID, "null output");
  for (int set = 0; set < RING; ++set) {   // already computed alongside the detector (spvo_set_prematch)?
    bool inflight = false;
    for (const auto &q : c->pendq) inflight |= q.ring == set;
    if (inflight) continue;   // that set belongs to a submission in flight
    for (const auto &mc : c->mcache[set])
      if (mc.valid && mc.slot_a == slot_a && mc.slot_b == slot_b && mc.gen_a == a.gen && mc.gen_b == b.gen && mc.selector == selector &&
          mc.cross == (cross_check ? 1 : 0) && mc.ratio == ratio) {
        if (a.n > 0) unpack_match(mc.h_out, a.n, train_idx, distance);
        return SPVO_OK;
      }
  }
  ++g_diag.match_miss;
  for (const auto &q : c->pendq)
    if (q.slot_l == slot_a || q.slot_r == slot_a || q.slot_l == slot_b || q.slot_r == slot_b)
      return fail(c, SPVO_ERR_STATE, "match not precomputed and a detector submission is rewriting the feature slots");
  PostScope ps(c);   // behind the queued tails: they share the matcher's scratch
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  int rc = ensure_match(c, a.n, b.n);
  if (rc) return rc;
  return run_match(c, MatchReq{a.d_desc, b.d_desc, a.n, b.n, nullptr, nullptr, a.d_sqn, b.d_sqn}, selector, cross_check ? 1 : 0, ratio, train_idx, distance);
}

int spvo_set_prematch(spvo_ctx *c, int enable, int selector, int cross_check, float ratio) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  if (selector != SPVO_SELECT_NN && selector != SPVO_SELECT_KNN) return fail(c, SPVO_ERR_INVALID, "bad selector");
  c->prematch = enable != 0;
  c->pm_selector = selector;
  c->pm_cross = cross_check ? 1 : 0;
  c->pm_ratio = ratio;
  for (auto &set : c->mcache)
    for (auto &mc : set) mc.valid = false;
  return SPVO_OK;
}

int spvo_set_match_fp8(spvo_ctx *c, int enable) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  c->match_fp8 = enable != 0;
  for (auto &set : c->mcache)
    for (auto &mc : set) mc.valid = false;
  return SPVO_OK;
}

int spvo_triangulate(spvo_ctx *c, const double P_l[12], const double P_r[12], const float *xy_l, const float *xy_r, int n, float *xyz) {
  if (!c || !P_l || !P_r || n < 0 || (n > 0 && (!xy_l || !xy_r || !xyz))) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (n == 0) return SPVO_OK;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  if (c->solve_pending.active) return fail(c, SPVO_ERR_STATE, "a solve is pending (spvo_solve_submit): complete it with spvo_solve_wait first");
  int rc = ensure_odometry(c, n, 0, 0);
  if (rc) return rc;
  HIP_TRY(c, hipMemcpyAsync(c->d_P, P_l, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_P + 12, P_r, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_pts_a, xy_l, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_pts_b, xy_r, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  {
    ScopedStage st(c, stage_id(c, "triangulate"));
    hipLaunchKernelGGL(triangulate_kernel, dim3((n + 63) / 64), dim3(64), 0, c->stream, c->d_P, c->d_P + 12, c->d_pts_a, c->d_pts_b, n, c->d_xyz);
  }
  HIP_TRY(c, hipGetLastError());
  HIP_TRY(c, hipMemcpyAsync(xyz, c->d_xyz, (size_t)n * 3 * sizeof(float), hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  return SPVO_OK;
}

int spvo_pnp_ransac(spvo_ctx *c, const double K[9], const float *xyz, const float *xy, int n, const spvo_ransac_opts *opts, double rvec[3], double tvec[3],
                    int32_t *inliers, int *n_inliers, int *ok) {
  if (!c || !K || !rvec || !tvec || !n_inliers || !ok || n < 0 || (n > 0 && (!xyz || !xy || !inliers))) return fail(c, SPVO_ERR_INVALID, "bad argument");
  spvo_ransac_opts o = {500, 2.0, 0.999, 0};
  if (opts) o = *opts;
  if (o.iterations <= 0 || o.iterations > 65536 || !(o.reproj_error > 0)) return fail(c, SPVO_ERR_INVALID, "bad RANSAC options");
  *ok = 0;
  *n_inliers = 0;
  if (n < 4) return SPVO_OK;  // not enough points for a model: prior is kept
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  if (c->solve_pending.active) return fail(c, SPVO_ERR_STATE, "a solve is pending (spvo_solve_submit): complete it with spvo_solve_wait first");
  int rc = ensure_odometry(c, n, o.iterations, 0);
  if (rc) return rc;
  double prior[6] = {rvec[0], rvec[1], rvec[2], tvec[0], tvec[1], tvec[2]};
  HIP_TRY(c, hipMemcpyAsync(c->d_P + 24, K, 9 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_P + 33, prior, 6 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_pts_a, xyz, (size_t)n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_pts_b, xy, (size_t)n * 2 * sizeof(float), hipMemcpyHostToDevice, c->stream));
  {
    ScopedStage st(c, stage_id(c, "ransac"));
    hipLaunchKernelGGL(ransac_hypothesis_kernel, dim3(o.iterations), dim3(64), 0, c->stream, c->d_P + 24, c->d_pts_a, c->d_pts_b, n, c->d_P + 33, o.seed, o.reproj_error * o.reproj_error, c->rw);
    hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(256), 0, c->stream, c->d_P + 24, c->d_pts_a, c->d_pts_b, n, c->d_P + 33, o.iterations, o.reproj_error * o.reproj_error, c->rw);
  }
  HIP_TRY(c, hipGetLastError());
  double res[8];
  HIP_TRY(c, hipMemcpyAsync(res, c->rw.result, sizeof res, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  *ok = res[6] != 0;
  *n_inliers = (int)res[7];
  for (int k = 0; k < 3; ++k) { rvec[k] = res[k]; tvec[k] = res[3 + k]; }
  if (*n_inliers > 0) HIP_TRY(c, hipMemcpy(inliers, c->rw.inliers, (size_t)(*n_inliers) * sizeof(int), hipMemcpyDeviceToHost));
  return SPVO_OK;
}

int spvo_pnp_refine(spvo_ctx *c, const double P_l[12], const double P_r[12], const spvo_obs *obs, int n_obs, const spvo_refine_opts *opts, double q[4], double t[3],
                    spvo_refine_summary *summary) {
  if (!c || !P_l || !P_r || !q || !t || n_obs < 0 || (n_obs > 0 && !obs)) return fail(c, SPVO_ERR_INVALID, "bad argument");
  spvo_refine_opts o = {40, 1.0};
  if (opts) o = *opts;
  if (o.max_iterations < 0 || !(o.huber_delta > 0)) return fail(c, SPVO_ERR_INVALID, "bad refine options");
  static_assert(sizeof(spvo_obs) == sizeof(ObsDev), "spvo_obs layout");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  if (c->solve_pending.active) return fail(c, SPVO_ERR_STATE, "a solve is pending (spvo_solve_submit): complete it with spvo_solve_wait first");
  int rc = ensure_odometry(c, 0, 0, n_obs);
  if (rc) return rc;
  double start[7] = {q[0], q[1], q[2], q[3], t[0], t[1], t[2]};
  HIP_TRY(c, hipMemcpyAsync(c->d_P, P_l, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_P + 12, P_r, 12 * sizeof(double), hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(c->d_P + 40, start, sizeof start, hipMemcpyHostToDevice, c->stream));
  if (n_obs) HIP_TRY(c, hipMemcpyAsync(c->d_obs, obs, (size_t)n_obs * sizeof(spvo_obs), hipMemcpyHostToDevice, c->stream));
  {
    ScopedStage st(c, stage_id(c, "refine"));
    hipLaunchKernelGGL(pnp_refine_kernel<512>, dim3(1), dim3(512), 0, c->stream, c->d_P, c->d_P + 12, c->d_obs, n_obs, (const int *)nullptr, c->d_P + 40, o.max_iterations, o.huber_delta, c->d_refine);
  }
  HIP_TRY(c, hipGetLastError());
  RefineOut r;
  HIP_TRY(c, hipMemcpyAsync(&r, c->d_refine, sizeof r, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  for (int k = 0; k < 4; ++k) q[k] = r.v[k];
  for (int k = 0; k < 3; ++k) t[k] = r.v[4 + k];
  if (summary) {
    summary->iterations = (int)r.v[7];
    summary->converged = (int)r.v[8];
    summary->usable = (int)r.v[9];
    summary->initial_cost = r.v[10];
    summary->final_cost = r.v[11];
  }
  return SPVO_OK;
}

// Everything of a solve up to the event behind its last copy; the inputs are staged in pinned memory, so the caller's arrays are
// free again when this returns.  What spvo_solve_wait needs later (the prior, n, the refinement degree) stays in the context.
int spvo_solve_submit(spvo_ctx *c, const spvo_solve_input *in) {
  if (!c || !in) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (c->solve_pending.active) return fail(c, SPVO_ERR_STATE, "a solve is pending: complete it with spvo_solve_wait first");
  const int n = in->n;
  if (n < 0 || (n > 0 && (!in->xy_cl || !in->xy_cr || !in->xy_pl || !in->xy_pr))) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (in->ransac.iterations <= 0 || in->ransac.iterations > 65536 || !(in->ransac.reproj_error > 0) || in->refine.max_iterations < 0 ||
      !(in->refine.huber_delta > 0))
    return fail(c, SPVO_ERR_INVALID, "bad solver options");
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  spvo_ctx::SolvePending pend;
  pend.n = n; pend.refinement_degree = in->refinement_degree;
  for (int k = 0; k < 3; ++k) { pend.rvec[k] = in->rvec_pred[k]; pend.tvec[k] = in->tvec_pred[k]; }
  if (n == 0) { pend.active = true; c->solve_pending = pend; return SPVO_OK; }   // nothing to enqueue: _wait answers with the prior
  // this call runs on the context's second stream so that it overlaps a detector submission in
  // flight; buffers only grow on first use (then everything is drained once)
  const bool grow = n > c->odo_cap || in->ransac.iterations > c->ransac_cap || 4 * n > c->obs_cap || !c->d_P || n > c->solve_cap;
  if (grow) HIP_TRY(c, hipDeviceSynchronize());
  int rc = ensure_odometry(c, n, in->ransac.iterations, 4 * n);
  if (rc) return rc;
  if (n > c->solve_cap) {
    const int cap = std::max(n, 2048);
    for (void *hp : {(void *)c->h_solve_in, (void *)c->h_solve_res, (void *)c->h_solve_o}) if (hp) (void)hipHostFree(hp);
    for (void *dp : {(void *)c->d_solve_in, (void *)c->d_solve_res, (void *)c->d_solve_o, (void *)c->d_ctl}) if (dp) (void)hipFree(dp);
    c->h_solve_in = c->h_solve_o = nullptr; c->h_solve_res = nullptr;
    c->d_solve_in = c->d_solve_o = nullptr; c->d_solve_res = nullptr; c->d_ctl = nullptr;
    c->solve_cap = 0;   // a failed allocation below leaves a context that spvo_destroy and a later call can still handle
    const size_t in_bytes = 64 * sizeof(double) + (size_t)12 * cap * 4, o_bytes = (size_t)4 * cap * 4;
    if ((rc = dev_alloc(c, &c->d_solve_in, in_bytes))) return rc;
    if ((rc = dev_alloc(c, &c->d_solve_res, 40))) return rc;
    if ((rc = dev_alloc(c, &c->d_solve_o, o_bytes))) return rc;
    if ((rc = dev_alloc(c, &c->d_ctl, 4))) return rc;
    HIP_TRY(c, hipHostMalloc((void **)&c->h_solve_in, in_bytes));
    HIP_TRY(c, hipHostMalloc((void **)&c->h_solve_res, 40 * sizeof(double)));
    HIP_TRY(c, hipHostMalloc((void **)&c->h_solve_o, o_bytes));
    c->solve_cap = cap;
  }
  if (grow) HIP_TRY(c, hipDeviceSynchronize());
  static const bool solve_timing = std::getenv("SPVO_SOLVE_TIMING") != nullptr;   // diagnostic: host time per phase of this call
  static double tacc[4] = {0, 0, 0, 0};
  static long tcalls = 0;
  auto now_us = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  const double tm0 = solve_timing ? now_us() : 0;
  // ---- pack: 64 doubles, then cl cr pl pr [2n each], prev_xyz [3n], prev_valid [n]
  double *hdr = (double *)c->h_solve_in;
  std::memset(hdr, 0, 64 * sizeof(double));
  for (int k = 0; k < 12; ++k) { hdr[k] = in->P_l[k]; hdr[12 + k] = in->P_r[k]; }
  const int kidx[9] = {0, 1, 2, 4, 5, 6, 8, 9, 10};
  for (int k = 0; k < 9; ++k) hdr[24 + k] = in->P_l[kidx[k]];                      // K = P_l[:, :3]  (base.cpp:227)
  for (int k = 0; k < 3; ++k) { hdr[33 + k] = in->rvec_pred[k]; hdr[36 + k] = in->tvec_pred[k]; }
  hdr[39] = in->frame_count; hdr[40] = in->refinement_degree;
  hdr[41] = 8.0; hdr[42] = 0.1; hdr[43] = 10;                                      // hpp:145-147
  float *fw = (float *)(c->h_solve_in + 64 * sizeof(double));
  std::memcpy(fw, in->xy_cl, (size_t)2 * n * 4);
  std::memcpy(fw + 2 * n, in->xy_cr, (size_t)2 * n * 4);
  std::memcpy(fw + 4 * n, in->xy_pl, (size_t)2 * n * 4);
  std::memcpy(fw + 6 * n, in->xy_pr, (size_t)2 * n * 4);
  const bool have_prev = in->prev_xyz && in->prev_valid;
  if (have_prev) {
    std::memcpy(fw + 8 * n, in->prev_xyz, (size_t)3 * n * 4);
    std::memcpy(fw + 11 * n, in->prev_valid, (size_t)n * 4);
  }
  const size_t used = 64 * sizeof(double) + (size_t)12 * n * 4;
  const double tm1 = solve_timing ? now_us() : 0;
  HIP_TRY(c, hipMemcpyAsync(c->d_solve_in, c->h_solve_in, used, hipMemcpyHostToDevice, c->stream2));
  const double *dh = (const double *)c->d_solve_in;
  const float *df = (const float *)(c->d_solve_in + 64 * sizeof(double));
  float *d_xyz = (float *)c->d_solve_o;
  int *d_inl = (int *)(c->d_solve_o) + 3 * n;
  RansacWork rw = c->rw;
  rw.result = c->d_solve_res;
  rw.inliers = d_inl;
  const double thr2 = in->ransac.reproj_error * in->ransac.reproj_error;
  {
    ScopedStage st(c, stage_id(c, "solve"), 0, 0, c->stream2);
    hipLaunchKernelGGL(triangulate_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream2, dh, dh + 12, df, df + 2 * n, n, d_xyz);
    if (n >= 4) {
      hipLaunchKernelGGL(ransac_hypothesis_kernel, dim3(in->ransac.iterations), dim3(64), 0, c->stream2, dh + 24, d_xyz, df + 4 * n, n, dh + 33, in->ransac.seed, thr2, rw);
      hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(256), 0, c->stream2, dh + 24, d_xyz, df + 4 * n, n, dh + 33, in->ransac.iterations, thr2, rw);
      hipLaunchKernelGGL(solve_gate_build_kernel, dim3(1), dim3(256), 0, c->stream2, dh, c->d_solve_res, d_inl, d_xyz, df, df + 2 * n, df + 4 * n, df + 6 * n,
                         have_prev ? df + 8 * n : (const float *)nullptr, have_prev ? (const int *)(df + 11 * n) : (const int *)nullptr, c->d_obs, c->d_ctl,
                         c->d_solve_res + 8);
      hipLaunchKernelGGL(pnp_refine_kernel<512>, dim3(1), dim3(512), 0, c->stream2, dh, dh + 12, c->d_obs, 0, (const int *)c->d_ctl, c->d_solve_res + 8,
                         in->refine.max_iterations, in->refine.huber_delta, (RefineOut *)(c->d_solve_res + 24));
    }
    HIP_TRY(c, hipGetLastError());
  }
  HIP_TRY(c, hipMemcpyAsync(c->h_solve_o, c->d_solve_o, (size_t)4 * n * 4, hipMemcpyDeviceToHost, c->stream2));
  if (n >= 4) HIP_TRY(c, hipMemcpyAsync(c->h_solve_res, c->d_solve_res, 40 * sizeof(double), hipMemcpyDeviceToHost, c->stream2));
  if (!c->ev_solve) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_solve, hipEventDisableTiming));
  HIP_TRY(c, hipEventRecord(c->ev_solve, c->stream2));
  if (solve_timing) {
    const double tm2 = now_us();
    tacc[0] += tm1 - tm0; tacc[1] += tm2 - tm1;
    if (++tcalls % 200 == 0) {
      std::fprintf(stderr, "[solve timing] pack %.1f us, enqueue %.1f us (n = %d)\n", tacc[0] / 200, tacc[1] / 200, n);
      tacc[0] = tacc[1] = 0;
    }
  }
  pend.active = true;
  c->solve_pending = pend;
  return SPVO_OK;
}

int spvo_solve_wait(spvo_ctx *c, spvo_solve_output *out, float *xyz, int32_t *inliers) {
  if (!c || !out) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (!c->solve_pending.active) return fail(c, SPVO_ERR_STATE, "no solve pending");
  const spvo_ctx::SolvePending pend = c->solve_pending;
  const int n = pend.n;
  if (n > 0 && (!xyz || !inliers)) return fail(c, SPVO_ERR_INVALID, "bad argument");   // (the solve stays pending)
  c->solve_pending.active = false;
  std::memset(out, 0, sizeof *out);
  // rvec -> quaternion of the prior: the answer when nothing can be estimated (base.cpp:244-250, 274-280)
  auto prior_pose = [&]() {
    const double *r = pend.rvec;
    const double a = std::sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2]);
    double ax[3] = {r[0], r[1], r[2]};
    if (a > 0) for (int k = 0; k < 3; ++k) ax[k] /= a;
    const double sn = std::sin(a / 2);
    out->q[0] = ax[0] * sn; out->q[1] = ax[1] * sn; out->q[2] = ax[2] * sn; out->q[3] = std::cos(a / 2);
    for (int k = 0; k < 3; ++k) { out->t[k] = pend.tvec[k]; out->rvec[k] = pend.rvec[k]; out->tvec[k] = pend.tvec[k]; }
  };
  if (n == 0) { prior_pose(); return SPVO_OK; }
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  {
    const double tw0 = diag_now_us();
    HIP_TRY(c, wait_event(c->ev_solve));
    g_diag.max_solve_wait = std::max(g_diag.max_solve_wait, diag_now_us() - tw0);
  }
  std::memcpy(xyz, c->h_solve_o, (size_t)3 * n * 4);
  if (n < 4) { prior_pose(); return SPVO_OK; }                                      // no model possible: prior is kept
  const double *res = c->h_solve_res, *gate = res + 8, *ref = res + 24;
  out->pnp_ok = res[6] != 0;
  out->n_inliers = (int)res[7];
  if (out->n_inliers > 0) std::memcpy(inliers, (const int *)c->h_solve_o + 3 * n, (size_t)out->n_inliers * 4);
  out->accepted = gate[7] != 0;
  for (int k = 0; k < 3; ++k) { out->rvec[k] = gate[10 + k]; out->tvec[k] = gate[13 + k]; }
  const bool ran = out->accepted && pend.refinement_degree > 0;
  out->summary.iterations = (int)ref[7];
  out->summary.converged = (int)ref[8];
  out->summary.usable = (int)ref[9];
  out->summary.initial_cost = ref[10];
  out->summary.final_cost = ref[11];
  out->refined = ran && out->summary.usable && out->summary.converged;             // base.cpp:366-374
  const double *src = out->refined ? ref : gate;
  for (int k = 0; k < 4; ++k) out->q[k] = src[k];
  for (int k = 0; k < 3; ++k) out->t[k] = src[4 + k];
  return SPVO_OK;
}

int spvo_solve_stereo_odometry(spvo_ctx *c, const spvo_solve_input *in, spvo_solve_output *out, float *xyz, int32_t *inliers) {
  if (!c || !in || !out || (in->n > 0 && (!xyz || !inliers))) return fail(c, SPVO_ERR_INVALID, "null argument");
  const int rc = spvo_solve_submit(c, in);
  return rc ? rc : spvo_solve_wait(c, out, xyz, inliers);
}

void *spvo_stream(spvo_ctx *c) { return c ? (void *)c->stream : nullptr; }

int spvo_synchronize(spvo_ctx *c) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream_t));
  HIP_TRY(c, hipStreamSynchronize(c->stream2));
  return SPVO_OK;
}

int spvo_profile_enable(spvo_ctx *c, int on) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  if (!on) resolve_pending(c);
  c->prof = on != 0;
  return SPVO_OK;
}

int spvo_profile_only(spvo_ctx *c, const char *stage) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  resolve_pending(c);
  c->prof_only = (stage && *stage) ? stage_id(c, stage) : -1;
  return SPVO_OK;
}

int spvo_profile_reset(spvo_ctx *c) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  resolve_pending(c);
  for (auto &s : c->stages) { s.total_ms = 0; s.calls = 0; }
  return SPVO_OK;
}

int spvo_profile_count(spvo_ctx *c) {
  if (!c) return 0;
  resolve_pending(c);
  return (int)c->stages.size();
}

int spvo_profile_get(spvo_ctx *c, int i, char *name, size_t name_cap, double *total_ms, long long *calls, double *flops_per_call, double *bytes_per_call) {
  if (!c || i < 0 || i >= (int)c->stages.size()) return fail(c, SPVO_ERR_INVALID, "bad stage index");
  resolve_pending(c);
  const Stage &s = c->stages[i];
  if (name && name_cap) { std::strncpy(name, s.name.c_str(), name_cap - 1); name[name_cap - 1] = 0; }
  if (total_ms) *total_ms = s.total_ms;
  if (calls) *calls = s.calls;
  if (flops_per_call) *flops_per_call = s.flops;
  if (bytes_per_call) *bytes_per_call = s.bytes;
  return SPVO_OK;
}

}  // extern "C"
