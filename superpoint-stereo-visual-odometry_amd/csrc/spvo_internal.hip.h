// spvo_internal.hip.h -- what the translation units of libspvo.so share: the context (struct spvo_ctx), the execution plan's
// records, error / profiling helpers and the functions one unit offers the others.  Not installed, not part of the C ABI
// (include/spvo.h is); everything here has hidden visibility.
//
//   spvo_core.hip      context life cycle, engine files (plan loader, weight repacking), profiling entry points
//   spvo_net_f32.hip   FP32 engines: direct + Winograd convolution launchers, the layer executor (run_ops)
//   spvo_net_f16.hip   FP16 engines                     spvo_net_s3.hip   FP32 engines in split (bf16x3) mode
//   spvo_net_i8.hip    INT8 engines
//   spvo_detect.hip    preprocess, heat map / NMS / sampling, the detector submissions, spvo_forward, ORB
//   spvo_match.hip     descriptor matching (L2, Hamming)
//   spvo_solve.hip     triangulation, PnP-RANSAC, gating, Levenberg-Marquardt, the fused solve
#pragma once
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
#include "spvo_types.hip.h"

#pragma GCC visibility push(hidden)
using namespace spvo;

#include "launch_segments.hip.h"

namespace spvo_int {

constexpr int RING = 8;          // buffer sets a detector submission owns (network outputs, heat map, NMS state, counters, host mirrors)
constexpr int MAX_INFLIGHT = 6;  // detector submissions that may be queued at once (< RING - 1: the sets of the pairs just completed still serve their matches and mirrors)
constexpr int N_SLOTS = 16;      // feature slots: 8 stereo pairs (previous, current and up to six in flight)

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
  bool wino = false;       // FP32 engines: this 3x3 layer runs a Winograd kernel: F(2x2,3x3) (conv_wino2.hip.h) unless wino4 is set
  bool wino_narrow = false;   // ... with 32 instead of 64 output channels per workgroup (layers whose 64-channel tiles would leave CUs idle)
  bool wino4 = false;      // ... the F(4x4,3x3) form (conv_wino4.hip.h: 9/16 of F(2x2)'s matrix work; pooled layers with even H and W, unpooled layers of any size)
  bool dominant = false;   // the op with the most FLOPs: launched under its own kernel name (TAG = 1)
  float *d_w = nullptr, *d_b = nullptr, *d_bn_scale = nullptr, *d_bn_shift = nullptr;
  int *d_sched = nullptr;      // Winograd layers (8-wave form): {8 band counters, workgroups done}, zero between launches (SPVO_WINO_DYNAMIC=0: none)
  _Float16 *d_w16 = nullptr;   // FP16 engines: pack_conv_weights_f16()
  int8_t *d_w8 = nullptr;      // INT8 engines: pack_conv_weights_i8()
  unsigned short *d_ws3 = nullptr;   // FP32 engines in split mode: pack_conv_weights_s3()
  int *d_wq32 = nullptr;       // INT8 engines, depthwise: quantised weights [C][9] as int32
  int *d_wsel = nullptr;       // INT8 engines, depthwise: the dot4 operands of the fused block kernel (conv_i8_fused.hip.h: pack_dw_wsel)
  int fused_dw = -1;           // INT8 engines, pointwise 1x1 op: index of the depthwise op in front of it that runs in the SAME launch (dwpw_i8_kernel), or -1
  bool fused_stem = false;     // ... and ops 0, 1 (the fp32 stem) as well
  bool fused_away = false;     // INT8 engines: this op's work is done by a later op's launch
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
  double flops_sum = 0, bytes_sum = 0;   // ... summed over the timed calls (launches of two and of four images mix under trunk pairing)
};

struct Pending { int stage; hipEvent_t e0, e1; double flops = 0, bytes = 0; };

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
  float *d_dt = nullptr;      // [cap][match_ldt(cap)] approximate squared distances of every pair (K12a -> K12b; the fp8 shortlist mode)
  int2 *d_cand = nullptr;     // [cap][nt][MATCH_C] a tile's survivors per query row (fused K12a -> K12m), nt = ceil(cap / 128)
  int4 *d_meta = nullptr;     // [cap][nt] survivor count and the two smallest upper bounds per (row, tile)
  int *d_best_idx = nullptr;
  unsigned long long *d_train_best = nullptr;
  unsigned char *d_a8 = nullptr, *d_b8 = nullptr;   // fp8 copies of both sides (spvo_set_match_fp8)
  float2 *d_qa8 = nullptr, *d_qb8 = nullptr;        // [cap] {|x - x8|, |x8|} per row of those copies: the certificate of the exact second pass
  int2 *d_out = nullptr;      // packed result, points into spvo_ctx::d_match_out
};

struct NmsImage {
  NmsBuffers b;
};

}  // namespace spvo_int
using namespace spvo_int;

struct CropGeomS { int row_off = 0, col_off = 0, crop_rows = 0, crop_cols = 0; float scale = 1.f; };

struct PendingDetect {           // one spvo_detect*_submit in flight
  CropGeomS g;
  int rows = 0, cols = 0, slot_l = 0, slot_r = 0, prev_l = -1, ring = 0;
  bool rematch = false;          // the temporal partner's keypoints were redone after this submission matched against them
  int extras = 0;                // spvo_detect_submit: bit 0 resized images, bit 1 descriptors travel to the set's pinned mirrors
  bool early_res = false;        // the resized images leave for their pinned mirror behind the first layer (copy kernel on the tail stream, ev_res), under the network
  bool launched = false;         // its trunk and tail are enqueued (false: held for a partner, spvo_set_trunk_pairing)
  bool failed = false;           // its group's launch failed after the submission had been accepted: spvo_detect_wait / _collect takes it off the queue and reports that
  int img0 = 0;                  // its first image in the network's planes (0, or 2 as the second pair of a group)
  int tring = 0;                 // the set whose network outputs hold its detector / descriptor maps (its own, or its group's first)
  int ts = 0;                    // the tail stream its tail runs on (0: stream_t, 1: stream_tb)
  // preprocess fused into the first layer (conv_first_pre.hip.h): what the launch of its group needs of the submission's images
  bool pre_pending = false;
  const uint8_t *src[2] = {nullptr, nullptr};
  uint8_t *res_dst = nullptr;    // the submission's own buffer for its two resized images, or NULL (the context's)
  size_t stride = 0;
};


// tuning "trunk_timing" (diagnostic): timing events at both ends of every trunk and tail, and what is summed from them
struct TrunkDiag {
  static constexpr int TT = 16;
  hipEvent_t b[TT] = {}, e[TT] = {}, tb[TT] = {}, te[TT] = {};
  hipEvent_t base = nullptr;
  double base_host = 0, tail = 0, lag = 0, busy = 0, idle = 0, pairs = 0;
  int np[TT] = {};
  long n = 0;
  int late = 0;
  float max_idle = 0;
  std::string pat;
};

struct spvo_ctx {
  spvo_config cfg;
  int trunk_timing = 0, solve_timing = 0;   // diagnostic switches, read at spvo_create (spvo_set_tuning: "takes effect for contexts created afterwards")
  int inject_launch_failure = 0, launch_count = 0;   // tests of launch_group's error path: the n-th group launch of this context fails
  TrunkDiag tdiag;
  double solve_tacc[4] = {0, 0, 0, 0};
  long solve_tcalls = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;   // fused solve: overlaps with a detector submission in flight
  hipStream_t stream_t = nullptr;  // detector tail (heat map, NMS, sampling, matching): overlaps with the NEXT submission's network
  hipStream_t stream_tb = nullptr; // a second tail stream (tail_streams == 2): submissions alternate between the two by the parity of their set
  int tail_streams = 1;            // set by the plan loader from tuning "tail_streams" (1, or 2: opt-in for the small engines, needs GPU_MAX_HW_QUEUES >= 8)
  int ms_set = 0;                  // which set of matcher scratch enqueue_matches uses: the tail stream's index (0 for everything else)
  hipStream_t post = nullptr;      // where post-processing is enqueued right now: `stream`, or `stream_t` for a submission
  std::deque<PendingDetect> pendq;
  int held = 0;                  // submissions at the back of pendq whose trunk has not been launched yet (0 .. 2: trunk pairing)
  bool pair_trunks = false;      // spvo_set_trunk_pairing
  bool pair_always = true;       // ... the first pair of a group waits for its partner also when the network stream is idle (tuning "pair_always", read at spvo_create)
  int last_launch_ring = -1;     // ev_net[...] of the newest trunk launched
  int cur_ring = 0;                // set whose network outputs the running forward pass writes
  unsigned submit_count = 0;
  std::string error;
  bool weights = false;
  bool fp16 = false;               // the loaded engine's precision
  bool split_req = false;          // spvo_set_fp32_split / SPVO_FP32_SPLIT: FP32 engines loaded from now on run on the bf16x3 kernels
  bool s3 = false;                 // the loaded FP32 engine runs in split mode
  size_t head_start = 0;           // ops [head_start, end) = the 1x1 heads + L2 norm: a submission runs them on the tail stream
  // launch segments replayed from HIP graphs (top of this file): on for FP16 / INT8 engines (tuning "graphs": 0 off, 2 on for every engine)
  bool use_graphs = false;
  unsigned plan_gen = 0;           // grows with every engine load: part of every segment key
  unsigned alloc_gen = 0;          // ... and so does this, with every device / pinned-host allocation of the context (dev_alloc, ensure_host_sets, the matcher's buffers)
  LaunchRecorder rec;
  GraphEntry seg_T[RING][2], seg_H[RING][2], seg_A[RING], seg_B[RING];   // trunk / heads per (network set, pairs in the group); tail halves per submission set
  bool heads_fused = false;        // ... as ONE launch (heads.hip.h): FP32 engines whose tail is convPb (256 -> 65), convDb (256 -> 256), L2 norm
  bool heads_keep_raw = false;     // the fused launch also stores the un-normalised descriptor planes (spvo_forward / spvo_debug_tensor)
  float *d_heads_w = nullptr;      // pack_heads_weights()
  int8_t *d_heads_w8 = nullptr;    // INT8 engines (heads_i8.hip.h): pack_heads_weights_i8(), ...
  float *d_heads_qm = nullptr, *d_heads_b = nullptr;   // ... and per output channel of the 21 units: weight scale x input scale of its branch, bias
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
  int nms_first = 4;             // NMS launches enqueued with a submission: nms_first - 1 round launches (4 in-kernel rounds each) + the finishing kernel; what that leaves undecided is continued by the host (nms_settle)
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
  MatchScratch ms[2][2];         // [tail stream][stereo, temporal]
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
  bool host_sets_ready = false;  // d_resized_r / h_resized_r / h_desc_r of EVERY set are allocated
  hipEvent_t ev_net[RING] = {nullptr, nullptr, nullptr, nullptr}, ev_tail[RING] = {nullptr, nullptr, nullptr, nullptr};
  // a submission's tail in two parts: ev_feat = keypoints, counts and descriptors are final (what spvo_detect_wait needs), ev_tail = the
  // matches enqueued behind them have landed too (what spvo_match_slots needs); ev_copy = the descriptors of a host-image submission have reached their pinned mirror (copy kernel behind the matches)
  hipEvent_t ev_feat[RING] = {nullptr, nullptr, nullptr, nullptr}, ev_copy[RING] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_pre[RING] = {nullptr, nullptr, nullptr, nullptr}, ev_res[RING] = {nullptr, nullptr, nullptr, nullptr};   // first layer done (network stream) / resized images on the host (tail stream)
  hipEvent_t ev_up[RING] = {nullptr, nullptr, nullptr, nullptr};   // a queued host-image submission's upload, on the solver's stream, has landed (its preprocess kernel waits for it)
  hipEvent_t ev_post = nullptr, ev_post_b = nullptr;    // PostScope: orders a synchronous entry point behind what is left on the tail stream(s)
  hipEvent_t ev_heads[RING] = {};  // the group's heads are done (tail_streams == 2, heads on the tail stream: the second pair's stream waits for it)
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
    size_t pyr_cap = 0;       // bytes per pyramid buffer (all levels side by side), key / rank entries, resize-table ints the buffers hold:
    size_t key_cap = 0;       // each is compared with what an image NEEDS (they depend on rows and cols separately, not on rows x cols)
    size_t tab_cap = 0;
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
  // fused solve: one packed input, one packed result -- per buffer SET: up to three solves may be pending (spvo_solve_submit .. _wait), a frame's
  // chain enqueued before the previous frames' have been collected; the sets rotate, so the previous solve's points (prev_index) sit in the set before.
  // The TAIL kernel of a solve submitted with `late_prior` is held back (tail_deferred) and goes out in ONE launch with the hypotheses of the next
  // submission (solve_hyp_tail_kernel: the two overlap) -- or alone, when the solve is waited for first.
  static constexpr int SOLVE_SLOTS = 3;   // solves that may be pending
  static constexpr int SOLVE_BUFS = 4;    // buffer sets they rotate through (a pending solve's tail reads the set before its own)
  bool tail_deferred = false;             // the newest submission's tail kernel has not been launched yet ...
  int tail_slot = -1;                     // ... its set, and its arguments (SolveTailArgs, odometry.hip.h: plain data)
  alignas(16) char tail_args[320];
  int solve_fuse = 1;                     // tuning "solve_fuse" (read at spvo_create): 0 = every tail is launched with its own submission
  int *x_counts[SOLVE_BUFS] = {};         // RANSAC scratch of sets 1 .. (set 0: rw): a solve's hypotheses are written while its predecessor's are read
  double *x_poses[SOLVE_BUFS] = {};
  ObsDev *x_obs[SOLVE_BUFS] = {};         // residual blocks of sets 1 .. (set 0: d_obs)
  struct SolvePending { int n = 0, refinement_degree = 0, slot = 0, frame_count = 0; bool late = false; double rvec[3] = {0, 0, 0}, tvec[3] = {0, 0, 0}; };
  std::deque<SolvePending> solve_q;     // oldest first, at most SOLVE_SLOTS
  int solve_next_slot = 0;
  int solve_last_slot = -1, solve_last_n = 0;   // the most recent submission: where its triangulated points are and how many
  hipEvent_t ev_solve[SOLVE_BUFS] = {};
  int solve_cap = 0;
  char *d_solve_in[SOLVE_BUFS] = {}, *h_solve_in[SOLVE_BUFS] = {};      // 64 doubles + 12*cap words
  double *d_solve_res[SOLVE_BUFS] = {}, *h_solve_res[SOLVE_BUFS] = {};  // ransac[8] gate[16] refine[12] + pad
  char *d_solve_o[SOLVE_BUFS] = {}, *h_solve_o[SOLVE_BUFS] = {};        // xyz [3n] floats, inliers [n] ints
  int *d_ctl = nullptr;                   // [SOLVE_BUFS][4]

  // profiling
  bool prof = false;
  int prof_only = -1;            // >= 0: only this stage is timed (spvo_profile_only)
  std::vector<Stage> stages;
  std::vector<Pending> pending;
  bool pre_fused = false;          // set by the plan loader: a submission's crop / resize / normalise runs inside its group's first layer (conv_first_pre.hip.h; tuning "preprocess_fused")
  bool heads_on_net = false;       // set by the plan loader: the heads of a submission stay on the network stream (VGG fp32) or go to the tail stream
  std::vector<hipEvent_t> free_events;
};

namespace spvo_int {

// SPVO_TRUNK_TIMING diagnostics: where the host spends its time between two submissions (maxima over the 200 submissions of a report)
struct HostDiag { double t_last_submit = 0, max_interval = 0, max_tail_wait = 0, max_solve_wait = 0; int match_miss = 0, late = 0, depth_sum = 0;
                  double iv_tail = 0, iv_match = 0, iv_solve = 0; int iv_printed = 0; long launches = 0; };   // iv_*: host time inside the three waits since the previous launch
extern HostDiag g_diag;
inline double diag_now_us() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; }

int fail(spvo_ctx *c, int code, const char *fmt, ...);

#define HIP_TRY(c, expr)                                                                     \
  do {                                                                                       \
    hipError_t _e = (expr);                                                                  \
    if (_e != hipSuccess)                                                                    \
      return fail(c, SPVO_ERR_DEVICE, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), \
                  __FILE__, __LINE__);                                                       \
  } while (0)

template <typename T>
int dev_alloc(spvo_ctx *c, T **p, size_t count, bool zero = true) {
  ++c->alloc_gen;   // (a recorded launch segment holds device pointers: every allocation starts a new generation of segment keys)
  HIP_TRY(c, hipMalloc((void **)p, std::max<size_t>(count, 1) * sizeof(T)));
  if (zero) HIP_TRY(c, hipMemsetAsync(*p, 0, std::max<size_t>(count, 1) * sizeof(T), c->stream));
  return SPVO_OK;
}

int stage_id(spvo_ctx *c, const std::string &name);
hipError_t wait_event(hipEvent_t ev);
// a diagnostic switch (spvo_set_tuning, include/spvo.h): the value set for `name`, or `dflt`.  Never the environment.
int tuning(const char *name, int dflt);
// trunk pairing: a pair held for a partner whose predecessor's trunk has meanwhile finished is launched alone (called from the entry
// points a host passes through while it waits: the network stream must not idle because the partner is late)
int release_held_if_idle(spvo_ctx *c);
hipEvent_t get_event(spvo_ctx *c);
void resolve_pending(spvo_ctx *c);
// launch segments (spvo_core.hip): seg_begin opens one on `stream` when graphs are on for this context (and the profiler is off) and returns
// whether it did; seg_end closes it -- graph replay, or plain launches + a graph for the next time; seg_free_all drops every graph
bool seg_begin(spvo_ctx *c, GraphEntry *e, unsigned long long key, hipStream_t stream);
int seg_end(spvo_ctx *c);
void seg_abort(spvo_ctx *c);      // drops an open segment without launching it (a failed group launch; a stale one found at the next begin)
void seg_free_all(spvo_ctx *c);
unsigned tuning_generation();
inline unsigned long long seg_key(std::initializer_list<long long> v) {
  unsigned long long h = 1469598103934665603ull;
  for (long long x : v) { h ^= (unsigned long long)x + 0x9E3779B97F4A7C15ull; h *= 1099511628211ull; }
  return h | 1ull;   // never 0
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
    c->pending.push_back({id, e0, e1, c->stages[id].flops, c->stages[id].bytes});
    if (c->pending.size() > 8192) resolve_pending(c);
  }
};

// a tensor's buffer for the submission being enqueued (tensors a tail reads have one per submission set)
inline float *ring_ptr(spvo_ctx *c, const Tensor &t) { return t.dr[c->cur_ring] ? t.dr[c->cur_ring] : t.d; }

// post-processing issued by a synchronous entry point while submissions are queued goes behind them
struct PostScope {
  spvo_ctx *c;
  explicit PostScope(spvo_ctx *ctx) : c(ctx) {
    c->post = c->pendq.empty() ? c->stream : c->stream_t;
    c->ms_set = 0;
    // nothing queued, but the matches of the submission collected last may still run on the tail stream (spvo_detect_wait returns
    // when the FEATURES are final) and they share the matcher's scratch: the network stream waits for them, asynchronously
    if (c->pendq.empty() && c->ev_post && hipEventRecord(c->ev_post, c->stream_t) == hipSuccess) (void)hipStreamWaitEvent(c->stream, c->ev_post, 0);
    if (c->stream_tb && c->ev_post_b && hipEventRecord(c->ev_post_b, c->stream_tb) == hipSuccess) (void)hipStreamWaitEvent(c->post, c->ev_post_b, 0);
  }
  ~PostScope() { c->post = c->stream; }
};

// ---- spvo_core.hip
void free_plan(spvo_ctx *c);
// ---- spvo_net_*.hip
int launch_conv16(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream);
int launch_conv_s3(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream);
int launch_conv8(spvo_ctx *c, const Op &op, int img0, int batch, hipStream_t stream);
int launch_heads8(spvo_ctx *c, int batch, hipStream_t stream);   // spvo_net_i8.hip: the fused tail of an INT8 engine (heads_i8.hip.h)
void plan_int8_fusion(spvo_ctx *c);   // spvo_net_i8.hip: marks the MobileNet blocks (and the stem) of a loaded INT8 plan that run as one launch
int launch_maxpool_f16(spvo_ctx *c, const Tensor &ti, const Tensor &to, const float *tin, float *tout, int batch, hipStream_t stream);
void launch_unpad_c8(const Tensor &t, int batch, float *dst, hipStream_t stream);    // spvo_debug_tensor
void launch_unpad_s3(const Tensor &t, int batch, float *dst, hipStream_t stream);
void launch_unpad_c16(const Tensor &t, int batch, float *dst, hipStream_t stream);
int run_ops(spvo_ctx *c, int batch, size_t first, size_t last, hipStream_t stream);   // ops [first, last) on `stream`
int run_network(spvo_ctx *c, int batch);
// ---- spvo_detect.hip
void linear_coeffs(int dst, int src, std::vector<int> &idx, std::vector<int> &a0, std::vector<int> &a1);
// ---- spvo_match.hip
int ensure_match(spvo_ctx *c, int na, int nb);
struct MatchReq {
  const float *dA, *dB;
  int na, nb;                     // counts, or upper bounds when the pointers are set
  const int *na_ptr, *nb_ptr;
  const float *sqA, *sqB;         // squared norms if already known (feature slots), else NULL
};
int enqueue_matches(spvo_ctx *c, const MatchReq *req_in, int njobs, int selector, int cross_check, float ratio, int2 *host_out);

}  // namespace spvo_int
#pragma GCC visibility pop
