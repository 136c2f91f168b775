// spvo_core.hip -- context life cycle, engine files (the plan loader: SPVW0003 -> tensors, ops, repacked weights), profiling.
// Part of the extern "C" shim declared in include/spvo.h; the kernels live in the headers next to this file.
#include <atomic>
#include "spvo_internal.hip.h"
#include <mutex>
#include "conv_mfma.hip.h"
#include "conv_f16.hip.h"
#include "conv_bf16x3.hip.h"
#include "conv_wino2.hip.h"
#include "conv_wino4.hip.h"
#include "heads.hip.h"
#include "conv_i8.hip.h"
#include "conv_i8_fused.hip.h"
#include "heads_i8.hip.h"

namespace spvo_int {

thread_local std::string g_error;  // for calls without a context
HostDiag g_diag;

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

int stage_id(spvo_ctx *c, const std::string &name) {
  for (size_t i = 0; i < c->stages.size(); ++i)
    if (c->stages[i].name == name) return (int)i;
  Stage s;
  s.name = name;
  c->stages.push_back(s);
  return (int)c->stages.size() - 1;
}

// ---- diagnostic switches (include/spvo.h: spvo_set_tuning).  One process-wide table, filled by explicit calls only.
namespace {
const char *const kTuningNames[] = {"winograd", "wino4", "wino_narrow", "wino_dynamic", "winograd_min_tiles", "wino4_min_tiles", "merge_siblings", "heads_fused",
                                    "heads_on_net", "heads_split", "match_fused", "fp32_split", "prematch", "spin_wait", "trunk_timing", "solve_timing", "nms_first", "upload_side", "inject_launch_failure", "int8_fused", "pair_always", "preprocess_fused", "heads_keep_raw", "tail_streams", "solve_collect_first", "graphs", "solve_fuse", "solve_keep"};
constexpr int kTuningCount = sizeof kTuningNames / sizeof kTuningNames[0];
std::mutex g_tuning_mutex;
std::atomic<unsigned> g_tuning_gen{1};   // grows with every spvo_set_tuning / spvo_clear_tuning: launch code reads switches as it goes, so recorded launch segments depend on them
bool g_tuning_set[kTuningCount] = {};
int g_tuning_value[kTuningCount] = {};
int tuning_index(const char *name) {
  for (int i = 0; name && i < kTuningCount; ++i)
    if (std::strcmp(name, kTuningNames[i]) == 0) return i;
  return -1;
}
}  // namespace
int tuning(const char *name, int dflt) {
  const int i = tuning_index(name);
  if (i < 0) return dflt;
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  return g_tuning_set[i] ? g_tuning_value[i] : dflt;
}

// tuning "spin_wait" = 1: the waits of the per-frame path poll their event instead of sleeping in the driver (a sleeping host thread
// pays the wake-up latency of its core at every wait).  Off by default: on the bench box it changed nothing (the waits are
// dominated by GPU time), and a ROS node should not burn a core while it waits.
bool spin_wait_enabled() { return tuning("spin_wait", 0) != 0; }
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
      c->stages[p.stage].flops_sum += p.flops;
      c->stages[p.stage].bytes_sum += p.bytes;
    }
    c->free_events.push_back(p.e0);
    c->free_events.push_back(p.e1);
  }
  c->pending.clear();
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

// ---------------------------------------------------------------------------------------------------------------- launch segments
// (spvo_internal.hip.h, top: what they are and why)
__thread LaunchRecorder *t_rec = nullptr;
unsigned tuning_generation() { return g_tuning_gen.load(); }

void rec_flush_direct(LaunchRecorder *r) {
  for (const LaunchNode &n : r->nodes) {
    void *params[33];
    for (int i = 0; i < n.n_args; ++i) params[i] = r->arena.data() + n.arg_off[i];
    (void)hipLaunchKernel(n.func, n.grid, n.block, params, n.lds, r->stream);
  }
  r->nodes.clear();
  r->used = 0;
}

static void seg_drop(GraphEntry &e) {
  if (e.exec) (void)hipGraphExecDestroy(e.exec);
  if (e.graph) (void)hipGraphDestroy(e.graph);
  e = GraphEntry();
}

void seg_free_all(spvo_ctx *c) {
  for (int r = 0; r < RING; ++r) {
    for (int k = 0; k < 2; ++k) { seg_drop(c->seg_T[r][k]); seg_drop(c->seg_H[r][k]); }
    seg_drop(c->seg_A[r]);
    seg_drop(c->seg_B[r]);
  }
}

void seg_abort(spvo_ctx *c) {
  LaunchRecorder &r = c->rec;
  if (t_rec == &r) t_rec = nullptr;
  r.active = false; r.poisoned = false;
  r.nodes.clear();
  r.used = 0;
}

bool seg_begin(spvo_ctx *c, GraphEntry *e, unsigned long long key, hipStream_t stream) {
  LaunchRecorder &r = c->rec;
  if (t_rec == &r) seg_abort(c);   // (an error return between a begin and its end left it open: what it held was never launched and belongs to a failed submission)
  if (!c->use_graphs || !e || e->never || t_rec) return false;   // (with the profiler on, a segment that holds a timed stage is flushed by the stage's first event)
  if (r.arena.empty()) r.arena.resize(64 << 10);
  r.active = true; r.poisoned = false; r.stream = stream; r.entry = e; r.key = key;
  r.nodes.clear();
  r.used = 0;
  t_rec = &r;
  return true;
}

int seg_end(spvo_ctx *c) {
  LaunchRecorder &r = c->rec;
  if (t_rec != &r) return SPVO_OK;
  t_rec = nullptr;
  const bool was_active = r.active;
  r.active = false;
  if (r.poisoned || !was_active) return SPVO_OK;   // (flushed already; what came behind went out as plain launches)
  GraphEntry &e = *r.entry;
  const size_t n = r.nodes.size();
  if (n == 0) return SPVO_OK;
  auto dim_code = [](const LaunchNode &q) {
    return ((unsigned long long)q.grid.x << 40) ^ ((unsigned long long)q.grid.y << 28) ^ ((unsigned long long)q.grid.z << 20) ^ ((unsigned long long)q.block.x << 8) ^ (unsigned long long)q.lds * 0x9E3779B1ull;
  };
  if (e.valid && e.key == r.key && e.funcs.size() == n) {
    bool same = true;
    for (size_t i = 0; i < n && same; ++i) same = e.funcs[i] == r.nodes[i].func && e.dims[i] == dim_code(r.nodes[i]);
    if (same && ::hipGraphLaunch(e.exec, r.stream) == hipSuccess) {
      ++r.graph_launches;
      c->stages[stage_id(c, "segment_graph_launch")].calls += 1;   // (counted with profiling off too: tests and bench.py read it)
      r.nodes.clear();
      r.used = 0;
      return SPVO_OK;
    }
    seg_drop(e);   // not the segment it was recorded as (or the replay failed): plain launches now, a new graph next time
  }
  // plain launches; and a graph for the next time when this key comes by the second time in a row for this entry (a key that keeps changing
  // would pay an instantiation per launch)
  const bool build = e.seen_key == r.key && n >= 2;
  e.seen_key = r.key;
  ++r.direct_segments;
  c->stages[stage_id(c, "segment_plain_launch")].calls += 1;
  hipError_t first_err = hipSuccess;
  for (const LaunchNode &q : r.nodes) {   // (the nodes stay recorded: the graph below is built from them)
    void *params[33];
    for (int a = 0; a < q.n_args; ++a) params[a] = r.arena.data() + q.arg_off[a];
    const hipError_t le = hipLaunchKernel(q.func, q.grid, q.block, params, q.lds, r.stream);
    if (le != hipSuccess && first_err == hipSuccess) first_err = le;
  }
  if (first_err != hipSuccess) {   // what the enqueueing code's own hipGetLastError checks would have seen with plain launches
    r.nodes.clear();
    r.used = 0;
    return fail(c, SPVO_ERR_DEVICE, "kernel launch failed (%s) in a launch segment", hipGetErrorString(first_err));
  }
  if (build) {
    seg_drop(e);
    e.seen_key = r.key;
    bool ok = ::hipGraphCreate(&e.graph, 0) == hipSuccess;
    hipGraphNode_t prev = nullptr;
    for (size_t i = 0; i < n && ok; ++i) {
      const LaunchNode &q = r.nodes[i];
      void *params[33];
      for (int a = 0; a < q.n_args; ++a) params[a] = r.arena.data() + q.arg_off[a];
      hipKernelNodeParams kp{};
      kp.func = const_cast<void *>(q.func);
      kp.gridDim = q.grid; kp.blockDim = q.block; kp.sharedMemBytes = q.lds;
      kp.kernelParams = params; kp.extra = nullptr;
      hipGraphNode_t node = nullptr;
      ok = ::hipGraphAddKernelNode(&node, e.graph, prev ? &prev : nullptr, prev ? 1 : 0, &kp) == hipSuccess;
      prev = node;
    }
    if (ok) ok = ::hipGraphInstantiate(&e.exec, e.graph, nullptr, nullptr, 0) == hipSuccess;
    if (ok) {
      e.valid = true; e.key = r.key;
      e.funcs.clear(); e.dims.clear();
      for (const LaunchNode &q : r.nodes) { e.funcs.push_back(q.func); e.dims.push_back(dim_code(q)); }
    } else {
      (void)hipGetLastError();
      seg_drop(e);
      e.never = true;   // this segment cannot be a graph here: never try again
    }
  }
  r.nodes.clear();
  r.used = 0;
  return SPVO_OK;
}

void free_plan(spvo_ctx *c) {
  seg_free_all(c);
  ++c->plan_gen;
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
    if (o.d_wsel) (void)hipFree(o.d_wsel);
    if (o.d_qm) (void)hipFree(o.d_qm);
    if (o.d_sched) (void)hipFree(o.d_sched);
  }
  if (c->d_heads_w) (void)hipFree(c->d_heads_w);
  if (c->d_heads_w8) (void)hipFree(c->d_heads_w8);
  if (c->d_heads_qm) (void)hipFree(c->d_heads_qm);
  if (c->d_heads_b) (void)hipFree(c->d_heads_b);
  c->d_heads_w = nullptr; c->d_heads_w8 = nullptr; c->d_heads_qm = c->d_heads_b = nullptr; c->heads_fused = false;
  c->tensors.clear(); c->ops.clear(); c->weights = false; c->fp16 = false; c->int8 = false; c->s3 = false;
}

}  // namespace spvo_int

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
  c->H = cfg->net_height; c->W = cfg->net_width; c->Hc = c->H / 8; c->Wc = c->W / 8;
  c->nms_first = std::min(std::max(tuning("nms_first", 4), 1), NMS_MAX_LAUNCH);   // (diagnostic override)
  c->pair_always = tuning("pair_always", 1) != 0;
  c->trunk_timing = tuning("trunk_timing", 0);
  c->solve_timing = tuning("solve_timing", 0);
  c->inject_launch_failure = tuning("inject_launch_failure", 0);
  c->solve_fuse = tuning("solve_fuse", 1) != 0 ? 1 : 0;
  c->B = 4;   // images the activation buffers hold: two stereo pairs per trunk launch (spvo_set_trunk_pairing)
  // Non-blocking streams: work the caller puts on the NULL stream (a framework's default stream, a blocking hipMemcpy) must not
  // serialise the three streams of the pipeline against each other.  Device pointers handed to the *_dev entry points
  // have to be complete when the call is made (include/spvo.h).
  if (hipSetDevice(cfg->device) != hipSuccess || hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking) != hipSuccess || hipStreamCreateWithFlags(&c->stream_t, hipStreamNonBlocking) != hipSuccess ||
      hipStreamCreateWithFlags(&c->stream_tb, hipStreamNonBlocking) != hipSuccess) {
    delete c;
    return fail(nullptr, SPVO_ERR_DEVICE, "cannot create a stream on device %d", cfg->device);
  }
  c->post = c->stream;
  (void)hipEventCreateWithFlags(&c->ev_post, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&c->ev_post_b, hipEventDisableTiming);
  for (int r = 0; r < RING; ++r) (void)hipEventCreateWithFlags(&c->ev_heads[r], hipEventDisableTiming);
  c->split_req = tuning("fp32_split", 0) != 0;
  for (int r = 0; r < RING; ++r)
    if (hipEventCreateWithFlags(&c->ev_net[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_tail[r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_feat[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_copy[r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_pre[r], hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&c->ev_res[r], hipEventDisableTiming) != hipSuccess ||
        hipEventCreateWithFlags(&c->ev_up[r], hipEventDisableTiming) != hipSuccess) {
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
    if ((rc = dev_alloc(c, &c->d_dense_in, c->B * hw))) break;
    if ((rc = dev_alloc(c, &c->d_det_dense, (size_t)c->B * 65 * c->Hc * c->Wc))) break;
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
  if (c->stream_tb) (void)hipStreamSynchronize(c->stream_tb);
  for (hipEvent_t e : c->ev_solve) if (e) (void)hipEventDestroy(e);
  if (c->ev_post) (void)hipEventDestroy(c->ev_post);
  if (c->ev_post_b) (void)hipEventDestroy(c->ev_post_b);
  for (int r = 0; r < RING; ++r) if (c->ev_heads[r]) (void)hipEventDestroy(c->ev_heads[r]);
  resolve_pending(c);
  for (int r = 0; r < TrunkDiag::TT; ++r)
    for (hipEvent_t e : {c->tdiag.b[r], c->tdiag.e[r], c->tdiag.tb[r], c->tdiag.te[r]}) if (e) (void)hipEventDestroy(e);
  if (c->tdiag.base) (void)hipEventDestroy(c->tdiag.base);
  for (auto e : c->free_events) (void)hipEventDestroy(e);
  free_plan(c);
  void *ptrs[] = {c->d_dense_in, c->d_det_dense, c->d_resized, c->d_tab, c->d_img[0], c->d_img[1], c->d_xy_tmp, c->d_desc_tmp,
                  c->d_ma, c->d_mb, c->d_match_out, c->d_counters_all, c->d_xy_stage,
                  c->d_P, c->d_pts_a, c->d_pts_b, c->d_xyz, c->rw.counts, c->rw.poses, c->rw.result, c->rw.inliers, c->d_obs, c->d_refine};
  for (void *p : ptrs) if (p) (void)hipFree(p);
  for (auto &set : c->ms)
    for (auto &m : set)
      for (void *p : {(void *)m.d_na, (void *)m.d_nb, (void *)m.d_best_d2, (void *)m.d_dt, (void *)m.d_cand, (void *)m.d_meta, (void *)m.d_best_idx, (void *)m.d_train_best,
                      (void *)m.d_a8, (void *)m.d_b8, (void *)m.d_qa8, (void *)m.d_qb8})
        if (p) (void)hipFree(p);
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
    for (hipEvent_t e : {c->ev_net[r], c->ev_tail[r], c->ev_feat[r], c->ev_copy[r], c->ev_pre[r], c->ev_res[r], c->ev_up[r]}) if (e) (void)hipEventDestroy(e);
  }
  for (int i = 0; i < N_SLOTS; ++i) {
    void *q[] = {c->slots[i].d_xy, c->slots[i].d_xyf, c->slots[i].d_desc, c->slots[i].d_n, c->slots[i].d_sqn};
    for (void *p : q) if (p) (void)hipFree(p);
  }
  for (int sl = 0; sl < spvo_ctx::SOLVE_BUFS; ++sl) {
    for (void *dp : {(void *)c->x_counts[sl], (void *)c->x_poses[sl], (void *)c->x_obs[sl]}) if (dp) (void)hipFree(dp);
    for (void *hp : {(void *)c->h_solve_in[sl], (void *)c->h_solve_res[sl], (void *)c->h_solve_o[sl]}) if (hp) (void)hipHostFree(hp);
    for (void *dp : {(void *)c->d_solve_in[sl], (void *)c->d_solve_res[sl], (void *)c->d_solve_o[sl]}) if (dp) (void)hipFree(dp);
  }
  if (c->d_ctl) (void)hipFree(c->d_ctl);
  for (void *dp : {(void *)c->d_ham_a, (void *)c->d_ham_b, (void *)c->d_ham_idx, (void *)c->d_ham_dist, (void *)c->d_ham_vote}) if (dp) (void)hipFree(dp);
  for (void *dp : {(void *)c->orb.im, (void *)c->orb.score, (void *)c->orb.blur, (void *)c->orb.src, (void *)c->orb.tmp, (void *)c->orb.pattern, (void *)c->orb.taps, (void *)c->orb.keys,
                   (void *)c->orb.rank, (void *)c->orb.out_xy, (void *)c->orb.counters, (void *)c->orb.tab, (void *)c->orb.disc, (void *)c->orb.kps, (void *)c->orb.desc})
    if (dp) (void)hipFree(dp);
  for (auto hp : c->h_match_out) if (hp) (void)hipHostFree(hp);
  if (c->h_match_tmp) (void)hipHostFree(c->h_match_tmp);
  if (c->stream_t) (void)hipStreamDestroy(c->stream_t);
  if (c->stream_tb) (void)hipStreamDestroy(c->stream_tb);
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
  // INT8 engines: the ONNX graphs list the two branches one after the other -- convPa, convPb, convDa, convDb, L2 norm -- so the trailing run
  // of 1x1 layers the tail is made of (head_start below) would be convDb + norm only.  convPb does not feed convDa: it moves behind it,
  // the tail becomes convPb, convDb, norm as in the VGG plan, and heads_i8_kernel runs it as one launch.  (Tensor ids do not change.)
  if (hdr[5] == 2 && no >= 5) {
    const Raw &A = raws[no - 5], &B = raws[no - 4], &C = raws[no - 3], &D = raws[no - 2], &N = raws[no - 1];
    const bool branches = A.v[0] == OP_CONV && A.v[6] == 3 && B.v[0] == OP_CONV && B.v[6] == 1 && B.v[1] == A.v[2] && C.v[0] == OP_CONV && C.v[6] == 3 &&
                          C.v[1] != B.v[2] && D.v[0] == OP_CONV && D.v[6] == 1 && D.v[1] == C.v[2] && N.v[0] == OP_L2NORM && N.v[1] == D.v[2] &&
                          !(C.v[7] & FLAG_ADD) && !(B.v[7] & FLAG_ADD);
    if (branches) std::swap(raws[no - 4], raws[no - 3]);
  }
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
  if (tuning("heads_split", 1))   // 0: the heads are ordinary layers of the trunk (no separate placement)
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
    // + 24 rows of slack behind the last plane: a 16-row Winograd tile (conv_wino4.hip.h) stages 18 halo rows from its first row
    // on, up to 8 more than a plane padded to a multiple of 8 (+ 2) holds -- rows that only feed outputs below the image, which
    // are never stored, but the last plane's would lie behind the allocation
    const size_t slack = t.nhwc ? 0 : (size_t)24 * t.wp;
    int rc = dev_alloc(c, &t.d, t.per_image * c->B + slack);
    if (rc) return rc;
    bool head_input = false;   // read by the head ops, which a submission runs on its tail stream while the next trunk already runs
    for (size_t q = c->head_start; q < c->ops.size(); ++q) head_input |= c->ops[q].in == (int)ti || ((c->ops[q].flags & FLAG_ADD) && c->ops[q].residual == (int)ti);
    if ((int)ti == c->t_det || (int)ti == c->t_desc || head_input) {   // what a submission's tail reads while the next network pass already runs
      t.dr[0] = t.d;
      for (int r = 1; r < RING; ++r)
        if ((rc = dev_alloc(c, &t.dr[r], t.per_image * c->B + slack))) return rc;
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
        // r = fma(f32(acc), qm, bias) with 1 / s_out folded into both constants (oracle/net_int8.py: FOLDING): the kernels' final multiply is by 1
        std::vector<float> qm(op.cout), bq(op.cout);
        const float inv_dw = 1.f / to.scale;
        for (int o = 0; o < op.cout; ++o) { qm[o] = ws[o] * ti.scale; qm[o] = qm[o] * inv_dw; bq[o] = (payload + r.b_off)[o] * inv_dw; }
        op.inv_s_out = 1.f;
        int rc = dev_alloc(c, &op.d_wq32, wq32.size(), false);
        if (rc) return rc;
        if ((rc = dev_alloc(c, &op.d_qm, op.cout, false))) return rc;
        if ((rc = dev_alloc(c, &op.d_b, op.cout, false))) return rc;
        HIP_TRY(c, hipMemcpy(op.d_wq32, wq32.data(), wq32.size() * 4, hipMemcpyHostToDevice));
        const std::vector<int> wsel = pack_dw_wsel(wq.data(), op.cout);
        if ((rc = dev_alloc(c, &op.d_wsel, wsel.size(), false))) return rc;
        HIP_TRY(c, hipMemcpy(op.d_wsel, wsel.data(), wsel.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_qm, qm.data(), qm.size() * 4, hipMemcpyHostToDevice));
        HIP_TRY(c, hipMemcpy(op.d_b, bq.data(), (size_t)op.cout * 4, hipMemcpyHostToDevice));
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
        if (c->int8 && to.i8 && !add) {   // INT8 engines: the chain's last affine absorbs 1 / s_out (oracle/net_int8.py: FOLDING), one fp32 multiply per constant
          const float inv = 1.f / to.scale;
          for (int o = 0; o < op.cout; ++o) { sc[o] = sc[o] * inv; sh[o] = sh[o] * inv; }
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
      if (!c->s3 && op.ks == 3 && op.cin > 1 && !bn && !add && !pool && (op.cout % CO_TILE) == 0 && i + 1 < no &&
          tuning("merge_siblings", 1)) {
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
        // FOLDING (oracle/net_int8.py): quantised output, no residual, and an affine to fold into (the accumulator's, or a BatchNorm --
        // folded where its constants are built, above): the kernels then multiply by 1 at the end
        const bool fold = to.i8 && !add && (op.cin > 1 || bn);
        const float inv_fold = to.i8 ? 1.f / to.scale : 0.f;
        op.inv_s_out = fold ? 1.f : inv_fold;
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
        for (int o = 0; o < op.cout; ++o) {
          qm[o] = ws[o] * ti.scale; bp[o] = b[o];
          if (fold && !bn) { qm[o] = qm[o] * inv_fold; bp[o] = bp[o] * inv_fold; }   // (with a BatchNorm the BatchNorm's constants carry the factor)
        }
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
      // Winograd for the plain 3x3 layers (no BatchNorm / residual epilogue; a pooled layer needs even sizes: the pooling window lies
      // inside a Winograd tile).  Which form, by the number of workgroups the layer gives on this chip (`min_tiles`, default 3/4 of the
      // CUs): F(4x4,3x3) (conv_wino4.hip.h: 36 multiplies per 4x4 outputs instead of F(2x2)'s 64) where the layer has that many
      // 16 x 32 tiles; else F(2x2,3x3) (conv_wino2.hip.h) on 8 x 32 tiles of 64 output channels; else its narrow form (32 output
      // channels per workgroup: conv4a / conv4b at 45x147 are 120 wide tiles on 256 CUs); else the direct kernel.
      // Diagnostic switches (spvo_set_tuning): "winograd" = 0 direct kernels only, "wino4" = 0 F(2x2) only, "wino_narrow" = 0,
      // "winograd_min_tiles" / "wino4_min_tiles" the thresholds, "wino_dynamic" = 0 static tile assignment.
      {
        const bool wino_on = tuning("winograd", 1) != 0;
        const long wtiles = (long)((ti.W + WinoTile::TW - 1) / WinoTile::TW) * ((ti.H + WinoTile::TH - 1) / WinoTile::TH) * op.co_tiles * c->cfg.max_batch;
        const long min_tiles = tuning("winograd_min_tiles", 3 * c->num_cus / 4);
        const bool eligible = wino_on && op.ks == 3 && !bn && !add && (op.cin % WinoTile::CK) == 0 && (!pool || ((ti.H % 2) == 0 && (ti.W % 2) == 0));
        op.wino = eligible && wtiles >= min_tiles;
        // (narrow form: a workgroup's chain of items is as long as the layer's K whatever the number of workgroups, so half a chip of them
        // still beats the direct kernel -- conv4a / conv4b at 30 x 98 cells, the reference's 240 x 784 engines: 128 workgroups, 45 -> 30 us)
        if (eligible && !op.wino && tuning("wino_narrow", 1) && (op.cout % 32) == 0 && 3 * wtiles >= min_tiles) op.wino = op.wino_narrow = true;
      }
      const bool dynamic_tiles = tuning("wino_dynamic", 1) != 0;
      if (op.wino && !op.wino_narrow && (op.cin % Wino4Tile::CK) == 0 && (((ti.H | ti.W) & 1) == 0 || !pool) && tuning("wino4", 1)) {
        const long t4 = (long)((ti.W + Wino4Tile::TW - 1) / Wino4Tile::TW) * ((ti.H + Wino4Tile::TH - 1) / Wino4Tile::TH) * op.co_tiles * c->cfg.max_batch;
        if (t4 >= tuning("wino4_min_tiles", 3 * c->num_cus / 4)) {
          op.wino4 = true;
          op.ck = Wino4Tile::CK;
          op.n_chunks = op.cin / Wino4Tile::CK;
          const std::vector<float> pk = pack_conv_weights_wino4(w, b, op.cout, op.cin);
          int rc = dev_alloc(c, &op.d_w, pk.size(), false);
          if (rc) return rc;
          HIP_TRY(c, hipMemcpy(op.d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
          if (dynamic_tiles && (rc = dev_alloc(c, &op.d_sched, 16))) return rc;
          continue;
        }
      }
      if (op.wino) {
        op.ck = WinoTile::CK;
        op.n_chunks = op.cin / op.ck;
        if (op.wino_narrow) op.co_tiles = op.cout / 32;
        const std::vector<float> pk = pack_conv_weights_wino2(w, b, op.cout, op.cin, op.wino_narrow ? 32 : CO_TILE);
        int rc = dev_alloc(c, &op.d_w, pk.size(), false);
        if (rc) return rc;
        HIP_TRY(c, hipMemcpy(op.d_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
        if (dynamic_tiles && (rc = dev_alloc(c, &op.d_sched, 16))) return rc;
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
  // The tail of the SuperPoint graphs -- convPb 256 -> 65, convDb 256 -> 256 (both 1x1, plain, reading two channel ranges of one
  // tensor or two tensors), L2 normalisation -- as ONE launch (heads.hip.h) instead of three: FP32 and FP16 engines
  // (the latter: the loader wave converts the C8 fp16 activations, the weights are the fp16-rounded ones); tuning "heads_fused" = 0 keeps the plan's ops.
  if (!c->int8 && !c->s3 && c->head_start + 3 == c->ops.size() && tuning("heads_fused", 1)) {
    const size_t hs = c->head_start;
    const Op &pb = c->ops[hs], &db = c->ops[hs + 1], &nm = c->ops[hs + 2];
    const bool plain = pb.type == OP_CONV && db.type == OP_CONV && nm.type == OP_L2NORM && pb.ks == 1 && db.ks == 1 && pb.flags == 0 && db.flags == 0 &&
                       pb.cin == HEADS_CIN && db.cin == HEADS_CIN && pb.cout == 65 && db.cout == 256 && pb.out == c->t_det && pb.out_c_off == 0 &&
                       db.out_c_off == 0 && nm.in == db.out && nm.out == c->t_desc && c->tensors[pb.in].level == 3 && c->tensors[db.in].level == 3 &&
                       !c->tensors[pb.in].nhwc && !c->tensors[db.in].nhwc && !pb.merged && !db.merged &&
                       c->tensors[pb.in].f16 == c->fp16 && c->tensors[db.in].f16 == c->fp16 && (!c->fp16 || ((pb.in_c_off | db.in_c_off) % 8) == 0);   // (the two branches may read one tensor -- the VGG plan's merged convPa + convDa output -- or two)
    if (plain) {
      const std::vector<float> pk = pack_heads_weights(payload + raws[hs].w_off, payload + raws[hs].b_off, pb.cout, payload + raws[hs + 1].w_off, payload + raws[hs + 1].b_off, c->fp16);
      int rc = dev_alloc(c, &c->d_heads_w, pk.size(), false);
      if (rc) return rc;
      HIP_TRY(c, hipMemcpy(c->d_heads_w, pk.data(), pk.size() * 4, hipMemcpyHostToDevice));
      c->heads_fused = true;
    }
  }
  // INT8 engines (round 6): the same tail as ONE launch of heads_i8_kernel (heads_i8.hip.h) -- int8 C16 activations of both branches in,
  // fp32 detector planes, un-normalised and normalised descriptors out: oracle/net_int8.py's arithmetic for a layer with an fp32 output
  if (c->int8 && c->head_start + 3 == c->ops.size() && tuning("heads_fused", 1)) {
    const size_t hs = c->head_start;
    const Op &pb = c->ops[hs], &db = c->ops[hs + 1], &nm = c->ops[hs + 2];
    const bool plain = pb.type == OP_CONV && db.type == OP_CONV && nm.type == OP_L2NORM && pb.ks == 1 && db.ks == 1 && pb.flags == 0 && db.flags == 0 &&
                       pb.cin == HEADS8_CIN && db.cin == HEADS8_CIN && pb.cout == 65 && db.cout == 256 && pb.out == c->t_det && pb.out_c_off == 0 &&
                       db.out_c_off == 0 && nm.in == db.out && nm.out == c->t_desc && c->tensors[pb.in].level == 3 && c->tensors[db.in].level == 3 &&
                       c->tensors[pb.in].i8 && c->tensors[db.in].i8 && !c->tensors[pb.out].i8 && !c->tensors[db.out].i8 && ((pb.in_c_off | db.in_c_off) % 16) == 0;
    if (plain) {
      std::vector<int8_t> wq_p, wq_d;
      std::vector<float> ws_p, ws_d;
      quantize_conv_weights(payload + raws[hs].w_off, pb.cout, pb.cin, wq_p, ws_p);
      quantize_conv_weights(payload + raws[hs + 1].w_off, db.cout, db.cin, wq_d, ws_d);
      const std::vector<int8_t> pk = pack_heads_weights_i8(wq_p.data(), pb.cout, wq_d.data());
      std::vector<float> qm((size_t)HEADS8_UNITS * 16, 0.f), bp((size_t)HEADS8_UNITS * 16, 0.f);
      const float *b_p = payload + raws[hs].b_off, *b_d = payload + raws[hs + 1].b_off;
      for (int o = 0; o < pb.cout; ++o) { qm[o] = ws_p[o] * c->tensors[pb.in].scale; bp[o] = b_p[o]; }
      for (int o = 0; o < db.cout; ++o) { qm[16 * HEADS8_DET_UNITS + o] = ws_d[o] * c->tensors[db.in].scale; bp[16 * HEADS8_DET_UNITS + o] = b_d[o]; }
      int rc = dev_alloc(c, &c->d_heads_w8, pk.size(), false);
      if (rc) return rc;
      if ((rc = dev_alloc(c, &c->d_heads_qm, qm.size(), false))) return rc;
      if ((rc = dev_alloc(c, &c->d_heads_b, bp.size(), false))) return rc;
      HIP_TRY(c, hipMemcpy(c->d_heads_w8, pk.data(), pk.size(), hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(c->d_heads_qm, qm.data(), qm.size() * 4, hipMemcpyHostToDevice));
      HIP_TRY(c, hipMemcpy(c->d_heads_b, bp.data(), bp.size() * 4, hipMemcpyHostToDevice));
      c->heads_fused = true;
    }
  }
  if (c->int8 && tuning("int8_fused", 1)) plan_int8_fusion(c);
  {   // mark the dominant layer
    Op *best = nullptr;
    for (auto &o : c->ops) if (o.type == OP_CONV && (!best || o.flops_per_image > best->flops_per_image)) best = &o;
    if (best) best->dominant = true;
  }
  {   // a submission's preprocess inside the first layer's launch (conv_first_pre.hip.h): FP32 engines whose first op is the plain 3x3 layer on the input plane
    const Op &o0 = c->ops[0];
    c->pre_fused = !c->fp16 && !c->int8 && !c->s3 && c->head_start >= 1 && o0.type == OP_CONV && o0.cin == 1 && o0.ks == 3 && o0.in == c->t_input && !o0.d_bn_scale &&
                   !(o0.flags & (FLAG_POOL | FLAG_ADD)) && !c->tensors[o0.out].nhwc && (c->W % 8) == 0 && tuning("preprocess_fused", 1) != 0;
  }
  {   // one tail stream, or two when the caller says so (tuning "tail_streams" = 2: spvo_detect.hip; only with nothing in flight -- the parity
      // of a submission's set selects its stream)
    const int ts = tuning("tail_streams", 0);
    if (c->pendq.empty()) c->tail_streams = ts == 2 ? 2 : 1;
  }
  {   // where a submission's heads run (spvo_detect.hip): behind the trunk on the network stream when most of the trunk's work is in
      // launches of one 512-thread workgroup per CU (conv_wino4.hip.h), beside which they would starve; on the tail stream otherwise
    double all = 0, big = 0;
    for (const auto &o : c->ops) if (o.type == OP_CONV) { all += o.flops_per_image; if (o.wino4) big += o.flops_per_image; }
    c->use_graphs = tuning("graphs", 1) == 2 || ((c->fp16 || c->int8) && tuning("graphs", 1) != 0);   // launch segments as HIP graphs: where the host bounds the frame loop
    c->heads_on_net = big > 0.8 * all;   // VGG fp32 at 360x1176: 0.95 (on the network stream: 1308 against 1260 frames/s); sp_squeeze: 0.66 (tail stream: 1286 against 1248)
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

void *spvo_stream(spvo_ctx *c) { return c ? (void *)c->stream : nullptr; }

int spvo_synchronize(spvo_ctx *c) {
  if (!c) return fail(c, SPVO_ERR_INVALID, "null context");
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream_t));
  HIP_TRY(c, hipStreamSynchronize(c->stream_tb));
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
  for (auto &s : c->stages) { s.total_ms = 0; s.calls = 0; s.flops_sum = 0; s.bytes_sum = 0; }
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
  // per call = the MEAN over the timed calls (so that flops / (total_ms / calls) is the rate also when launches of different batch mix)
  if (flops_per_call) *flops_per_call = s.calls ? s.flops_sum / s.calls : s.flops;
  if (bytes_per_call) *bytes_per_call = s.calls ? s.bytes_sum / s.calls : s.bytes;
  return SPVO_OK;
}

int spvo_set_tuning(const char *name, int value) {
  const int i = tuning_index(name);
  if (i < 0) return SPVO_ERR_INVALID;
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  g_tuning_set[i] = true;
  g_tuning_value[i] = value;
  ++g_tuning_gen;
  return SPVO_OK;
}

int spvo_get_tuning(const char *name, int dflt) { return tuning(name, dflt); }

void spvo_clear_tuning(void) {
  std::lock_guard<std::mutex> lock(g_tuning_mutex);
  for (int i = 0; i < kTuningCount; ++i) g_tuning_set[i] = false;
  ++g_tuning_gen;
}

int spvo_profile_stage_kernel(spvo_ctx *c, const char *stage, char *name, size_t name_cap, double *executed_per_algorithmic) {
  if (!c || !stage) return fail(c, SPVO_ERR_INVALID, "null argument");
  for (size_t i = 0; i < c->ops.size(); ++i) {
    const Op &op = c->ops[i];
    if (op.stage < 0 || op.stage >= (int)c->stages.size() || c->stages[op.stage].name != stage) continue;
    const char *k = "other";
    double f = 1.0;
    if (op.type == OP_CONV) {
      if (c->int8) k = (op.fused_dw >= 0 || op.fused_stem) ? "dwpw_i8_kernel" : "conv_i8_kernel";   // (the fused block: depthwise 3x3 + pointwise 1x1 in one launch)
      else if (c->fp16) k = "conv_f16_kernel";
      else if (c->s3) { k = "conv_s3_kernel"; f = 6.0; }
      else if (op.wino4) { k = "conv_wino4_kernel"; f = 0.25; }
      else if (op.wino) { k = "conv_wino2_kernel"; f = 4.0 / 9.0; }
      else k = "conv_mfma_kernel";
    }
    if (op.type == OP_DWCONV) k = c->int8 ? "dwconv3x3_i8_kernel" : c->fp16 ? "dwconv3x3_f16_kernel" : "dwconv3x3_kernel";
    if (name && name_cap) { std::strncpy(name, k, name_cap - 1); name[name_cap - 1] = 0; }
    if (executed_per_algorithmic) *executed_per_algorithmic = f;
    return SPVO_OK;
  }
  return fail(c, SPVO_ERR_INVALID, "no layer of the loaded engine is timed under that stage name");
}

}  // extern "C"
