// spvo_solve.hip -- solveStereoOdometry (feature_detection_base.cpp:125-399) on the device: K14 triangulation, K15 PnP-RANSAC, gating, K16 LM.
#include "spvo_internal.hip.h"
#include "odometry.hip.h"

namespace spvo_int {

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
    for (int sl = 1; sl < spvo_ctx::SOLVE_BUFS; ++sl) {   // (the fused solve's sets 1 ..: spvo_solve_submit)
      drop(c->x_counts[sl]); drop(c->x_poses[sl]);
      if ((rc = dev_alloc(c, &c->x_counts[sl], cap))) return rc;
      if ((rc = dev_alloc(c, &c->x_poses[sl], (size_t)cap * 7))) return rc;
    }
    c->ransac_cap = cap;
  }
  if (n_obs > c->obs_cap) {
    const int cap = std::max(n_obs, 8192);
    c->obs_cap = 0;
    drop(c->d_obs);
    if ((rc = dev_alloc(c, &c->d_obs, cap))) return rc;
    for (int sl = 1; sl < spvo_ctx::SOLVE_BUFS; ++sl) {
      drop(c->x_obs[sl]);
      if ((rc = dev_alloc(c, &c->x_obs[sl], cap))) return rc;
    }
    c->obs_cap = cap;
  }
  return SPVO_OK;
}

}  // namespace spvo_int

// ===========================================================================
extern "C" {

int spvo_triangulate(spvo_ctx *c, const double P_l[12], const double P_r[12], const float *xy_l, const float *xy_r, int n, float *xyz) {
  if (!c || !P_l || !P_r || n < 0 || (n > 0 && (!xy_l || !xy_r || !xyz))) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (n == 0) return SPVO_OK;
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  if (!c->solve_q.empty()) return fail(c, SPVO_ERR_STATE, "a solve is pending (spvo_solve_submit): complete it with spvo_solve_wait first");
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
  if (!c->solve_q.empty()) return fail(c, SPVO_ERR_STATE, "a solve is pending (spvo_solve_submit): complete it with spvo_solve_wait first");
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
    hipLaunchKernelGGL(ransac_select_kernel, dim3(1), dim3(SOLVE_TAIL_THREADS), 0, c->stream, c->d_P + 24, c->d_pts_a, c->d_pts_b, n, c->d_P + 33, o.iterations, o.reproj_error * o.reproj_error, c->rw);
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
  if (!c->solve_q.empty()) return fail(c, SPVO_ERR_STATE, "a solve is pending (spvo_solve_submit): complete it with spvo_solve_wait first");
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
// free again when this returns.  What the wait needs later (n, the refinement degree, the prior if it was given here) stays in the context.
// Up to SOLVE_SLOTS submissions may be pending: nothing of a frame's chain needs the previous frame's POSE any more (prior-free
// hypotheses; the gate is evaluated by the wait), and the previous frame's POINTS can be referred to where they lie on the device
// (prev_index) -- so frame k's chain is enqueued behind frame k - 1's on the solver's stream without the host having seen k - 1's result.
int spvo_solve_submit(spvo_ctx *c, const spvo_solve_input *in) {
  if (!c || !in) return fail(c, SPVO_ERR_INVALID, "null argument");
  if ((int)c->solve_q.size() >= spvo_ctx::SOLVE_SLOTS) return fail(c, SPVO_ERR_STATE, "%d solves are pending: complete the oldest with spvo_solve_wait first", spvo_ctx::SOLVE_SLOTS);
  const int n = in->n;
  if (n < 0 || (n > 0 && (!in->xy_cl || !in->xy_cr || !in->xy_pl || !in->xy_pr))) return fail(c, SPVO_ERR_INVALID, "bad argument");
  if (in->ransac.iterations <= 0 || in->ransac.iterations > 65536 || !(in->ransac.reproj_error > 0) || in->refine.max_iterations < 0 ||
      !(in->refine.huber_delta > 0))
    return fail(c, SPVO_ERR_INVALID, "bad solver options");
  if (in->prev_index && (in->prev_xyz || in->prev_valid)) return fail(c, SPVO_ERR_INVALID, "prev_index and prev_xyz / prev_valid exclude each other");
  if (in->prev_index) {   // refers to the points of the submission just before this one: they must exist and the indices must lie inside them
    if (c->solve_last_slot < 0) return fail(c, SPVO_ERR_STATE, "prev_index needs a previous spvo_solve_submit on this context");
    for (int i = 0; i < n; ++i)
      if (in->prev_index[i] < -1 || in->prev_index[i] >= c->solve_last_n) return fail(c, SPVO_ERR_INVALID, "prev_index[%d] = %d outside the previous solve's %d points", i, in->prev_index[i], c->solve_last_n);
  }
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  // (a held pair's launch that fails here is not this call's failure: launch_group marks the submission, and the spvo_detect_wait /
  // _collect that asks for it reports the error)
  (void)release_held_if_idle(c);
  spvo_ctx::SolvePending pend;
  pend.n = n; pend.refinement_degree = in->refinement_degree; pend.late = in->late_prior != 0; pend.frame_count = in->frame_count;
  for (int k = 0; k < 3; ++k) { pend.rvec[k] = in->rvec_pred[k]; pend.tvec[k] = in->tvec_pred[k]; }
  // this call runs on the context's second stream so that it overlaps a detector submission in
  // flight; buffers only grow on first use (then everything is drained once)
  const bool grow = n > c->odo_cap || in->ransac.iterations > c->ransac_cap || 4 * n > c->obs_cap || !c->d_P || n > c->solve_cap || !c->solve_cap;
  if (grow && !c->solve_q.empty()) return fail(c, SPVO_ERR_STATE, "the solver's buffers have to grow for %d correspondences: complete the pending solve first", n);
  if (grow) HIP_TRY(c, hipDeviceSynchronize());
  int rc = ensure_odometry(c, std::max(n, 1), in->ransac.iterations, 4 * std::max(n, 1));
  if (rc) return rc;
  if (n > c->solve_cap || !c->solve_cap) {
    const int cap = std::max(std::max(n, 2048), c->cfg.max_keypoints);
    for (int sl = 0; sl < spvo_ctx::SOLVE_BUFS; ++sl) {
      for (void *hp : {(void *)c->h_solve_in[sl], (void *)c->h_solve_res[sl], (void *)c->h_solve_o[sl]}) if (hp) (void)hipHostFree(hp);
      for (void *dp : {(void *)c->d_solve_in[sl], (void *)c->d_solve_res[sl], (void *)c->d_solve_o[sl]}) if (dp) (void)hipFree(dp);
      c->h_solve_in[sl] = c->h_solve_o[sl] = nullptr; c->h_solve_res[sl] = nullptr;
      c->d_solve_in[sl] = c->d_solve_o[sl] = nullptr; c->d_solve_res[sl] = nullptr;
    }
    if (c->d_ctl) (void)hipFree(c->d_ctl);
    c->d_ctl = nullptr;
    c->solve_cap = 0;   // a failed allocation below leaves a context that spvo_destroy and a later call can still handle
    c->solve_last_slot = -1; c->solve_last_n = 0;
    if (in->prev_index) return fail(c, SPVO_ERR_STATE, "the solver's buffers grew: the previous solve's points are gone (pass prev_xyz for this frame)");
    const size_t in_bytes = 64 * sizeof(double) + (size_t)12 * cap * 4, o_bytes = (size_t)4 * cap * 4;
    for (int sl = 0; sl < spvo_ctx::SOLVE_BUFS; ++sl) {
      if ((rc = dev_alloc(c, &c->d_solve_in[sl], in_bytes))) return rc;
      if ((rc = dev_alloc(c, &c->d_solve_res[sl], 40))) return rc;
      if ((rc = dev_alloc(c, &c->d_solve_o[sl], o_bytes))) return rc;
      HIP_TRY(c, hipHostMalloc((void **)&c->h_solve_in[sl], in_bytes));
      HIP_TRY(c, hipHostMalloc((void **)&c->h_solve_res[sl], 40 * sizeof(double)));
      HIP_TRY(c, hipHostMalloc((void **)&c->h_solve_o[sl], o_bytes));
    }
    if ((rc = dev_alloc(c, &c->d_ctl, 4 * spvo_ctx::SOLVE_BUFS))) return rc;
    c->solve_cap = cap;
  }
  if (grow) HIP_TRY(c, hipDeviceSynchronize());
  const int sl = c->solve_next_slot;
  pend.slot = sl;
  const bool solve_timing = c->solve_timing != 0;   // diagnostic (read at spvo_create): host time per phase of this call
  double *tacc = c->solve_tacc;
  long &tcalls = c->solve_tcalls;
  auto now_us = []() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e6 + ts.tv_nsec * 1e-3; };
  const double tm0 = solve_timing ? now_us() : 0;
  // ---- pack: 64 doubles, then cl cr pl pr [2n each], prev_xyz [3n], prev_valid or prev_index [n]
  double *hdr = (double *)c->h_solve_in[sl];
  std::memset(hdr, 0, 64 * sizeof(double));
  for (int k = 0; k < 12; ++k) { hdr[k] = in->P_l[k]; hdr[12 + k] = in->P_r[k]; }
  const int kidx[9] = {0, 1, 2, 4, 5, 6, 8, 9, 10};
  for (int k = 0; k < 9; ++k) hdr[24 + k] = in->P_l[kidx[k]];                      // K = P_l[:, :3]  (base.cpp:227)
  if (!pend.late) {   // the prior is known now: the device evaluates the gate itself (a rejected frame then skips its refinement)
    for (int k = 0; k < 3; ++k) { hdr[33 + k] = in->rvec_pred[k]; hdr[36 + k] = in->tvec_pred[k]; }
    hdr[39] = in->frame_count; hdr[41] = 8.0; hdr[42] = 0.1; hdr[43] = 10; hdr[44] = 1;   // hpp:145-147
  }
  hdr[40] = in->refinement_degree;
  float *fw = (float *)(c->h_solve_in[sl] + 64 * sizeof(double));
  if (n > 0) {
    std::memcpy(fw, in->xy_cl, (size_t)2 * n * 4);
    std::memcpy(fw + 2 * n, in->xy_cr, (size_t)2 * n * 4);
    std::memcpy(fw + 4 * n, in->xy_pl, (size_t)2 * n * 4);
    std::memcpy(fw + 6 * n, in->xy_pr, (size_t)2 * n * 4);
  }
  const bool have_prev = in->prev_xyz && in->prev_valid, have_index = in->prev_index != nullptr;
  if (have_prev) {
    std::memcpy(fw + 8 * n, in->prev_xyz, (size_t)3 * n * 4);
    std::memcpy(fw + 11 * n, in->prev_valid, (size_t)n * 4);
  } else if (have_index && n > 0) {
    std::memcpy(fw + 11 * n, in->prev_index, (size_t)n * 4);
  }
  const float *prev_pts = have_index ? (const float *)c->d_solve_o[c->solve_last_slot] : nullptr;   // (the set before this one: nothing rewrites it while this solve is pending)
  if (n > 0) {
    const size_t used = 64 * sizeof(double) + (size_t)12 * n * 4;
    const double tm1 = solve_timing ? now_us() : 0;
    const double *dh = (const double *)c->d_solve_in[sl];
    const float *df = (const float *)(c->d_solve_in[sl] + 64 * sizeof(double));
    float *d_xyz = (float *)c->d_solve_o[sl];
    int *d_inl = (int *)(c->d_solve_o[sl]) + 3 * n;
    RansacWork rw = c->rw;
    if (sl) { rw.counts = c->x_counts[sl]; rw.poses = c->x_poses[sl]; }
    rw.result = c->d_solve_res[sl];
    rw.inliers = d_inl;
    const double thr2 = in->ransac.reproj_error * in->ransac.reproj_error;
    {
      ScopedStage st(c, stage_id(c, "solve"), 0, 0, c->stream2);
      // (the inputs travel inside the first kernel, the results inside the last one: see odometry.hip.h)
      hipLaunchKernelGGL(solve_in_triangulate_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream2, reinterpret_cast<const uint4 *>(c->h_solve_in[sl]),
                         reinterpret_cast<uint4 *>(c->d_solve_in[sl]), (int)(used / 16), n, d_xyz);
      if (n >= 4) {
        SolveTailArgs ta;   // selection + refit, residual blocks, refinement, results to the pinned buffers: one launch (odometry.hip.h)
        ta.hdr = dh; ta.xyz = d_xyz; ta.xy_cl = df; ta.xy_cr = df + 2 * n; ta.xy_pl = df + 4 * n; ta.xy_pr = df + 6 * n;
        ta.prev_xyz = have_prev ? df + 8 * n : nullptr; ta.prev_valid = have_prev ? (const int *)(df + 11 * n) : nullptr;
        ta.prev_index = have_index ? (const int *)(df + 11 * n) : nullptr; ta.prev_pts = prev_pts;
        ta.n = n; ta.iterations = in->ransac.iterations; ta.thr2 = thr2; ta.w = rw;
        ta.obs = sl ? c->x_obs[sl] : c->d_obs; ta.ctl = c->d_ctl + 4 * sl; ta.res = c->d_solve_res[sl];
        ta.max_iterations = in->refine.max_iterations; ta.huber_delta = in->refine.huber_delta;
        ta.d_o = reinterpret_cast<const unsigned *>(c->d_solve_o[sl]); ta.h_o = reinterpret_cast<unsigned *>(c->h_solve_o[sl]); ta.o_words = 4 * n; ta.h_res = c->h_solve_res[sl];
        // The hypotheses of THIS solve -- in one launch with the tail of the solve submitted before, if that one is still held back
        // (solve_hyp_tail_kernel: the two overlap; ev_solve of the older solve is recorded behind the launch) ...
        if (c->tail_deferred) {
          SolveTailArgs prev_ta;
          std::memcpy(&prev_ta, c->tail_args, sizeof prev_ta);
          const SolveHypArgs ha{dh + 24, d_xyz, df + 4 * n, n, dh + 33, in->ransac.seed, thr2, rw, in->ransac.iterations};
          const int per_wg = SOLVE_TAIL_THREADS / 64;
          hipLaunchKernelGGL(solve_hyp_tail_kernel, dim3(1 + (in->ransac.iterations + per_wg - 1) / per_wg), dim3(SOLVE_TAIL_THREADS), 0, c->stream2, ha, prev_ta);
          HIP_TRY(c, hipGetLastError());
          HIP_TRY(c, hipEventRecord(c->ev_solve[c->tail_slot], c->stream2));
          c->tail_deferred = false;
        } else {
          hipLaunchKernelGGL(ransac_hypothesis_kernel, dim3(in->ransac.iterations), dim3(64), 0, c->stream2, dh + 24, d_xyz, df + 4 * n, n, dh + 33, in->ransac.seed, thr2, rw);
        }
        // ... and its own tail: held back for the next submission's launch when the caller asks for it (late_prior = 2: it keeps two solves
        // pending behind every submit, so nobody waits for this one before the next is submitted), at once otherwise
        if (in->late_prior == 2 && c->solve_fuse) {
          static_assert(sizeof(SolveTailArgs) <= sizeof(spvo_ctx::tail_args) && std::is_trivially_copyable<SolveTailArgs>::value, "tail_args holds a SolveTailArgs");
          std::memcpy(c->tail_args, &ta, sizeof ta);
          c->tail_deferred = true;
          c->tail_slot = sl;
        } else {
          hipLaunchKernelGGL(solve_tail_kernel, dim3(1), dim3(SOLVE_TAIL_THREADS), 0, c->stream2, ta);
        }
      }
      HIP_TRY(c, hipGetLastError());
    }
    if (n < 4) HIP_TRY(c, hipMemcpyAsync(c->h_solve_o[sl], c->d_solve_o[sl], (size_t)4 * n * 4, hipMemcpyDeviceToHost, c->stream2));   // (no model possible: only the points travel)
    if (!c->ev_solve[sl]) HIP_TRY(c, hipEventCreateWithFlags(&c->ev_solve[sl], hipEventDisableTiming));
    if (!(c->tail_deferred && c->tail_slot == sl)) HIP_TRY(c, hipEventRecord(c->ev_solve[sl], c->stream2));   // (a held-back tail: the event follows its launch)
    if (solve_timing) {
      const double tm2 = now_us();
      tacc[0] += tm1 - tm0; tacc[1] += tm2 - tm1;
      if (++tcalls % 200 == 0) {
        std::fprintf(stderr, "[solve timing] pack %.1f us, enqueue %.1f us (n = %d)\n", tacc[0] / 200, tacc[1] / 200, n);
        tacc[0] = tacc[1] = 0;
      }
    }
  }   // (n == 0: nothing to enqueue, the wait answers with the prior)
  c->solve_q.push_back(pend);
  c->solve_last_slot = sl; c->solve_last_n = n;
  c->solve_next_slot = (sl + 1) % spvo_ctx::SOLVE_BUFS;
  return SPVO_OK;
}

// The oldest pending solve's results, the gate (base.cpp:241-272) applied HERE with the prior (rvec_pred, tvec_pred, frame_count) as the
// caller knows it now: `prior` = nullptr takes what spvo_solve_submit was given.
static int solve_wait_impl(spvo_ctx *c, const double *prior_rvec, const double *prior_tvec, int prior_frame_count, bool have_prior, spvo_solve_output *out, float *xyz,
                           int32_t *inliers) {
  if (!c || !out) return fail(c, SPVO_ERR_INVALID, "null argument");
  if (c->solve_q.empty()) return fail(c, SPVO_ERR_STATE, "no solve pending");
  if (hipSetDevice(c->cfg.device) == hipSuccess) (void)release_held_if_idle(c);   // (a failing launch is reported by that pair's spvo_detect_wait: launch_group)
  spvo_ctx::SolvePending pend = c->solve_q.front();
  const int n = pend.n, sl = pend.slot;
  if (n > 0 && (!xyz || !inliers)) return fail(c, SPVO_ERR_INVALID, "bad argument");   // (the solve stays pending)
  if (pend.late && !have_prior) return fail(c, SPVO_ERR_STATE, "the pending solve was submitted with late_prior: complete it with spvo_solve_wait_prior");
  if (have_prior && pend.late) {   // (a submission that carried its own prior keeps it: the device has gated against it)
    for (int k = 0; k < 3; ++k) { pend.rvec[k] = prior_rvec[k]; pend.tvec[k] = prior_tvec[k]; }
    pend.frame_count = prior_frame_count;
  }
  c->solve_q.pop_front();
  std::memset(out, 0, sizeof *out);
  // rvec -> quaternion of the prior: the answer when nothing can be estimated or the gate rejects (base.cpp:244-260, 274-280)
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
  if (c->tail_deferred && c->tail_slot == sl) {   // its tail was held back for a successor that did not come first: it goes out alone now
    SolveTailArgs ta;
    std::memcpy(&ta, c->tail_args, sizeof ta);
    hipLaunchKernelGGL(solve_tail_kernel, dim3(1), dim3(SOLVE_TAIL_THREADS), 0, c->stream2, ta);
    HIP_TRY(c, hipGetLastError());
    HIP_TRY(c, hipEventRecord(c->ev_solve[sl], c->stream2));
    c->tail_deferred = false;
  }
  {
    const double tw0 = diag_now_us();
    HIP_TRY(c, wait_event(c->ev_solve[sl]));
    g_diag.max_solve_wait = std::max(g_diag.max_solve_wait, diag_now_us() - tw0);
    g_diag.iv_solve += diag_now_us() - tw0;
  }
  std::memcpy(xyz, c->h_solve_o[sl], (size_t)3 * n * 4);
  if (n < 4) { prior_pose(); return SPVO_OK; }                                      // no model possible: prior is kept
  const double *res = c->h_solve_res[sl], *gate = res + 8, *ref = res + 24;
  out->pnp_ok = res[6] != 0;
  out->n_inliers = (int)res[7];
  if (out->n_inliers > 0) std::memcpy(inliers, (const int *)c->h_solve_o[sl] + 3 * n, (size_t)out->n_inliers * 4);
  // the gate (base.cpp:241-272; constants hpp:145-147): acceleration against the prediction; rejected or no model => the prediction is returned
  constexpr double TIME_INTERVAL = 0.1, MAX_ACCELERATION = 8.0;
  constexpr int IGNORE_FRAME_COUNT = 10;
  const double dx = gate[13] - pend.tvec[0], dy = gate[14] - pend.tvec[1], dz = gate[15] - pend.tvec[2];
  const double acc = std::sqrt(dx * dx + dy * dy + dz * dz) / TIME_INTERVAL;
  // (a submission that carried its prior had the same test evaluated on the device, which then skipped the refinement of a rejected frame: its
  // decision is the one that counts -- gate[7]; the two evaluations are the same three subtractions, a square root and a division in double)
  out->accepted = pend.late ? (out->pnp_ok && !(pend.frame_count > IGNORE_FRAME_COUNT && acc > MAX_ACCELERATION)) : (out->pnp_ok && gate[7] != 0);
  if (!out->accepted) { prior_pose(); return SPVO_OK; }                             // (whatever the device refined from the rejected pose is void)
  for (int k = 0; k < 3; ++k) { out->rvec[k] = gate[10 + k]; out->tvec[k] = gate[13 + k]; }
  const bool ran = pend.refinement_degree > 0;
  if (ran) {
    out->summary.iterations = (int)ref[7];
    out->summary.converged = (int)ref[8];
    out->summary.usable = (int)ref[9];
    out->summary.initial_cost = ref[10];
    out->summary.final_cost = ref[11];
  }
  out->refined = ran && out->summary.usable && out->summary.converged;             // base.cpp:366-374
  const double *src = out->refined ? ref : gate;
  for (int k = 0; k < 4; ++k) out->q[k] = src[k];
  for (int k = 0; k < 3; ++k) out->t[k] = src[4 + k];
  return SPVO_OK;
}

int spvo_solve_wait(spvo_ctx *c, spvo_solve_output *out, float *xyz, int32_t *inliers) { return solve_wait_impl(c, nullptr, nullptr, 0, false, out, xyz, inliers); }

int spvo_solve_wait_prior(spvo_ctx *c, const double rvec_pred[3], const double tvec_pred[3], int frame_count, spvo_solve_output *out, float *xyz, int32_t *inliers) {
  if (!rvec_pred || !tvec_pred) return fail(c, SPVO_ERR_INVALID, "null prior");
  return solve_wait_impl(c, rvec_pred, tvec_pred, frame_count, true, out, xyz, inliers);
}

int spvo_solve_pending(spvo_ctx *c) { return c ? (int)c->solve_q.size() : 0; }

int spvo_solve_stereo_odometry(spvo_ctx *c, const spvo_solve_input *in, spvo_solve_output *out, float *xyz, int32_t *inliers) {
  if (!c || !in || !out || (in->n > 0 && (!xyz || !inliers))) return fail(c, SPVO_ERR_INVALID, "null argument");
  const int rc = spvo_solve_submit(c, in);
  return rc ? rc : spvo_solve_wait(c, out, xyz, inliers);
}

}  // extern "C"
