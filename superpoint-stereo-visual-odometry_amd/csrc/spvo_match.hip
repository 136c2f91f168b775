// spvo_match.hip -- descriptor matching (matchDescriptors, feature_detection_base.cpp:434-500): L2 (K12, K13) and Hamming (K12h).
#include "spvo_internal.hip.h"
#include "match.hip.h"

namespace spvo_int {

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
  for (auto &set : c->ms)
    for (auto &m : set) {
      drop(m.d_na); drop(m.d_nb); drop(m.d_best_d2); drop(m.d_dt); drop(m.d_cand); drop(m.d_meta); drop(m.d_best_idx); drop(m.d_train_best); drop(m.d_a8); drop(m.d_b8); drop(m.d_qa8); drop(m.d_qb8);
      m.d_out = nullptr;
    }
  for (auto &p : c->h_match_out) { if (p) (void)hipHostFree(p); p = nullptr; }
  if (c->h_match_tmp) (void)hipHostFree(c->h_match_tmp);
  c->h_match_tmp = nullptr;
  for (auto &set : c->mcache) for (auto &mc : set) { mc.valid = false; mc.h_out = nullptr; }
  int rc;
  if ((rc = dev_alloc(c, &c->d_ma, (size_t)cap * MATCH_D))) return rc;
  if ((rc = dev_alloc(c, &c->d_mb, (size_t)cap * MATCH_D))) return rc;
  if ((rc = dev_alloc(c, &c->d_match_out, (size_t)4 * cap))) return rc;
  for (int k4 = 0; k4 < 4; ++k4) {
    const int k = k4 & 1;
    MatchScratch &m = c->ms[k4 >> 1][k];
    if ((rc = dev_alloc(c, &m.d_na, cap + 4))) return rc;   // K12b reads the norms four at a time
    if ((rc = dev_alloc(c, &m.d_nb, cap + 4))) return rc;
    if ((rc = dev_alloc(c, &m.d_best_d2, (size_t)cap * 2))) return rc;
    if ((rc = dev_alloc(c, &m.d_dt, (size_t)cap * match_ldt(cap)))) return rc;
    const size_t nt = (size_t)(cap + MATCH_TT - 1) / MATCH_TT;
    if ((rc = dev_alloc(c, &m.d_cand, (size_t)cap * nt * MATCH_C))) return rc;
    if ((rc = dev_alloc(c, &m.d_meta, (size_t)cap * nt))) return rc;
    if ((rc = dev_alloc(c, &m.d_best_idx, (size_t)cap * 2))) return rc;
    if ((rc = dev_alloc(c, &m.d_train_best, cap))) return rc;
    if ((rc = dev_alloc(c, &m.d_a8, (size_t)cap * MATCH_D))) return rc;
    if ((rc = dev_alloc(c, &m.d_b8, (size_t)cap * MATCH_D))) return rc;
    if ((rc = dev_alloc(c, &m.d_qa8, cap))) return rc;
    if ((rc = dev_alloc(c, &m.d_qb8, cap))) return rc;
    m.d_out = c->d_match_out + (size_t)k4 * cap;
  }
  ++c->alloc_gen;
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

// Enqueue 1 or 2 matches as ONE set of launches (blockIdx.z / .y = job) and one result copy:
// packed {train_idx, distance bits} for job k lands at host_out + k*match_cap.
int enqueue_matches(spvo_ctx *c, const MatchReq *req_in, int njobs, int selector, int cross_check, float ratio, int2 *host_out) {
  MatchJobs jobs;
  int na_max = 0, nb_max = 0;
  // NN + cross-check is cv::batchDistance's crosscheck: the search runs from the TRAIN rows to the query rows, so the
  // two sides change places for the distance GEMM and the re-rank; match_select_cross_kernel writes one entry per
  // query row again
  const bool swap = selector == SPVO_SELECT_NN && cross_check;
  // host_out is pinned memory of the context (h_match_out / h_match_tmp): the kernels that decide a row write its entry there
  // themselves -- 8 bytes per row over PCIe -- instead of a 16 KB copy behind them (a 19 us blit launch between the merge and the
  // event the host waits for).
  MatchReq req[2];
  for (int k = 0; k < njobs; ++k) {
    req[k] = req_in[k];
    if (swap) {
      std::swap(req[k].dA, req[k].dB); std::swap(req[k].na, req[k].nb);
      std::swap(req[k].na_ptr, req[k].nb_ptr); std::swap(req[k].sqA, req[k].sqB);
    }
    MatchScratch &m = c->ms[c->ms_set & 1][k];
    MatchJob &j = jobs.j[k];
    j.A = req[k].dA; j.B = req[k].dB;
    j.na = req[k].na; j.nb = req[k].nb;
    j.na_ptr = req[k].na_ptr; j.nb_ptr = req[k].nb_ptr;
    j.nA = req[k].sqA ? req[k].sqA : m.d_na;
    j.nB = req[k].sqB ? req[k].sqB : m.d_nb;
    j.dt = m.d_dt; j.cand = m.d_cand; j.meta = m.d_meta; j.best_d2 = m.d_best_d2; j.best_idx = m.d_best_idx; j.train_best = m.d_train_best; j.out = host_out + (size_t)k * c->match_cap;
    j.A8 = j.B8 = nullptr;
    j.qA8 = j.qB8 = nullptr;
    if (c->match_fp8) {
      hipLaunchKernelGGL(desc_to_fp8_kernel, dim3((req[k].na + 3) / 4), dim3(256), 0, c->post, req[k].dA, req[k].na, req[k].na_ptr, m.d_a8, m.d_qa8);
      hipLaunchKernelGGL(desc_to_fp8_kernel, dim3((req[k].nb + 3) / 4), dim3(256), 0, c->post, req[k].dB, req[k].nb, req[k].nb_ptr, m.d_b8, m.d_qb8);
      j.A8 = m.d_a8; j.B8 = m.d_b8;
      j.qA8 = m.d_qa8; j.qB8 = m.d_qb8;
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
    HIP_TRY(c, hipFuncSetAttribute((const void *)match_gemm_kernel<false, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    HIP_TRY(c, hipFuncSetAttribute((const void *)match_gemm_kernel<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    attr[c->cfg.device & 63] = true;
  }
  // Default: the fused form -- the distance tile is reduced per query row in LDS (K12a) and a short merge (K12m) finishes the row;
  // dt never goes to HBM.  The fp8 shortlist mode (and tuning "match_fused" = 0, for A/B measurements) keeps the two-kernel form that
  // writes every dt and reads it back.  Column tiles are lanes of the merge: capacities beyond 64 tiles use the unfused form.
  const bool fused_on = tuning("match_fused", 1) != 0;
  const int nt_stride = (c->match_cap + MATCH_TT - 1) / MATCH_TT;
  const bool fused = fused_on && !c->match_fp8 && nt_stride <= 64;
  const int gx = groups, gy = (na_max + MATCH_QT - 1) / MATCH_QT;
  const dim3 gg(8 * ((gx * gy * njobs + 7) / 8));   // one-dimensional: XCD-contiguous chunks of the tile order (see the kernel)
  {
    ScopedStage sg(c, stage_id(c, "match_gemm"), fl, 4.0 * (na_max + nb_max) * MATCH_D * njobs);
    if (c->match_fp8) hipLaunchKernelGGL((match_gemm_kernel<true, false>), gg, dim3(256), MATCH_LDS_BYTES_FP8, c->post, jobs, ldt, 0.f, 0, gx, gy, njobs);
    else if (fused) hipLaunchKernelGGL((match_gemm_kernel<false, true>), gg, dim3(256), lds, c->post, jobs, ldt, MATCH_ERR_REL, nt_stride, gx, gy, njobs);
    else hipLaunchKernelGGL((match_gemm_kernel<false, false>), gg, dim3(256), lds, c->post, jobs, ldt, 0.f, 0, gx, gy, njobs);
  }
  {
    ScopedStage sr(c, stage_id(c, "match_rerank"));
    const float err = c->match_fp8 ? MATCH_ERR_REL_FP8 : MATCH_ERR_REL;
    const dim3 gr((na_max + 3) / 4, njobs);
    // rows of up to 1024 columns stay in registers between the two passes of the re-rank (8 chunks for 2048 columns
    // measured 2.4x SLOWER than the chunked form: 232 registers, 70 KB of LDS)
    if (fused) hipLaunchKernelGGL(match_merge_kernel<>, gr, dim3(256), sizeof(MatchRerankLds<0>), c->post, jobs, nt_stride, ldt, MATCH_ERR_REL, selector, cross_check, ratio);
    else if (nb_max <= 1024) hipLaunchKernelGGL(match_rerank_kernel<4>, gr, dim3(256), sizeof(MatchRerankLds<4>), c->post, jobs, ldt, err, selector, cross_check, ratio);
    else hipLaunchKernelGGL(match_rerank_kernel<0>, gr, dim3(256), sizeof(MatchRerankLds<0>), c->post, jobs, ldt, err, selector, cross_check, ratio);
  }
  if (swap) hipLaunchKernelGGL(match_select_cross_kernel, dim3((nb_max + 255) / 256, njobs), dim3(256), 0, c->post, jobs);
  HIP_TRY(c, hipGetLastError());
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

}  // namespace spvo_int

// ===========================================================================
extern "C" {

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
  HIP_TRY(c, hipSetDevice(c->cfg.device));
  (void)release_held_if_idle(c);   // (a failing launch is reported by that pair's spvo_detect_wait: launch_group)
  const FeatureSlot &a = c->slots[slot_a], &b = c->slots[slot_b];
  if (a.n > 0 && (!train_idx || !distance)) return fail(c, SPVO_ERR_INVALID, "null output");
  for (int set = 0; set < RING; ++set) {   // already computed alongside the detector (spvo_set_prematch)?
    bool inflight = false;
    for (const auto &q : c->pendq) inflight |= q.ring == set;
    if (inflight) continue;   // that set belongs to a submission in flight
    for (const auto &mc : c->mcache[set])
      if (mc.valid && mc.slot_a == slot_a && mc.slot_b == slot_b && mc.gen_a == a.gen && mc.gen_b == b.gen && mc.selector == selector &&
          mc.cross == (cross_check ? 1 : 0) && mc.ratio == ratio) {
        // the matches were enqueued behind the submission's features; spvo_detect_wait returned when the features were final
        const double tw0 = diag_now_us();
        HIP_TRY(c, wait_event(c->ev_tail[set]));
        g_diag.iv_match += diag_now_us() - tw0;
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

int spvo_get_match_fp8(const spvo_ctx *c) { return c ? (c->match_fp8 ? 1 : 0) : SPVO_ERR_INVALID; }

}  // extern "C"
