"""GPU parity of K14 (triangulation), K15 (RANSAC) and K16 (LM refinement) against the oracle."""
import os

import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import odometry as od
from spvo import capi, synth
from tests.test_oracle_cpu import _scene

pytestmark = pytest.mark.gpu


def test_triangulate(ctx_vgg, golden_dir):
    o = np.load(os.path.join(golden_dir, "oracle_odometry.npz"))
    got = ctx_vgg.triangulate(o["P_l"], o["P_r"], o["cl"], o["cr"])
    ref = od.triangulate(o["P_l"], o["P_r"], o["cl"], o["cr"])
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) <= 1e-4     # north_star tolerance
    assert np.max(np.abs(got - ref) / np.maximum(np.abs(ref), 1e-3)) <= 3e-7     # what is actually achieved: never more than two f32 units ...
    same = (got.view(np.int32) == ref.view(np.int32)).all(axis=1)
    assert same.mean() >= 0.98, same.mean()                                       # ... and the oracle's f32 bits on (nearly) every point: the homogeneous vector is rounded
    #                                                                               to f32 at the SVD's unit 2-norm scale, as cv::triangulatePoints stores it (base.cpp:212)
    assert len(ctx_vgg.triangulate(o["P_l"], o["P_r"], o["cl"][:0], o["cr"][:0])) == 0
    # bug-compatible projection matrices (denormal P[0][1]) go through unchanged
    P_l, P_r = o["P_l"].copy(), o["P_r"].copy()
    P_l[0, 1] = P_r[0, 1] = 1.5e-314
    assert np.allclose(ctx_vgg.triangulate(P_l, P_r, o["cl"], o["cr"]), ref, rtol=1e-5)


def test_triangulate_unfiltered_correspondences(ctx_vgg, golden_dir):
    """spvo_triangulate on what the pipeline's own filters (y threshold, minimum disparity: base.cpp:127-207) would have dropped:
    almost-zero disparity (the null vector's w is ~0, nearly orthogonal to the inverse iteration's start vector) and large vertical
    offsets (sigma_4 / sigma_3 not small).  The kernel iterates until the direction stands still; the result is the SVD's null
    vector (oracle: numpy.linalg.svd), compared as a direction -- the dehomogenised point of such a pair is ill-conditioned itself."""
    o = np.load(os.path.join(golden_dir, "oracle_odometry.npz"))
    P_l, P_r = o["P_l"], o["P_r"]
    rng = np.random.RandomState(3)
    n = 400
    cl = np.stack([rng.uniform(50, 1100, n), rng.uniform(20, 340, n)], 1).astype(np.float32)
    disp = np.concatenate([np.full(100, 1e-3), np.full(100, 0.02), rng.uniform(0.25, 60, 200)]).astype(np.float32)
    dy = np.concatenate([np.zeros(200), rng.uniform(-40, 40, 200)]).astype(np.float32)
    cr = np.stack([cl[:, 0] - disp, cl[:, 1] + dy], 1).astype(np.float32)
    got = ctx_vgg.triangulate(P_l, P_r, cl, cr).astype(np.float64)
    assert np.isfinite(got).all()
    for i in range(n):
        A = np.stack([cl[i, 0] * P_l[2] - P_l[0], cl[i, 1] * P_l[2] - P_l[1], cr[i, 0] * P_r[2] - P_r[0], cr[i, 1] * P_r[2] - P_r[1]]).astype(np.float64)
        v = np.linalg.svd(A)[2][-1]
        g = np.append(got[i], 1.0)
        cosang = abs(g @ v) / (np.linalg.norm(g) * np.linalg.norm(v))
        assert 1.0 - cosang <= 1e-9, (i, disp[i], dy[i], 1.0 - cosang)     # the same direction (f32 rounding of the stored point)


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_ransac_matches_oracle(ctx_vgg, seed):
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(seed=seed, n=400, noise=0.3, outliers=0.3)
    pts = od.triangulate(P_l, P_r, cl, cr)
    prior_r, prior_t = np.array([0.0, 0.0, 0.0]), np.array([0.0, 0.0, 0.8])
    ok, r, t, inl = ctx_vgg.pnp_ransac(P_l[:, :3], pts, pl, prior_r, prior_t, 500, 2.0, seed)
    rok, rr, rt, rinl = od.pnp_ransac(P_l[:, :3], pts, pl, prior_r, prior_t, 500, 2.0, seed)
    assert ok == rok and np.array_equal(inl, rinl)                       # integer output: bit-exact
    assert np.allclose(r, rr, atol=1e-8) and np.allclose(t, rt, atol=1e-8)
    assert np.allclose(r, rv, atol=3e-3) and np.allclose(t, tv, atol=3e-2)


def test_ransac_degenerate_inputs(ctx_vgg):
    P_l, P_r, Xc, cl, cr, pl, *_ = _scene(n=30)
    pts = od.triangulate(P_l, P_r, cl, cr)
    ok, r, t, inl = ctx_vgg.pnp_ransac(P_l[:, :3], pts[:3], pl[:3], np.zeros(3), np.ones(3), 500, 2.0, 0)
    assert not ok and len(inl) == 0 and np.all(t == 1)                   # fewer than 4 points: prior is kept
    # pure garbage correspondences: oracle and GPU agree on failure or on the same (small) consensus
    rng = np.random.RandomState(0)
    junk = rng.uniform(0, 1000, (30, 2)).astype(np.float32)
    g = ctx_vgg.pnp_ransac(P_l[:, :3], pts, junk, np.zeros(3), np.zeros(3), 200, 2.0, 1)
    o = od.pnp_ransac(P_l[:, :3], pts, junk, np.zeros(3), np.zeros(3), 200, 2.0, 1)
    assert g[0] == o[0] and np.array_equal(g[3], o[3])


def _obs_from(P_l, P_r, pts, pl, pr, cl, cr, Xp, inl, degree):
    X, uv, cam, inv = [], [], [], []
    for i in inl:
        X.append(pts[i]); uv.append(pl[i]); cam.append(0); inv.append(0)
        if degree >= 2:
            X.append(pts[i]); uv.append(pr[i]); cam.append(1); inv.append(0)
        if degree >= 3:
            X.append(Xp[i]); uv.append(cl[i]); cam.append(0); inv.append(1)
        if degree >= 4:
            X.append(Xp[i]); uv.append(cr[i]); cam.append(1); inv.append(1)
    return (np.asarray(X, np.float32), np.asarray(uv, np.float32), np.asarray(cam), np.asarray(inv))


@pytest.mark.parametrize("degree", [1, 2, 4])
def test_refine_matches_oracle(ctx_vgg, degree):
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(seed=3, n=300, noise=0.4, outliers=0.1)
    pts = od.triangulate(P_l, P_r, cl, cr)
    R = od.quat_to_rot(od.rvec_to_quat(rv))
    Xp = (Xc @ R.T + tv).astype(np.float32)
    inl = np.nonzero(~bad)[0]
    X, uv, cam, inv = _obs_from(P_l, P_r, pts, pl, pr, cl, cr, Xp, inl, degree)
    q0 = od.rvec_to_quat(rv + 0.01)
    t0 = tv + 0.05
    q, t, s = ctx_vgg.pnp_refine(P_l, P_r, capi.obs_array(X, uv, cam, inv), q0, t0)
    rq, rt, rs = od.pnp_refine(P_l, P_r, (X.astype(np.float64), uv.astype(np.float64), cam, inv), q0, t0)
    assert (s.iterations, bool(s.converged), bool(s.usable)) == (rs.iterations, rs.converged, rs.usable)
    assert np.allclose(q, rq, atol=1e-9) and np.allclose(t, rt, atol=1e-9)     # far inside the 1e-4 bar
    assert s.initial_cost == pytest.approx(rs.initial_cost, rel=1e-10)
    assert s.final_cost == pytest.approx(rs.final_cost, rel=1e-9)
    assert np.allclose(od.quat_to_rvec(q), rv, atol=2e-3) and np.allclose(t, tv, atol=2e-2)


def test_refine_edge_cases(ctx_vgg):
    P_l, P_r = synth.projection_matrices()
    q, t, s = ctx_vgg.pnp_refine(P_l, P_r, capi.obs_array([], [], [], []), [0, 0, 0, 1], [0, 0, 0])
    assert s.converged and s.usable and np.allclose(q, [0, 0, 0, 1])
    # max_iterations = 1 -> NO_CONVERGENCE (the reference then discards the result, base.cpp:366-374)
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(seed=4, n=80)
    pts = od.triangulate(P_l, P_r, cl, cr)
    obs = capi.obs_array(pts, pl, np.zeros(80, int), np.zeros(80, int))
    q, t, s = ctx_vgg.pnp_refine(P_l, P_r, obs, [0, 0, 0, 1], [0, 0, 0], max_iterations=1)
    _, _, rs = od.pnp_refine(P_l, P_r, (pts.astype(np.float64), pl.astype(np.float64), np.zeros(80, int), np.zeros(80, int)),
                             np.array([0, 0, 0, 1.0]), np.zeros(3), max_iterations=1)
    assert not s.converged and s.usable and s.iterations == rs.iterations == 1


@pytest.mark.parametrize("degree,with_prev", [(4, True), (4, False), (2, True), (0, True)])
def test_fused_solve_equals_staged_calls(ctx_vgg, degree, with_prev):
    """spvo_solve_stereo_odometry == triangulate + ransac + host gating + refine, to the last bit."""
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(seed=5, n=350, noise=0.3, outliers=0.25)
    R = od.quat_to_rot(od.rvec_to_quat(rv))
    Xp = (Xc @ R.T + tv).astype(np.float32)
    rng = np.random.RandomState(0)
    pvalid = (rng.rand(350) < 0.7).astype(np.int32)
    prior_r, prior_t = np.array([0.0, 0.0, 0.0]), np.array([0.0, 0.0, 0.9])
    f = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, Xp if with_prev else None, pvalid if with_prev else None,
                      prior_r, prior_t, frame_count=3, refinement_degree=degree, seed=2)
    pts = ctx_vgg.triangulate(P_l, P_r, cl, cr)
    ok, r, t, inl = ctx_vgg.pnp_ransac(P_l[:, :3], pts, pl, prior_r, prior_t, 500, 2.0, 2)
    assert np.array_equal(f["xyz"], pts) and f["pnp_ok"] == ok and np.array_equal(f["inliers"], inl)
    assert np.array_equal(f["rvec"], r) and np.array_equal(f["tvec"], t) and f["accepted"]
    X, uv, cam, inv = [], [], [], []
    for i in inl:
        X.append(pts[i]); uv.append(pl[i]); cam.append(0); inv.append(0)
        if degree >= 2:
            X.append(pts[i]); uv.append(pr[i]); cam.append(1); inv.append(0)
        if with_prev and pvalid[i] and degree >= 3:
            X.append(Xp[i]); uv.append(cl[i]); cam.append(0); inv.append(1)
        if with_prev and pvalid[i] and degree >= 4:
            X.append(Xp[i]); uv.append(cr[i]); cam.append(1); inv.append(1)
    q0 = od.rvec_to_quat(r)
    if degree > 0:
        q, t2, s = ctx_vgg.pnp_refine(P_l, P_r, capi.obs_array(X, uv, cam, inv), q0, t)
        assert f["refined"] == bool(s.converged and s.usable) and f["iterations"] == s.iterations
        assert np.allclose(f["q"], q, atol=1e-12) and np.allclose(f["t"], t2, atol=1e-12)
    else:
        assert not f["refined"] and np.allclose(f["q"], q0, atol=1e-15) and np.array_equal(f["t"], t)
    # and against the oracle (same bars as the staged tests)
    ook, orr, ot, oinl = od.pnp_ransac(P_l[:, :3], od.triangulate(P_l, P_r, cl, cr), pl, prior_r, prior_t, 500, 2.0, 2)
    assert np.array_equal(f["inliers"], oinl) and np.allclose(f["tvec"], ot, atol=1e-8)


def test_fused_solve_gating_and_small_inputs(ctx_vgg):
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(seed=6, n=200, noise=0.2)
    # acceleration gate (base.cpp:251-260): prior far from the estimate, frame_count > 10 -> prior is returned
    prior_t = np.array([0.0, 0.0, 5.0])
    f = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, np.zeros(3), prior_t, frame_count=11)
    assert f["pnp_ok"] and not f["accepted"] and not f["refined"]
    assert np.allclose(f["t"], prior_t) and np.allclose(f["q"], [0, 0, 0, 1])
    # ... but not during the first IGNORE_FRAME_COUNT frames
    f = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, np.zeros(3), prior_t, frame_count=10)
    assert f["accepted"] and f["refined"] and np.allclose(f["t"], tv, atol=2e-2)
    # fewer than 4 correspondences / none at all: no model, the prior comes back (base.cpp:244-250)
    for k in (0, 3):
        f = ctx_vgg.solve(P_l, P_r, cl[:k], cr[:k], pl[:k], pr[:k], None, None, [0, 0.1, 0], [1, 2, 3])
        assert not f["pnp_ok"] and not f["accepted"] and len(f["inliers"]) == 0 and np.allclose(f["t"], [1, 2, 3])
        assert np.allclose(f["q"], od.rvec_to_quat([0, 0.1, 0]))


def test_solve_in_two_halves(ctx_vgg):
    """spvo_solve_submit + spvo_solve_wait == spvo_solve_stereo_odometry bit for bit; the inputs may be overwritten between the
    halves (they are staged at submit); THREE solves may be pending (round 6), a fourth is refused, and the stand-alone solver entry
    points refuse while any is."""
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(seed=9, n=300, noise=0.3, outliers=0.2)
    prior_r, prior_t = np.zeros(3), np.array([0.0, 0.0, 0.9])
    ref = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, prior_r, prior_t, frame_count=3, seed=4)
    arrs = [a.copy() for a in (cl, cr, pl, pr)]
    n = ctx_vgg.solve(P_l, P_r, *arrs, None, None, prior_r, prior_t, frame_count=3, seed=4, split="submit")
    for a in arrs:
        a[:] = -1.0                                                     # the caller's arrays are free after submit
    n2 = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, prior_r, prior_t, frame_count=3, seed=4, split="submit")   # a second one behind it
    n3 = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, prior_r, prior_t, frame_count=3, seed=4, split="submit")   # ... and a third
    assert ctx_vgg.solve_pending() == 3
    with pytest.raises(capi.SpvoError) as e:
        ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, prior_r, prior_t, split="submit")
    assert e.value.code == -4
    with pytest.raises(capi.SpvoError) as e:
        ctx_vgg.triangulate(P_l, P_r, cl, cr)
    assert e.value.code == -4
    for got in (ctx_vgg.solve_wait(n), ctx_vgg.solve_wait(n2), ctx_vgg.solve_wait(n3)):       # oldest first; all are the one-piece call's result
        for k in ("q", "t", "rvec", "tvec", "inliers", "xyz"):
            assert np.array_equal(got[k], ref[k]), k
        assert got["refined"] == ref["refined"] and got["iterations"] == ref["iterations"]
    assert ctx_vgg.solve_pending() == 0
    with pytest.raises(capi.SpvoError) as e:                            # nothing pending any more
        ctx_vgg.solve_wait(n)
    assert e.value.code == -4
    # no correspondences: nothing is enqueued, the prior comes back from the second half
    n0 = ctx_vgg.solve(P_l, P_r, cl[:0], cr[:0], pl[:0], pr[:0], None, None, [0, 0.1, 0], [1, 2, 3], split="submit")
    f = ctx_vgg.solve_wait(n0)
    assert not f["pnp_ok"] and np.allclose(f["t"], [1, 2, 3]) and np.allclose(f["q"], od.rvec_to_quat([0, 0.1, 0]))


@pytest.mark.parametrize("degree", [4, 2])
def test_two_frames_in_flight_with_late_prior_and_point_indices(ctx_vgg, degree):
    """Round 6: frame k + 1 is submitted BEFORE frame k has been waited for -- its previous-frame points referred to by index where frame
    k's submission left them on the device (prev_index), the motion prior and the frame count handed over at the wait
    (spvo_solve_wait_prior), where the gate (base.cpp:241-272) is evaluated.  Against the sequential form (wait for k, copy its points
    and its accepted pose into k + 1's input): every output bit for bit -- the device chain of a frame does not depend on the previous
    frame's pose (prior-free RANSAC, refinement from the RANSAC pose), only the gate does."""
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(seed=11, n=320, noise=0.3, outliers=0.2)
    _, _, _, cl2, cr2, pl2, pr2, rv2, tv2, _ = _scene(seed=12, n=280, noise=0.3, outliers=0.2)
    rng = np.random.RandomState(1)
    idx = rng.randint(-1, 320, 280).astype(np.int32)                    # frame B's correspondences -> points of frame A (-1: none)
    prior_r, prior_t = np.zeros(3), np.array([0.0, 0.0, 0.85])
    for fc in (3, 11):                                                  # inside IGNORE_FRAME_COUNT / with the gate armed
        # sequential reference
        a = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, prior_r, prior_t, frame_count=fc, refinement_degree=degree, seed=1)
        pr_r, pr_t = (a["rvec"], a["tvec"]) if a["accepted"] else (prior_r, prior_t)
        pxyz = np.where(idx[:, None] >= 0, a["xyz"][np.maximum(idx, 0)], 0).astype(np.float32)
        b = ctx_vgg.solve(P_l, P_r, cl2, cr2, pl2, pr2, pxyz, (idx >= 0).astype(np.int32), pr_r, pr_t, frame_count=fc + 1, refinement_degree=degree, seed=1)
        # two in flight
        na = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, refinement_degree=degree, seed=1, split="submit", late_prior=True)
        nb = ctx_vgg.solve(P_l, P_r, cl2, cr2, pl2, pr2, None, None, refinement_degree=degree, seed=1, split="submit", late_prior=True, prev_index=idx)
        with pytest.raises(capi.SpvoError):                             # a late submission has no prior of its own
            ctx_vgg.solve_wait(na)
        a2 = ctx_vgg.solve_wait_prior(na, prior_r, prior_t, fc)
        pr_r2, pr_t2 = (a2["rvec"], a2["tvec"]) if a2["accepted"] else (prior_r, prior_t)
        b2 = ctx_vgg.solve_wait_prior(nb, pr_r2, pr_t2, fc + 1)
        for x, y in ((a, a2), (b, b2)):
            for k in ("q", "t", "rvec", "tvec", "inliers", "xyz"):
                assert np.array_equal(x[k], y[k]), (fc, k)
            assert (x["pnp_ok"], x["accepted"], x["refined"], x["iterations"]) == (y["pnp_ok"], y["accepted"], y["refined"], y["iterations"])
        assert b["accepted"] and (degree < 3 or (idx >= 0).any())
    # the gate at the wait: the same submission accepted or rejected by the prior it is GIVEN
    n1 = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, refinement_degree=degree, seed=1, split="submit", late_prior=True)
    f = ctx_vgg.solve_wait_prior(n1, np.zeros(3), [0, 0, 5.0], 11)
    assert f["pnp_ok"] and not f["accepted"] and not f["refined"] and np.allclose(f["t"], [0, 0, 5.0]) and f["iterations"] == 0
    # indices outside the previous submission's points are refused, and so is prev_index together with prev_xyz
    bad_idx = idx.copy(); bad_idx[0] = 320
    with pytest.raises(capi.SpvoError):
        ctx_vgg.solve(P_l, P_r, cl2, cr2, pl2, pr2, None, None, split="submit", late_prior=True, prev_index=bad_idx)
    assert ctx_vgg.solve_pending() == 0


@pytest.mark.parametrize("keep", [1, 2])
def test_pipelined_solves_with_the_tail_in_the_next_submissions_launch(ctx_vgg, keep):
    """Round 6: the tail kernel of a late_prior = 2 submission is held back and goes out in ONE launch with the hypotheses of the next submission
    (solve_hyp_tail_kernel: workgroup 0 = the older solve's selection / residual blocks / refinement, the other workgroups = eight
    hypotheses each of the newer one), or alone when the solve is waited for first.  Seven frames pipelined with `keep` solves left pending
    behind every submit (1: every tail goes out alone at its wait or fused, as it comes; 2: three pending, every tail but the last two fused)
    against the one-piece call sequence: every output bit for bit, whichever way a tail was launched."""
    scenes = [_scene(seed=20 + k, n=260 + 10 * k, noise=0.3, outliers=0.2) for k in range(7)]
    rng = np.random.RandomState(3)
    idxs = [None] + [rng.randint(-1, 260 + 10 * (k - 1), 260 + 10 * k).astype(np.int32) for k in range(1, 7)]
    # the one-piece reference: wait for k, copy its points and its accepted pose into k + 1's input
    want, prior, prev = [], (np.zeros(3), np.array([0.0, 0.0, 0.85])), None
    for k, (P_l, P_r, _, cl, cr, pl, pr, _, _, _) in enumerate(scenes):
        pxyz = pval = None
        if k:
            pxyz = np.where(idxs[k][:, None] >= 0, prev["xyz"][np.maximum(idxs[k], 0)], 0).astype(np.float32)
            pval = (idxs[k] >= 0).astype(np.int32)
        o = ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, pxyz, pval, prior[0], prior[1], frame_count=11 + k, refinement_degree=4, seed=1)
        if o["accepted"]: prior = (o["rvec"], o["tvec"])
        want.append(o); prev = o
    got, pend, prior = [], [], (np.zeros(3), np.array([0.0, 0.0, 0.85]))

    def collect():
        nonlocal prior
        o = ctx_vgg.solve_wait_prior(pend.pop(0), prior[0], prior[1], 11 + len(got))
        if o["accepted"]: prior = (o["rvec"], o["tvec"])
        got.append(o)

    for k, (P_l, P_r, _, cl, cr, pl, pr, _, _, _) in enumerate(scenes):
        pend.append(ctx_vgg.solve(P_l, P_r, cl, cr, pl, pr, None, None, refinement_degree=4, seed=1, split="submit", late_prior=2, prev_index=idxs[k]))
        assert ctx_vgg.solve_pending() == len(pend)
        while len(pend) > keep:
            collect()
    while pend:
        collect()
    assert ctx_vgg.solve_pending() == 0 and sum(o["accepted"] for o in want) >= 5
    for x, y in zip(want, got):
        for f in ("q", "t", "rvec", "tvec", "inliers", "xyz"):
            assert np.array_equal(x[f], y[f]), f
        assert (x["pnp_ok"], x["accepted"], x["refined"], x["iterations"]) == (y["pnp_ok"], y["accepted"], y["refined"], y["iterations"])
