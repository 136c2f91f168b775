"""CPU tests of the oracle: known-answer cases and the committed golden pins.

The reference holds no tests or golden vectors for this path,
so the pins are (a) hand-checkable known answers of each restated function and
(b) fixtures frozen from the oracle itself by tests/golden/make_golden.py.
"""
import os

import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import frontend as fe, matching, net, odometry as od
from spvo import synth, weights


# ------------------------------------------------------------------ preprocessing
def test_crop_geometry_table():
    # SURVEY.md appendix B (float32 arithmetic of base.cpp:75-119)
    assert fe.crop_geometry(376, 1241, 360, 1176)[:4] == (0, 6, 376, 1228)
    assert fe.crop_geometry(376, 1241, 240, 784)[:4] == (0, 6, 376, 1228)
    assert fe.crop_geometry(376, 1241, 376, 1240)[:4] == (0, 0, 376, 1240)
    assert fe.crop_geometry(376, 1241, 192, 640)[:4] == (2, 0, 372, 1241)
    assert fe.crop_geometry(375, 1242, 360, 1176)[:4] == (0, 8, 375, 1225)
    assert fe.crop_geometry(370, 1226, 360, 1176)[:4] == (0, 9, 370, 1208)
    assert abs(float(fe.crop_geometry(376, 1241, 360, 1176)[4]) - 0.9576547) < 1e-7


def test_resize_identity_and_constant():
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, (37, 53)).astype(np.uint8)
    assert np.array_equal(fe.resize_linear_u8(img, 37, 53), img)
    flat = np.full((40, 60), 77, np.uint8)
    assert np.all(fe.resize_linear_u8(flat, 23, 31) == 77)


def test_resize_2x_upsample_known_values():
    # 1-D ramp 0,100 upsampled x2: centres at -0.25, 0.25, 0.75, 1.25 -> 0, 25, 75, 100
    img = np.array([[0, 100]], np.uint8).repeat(2, 0)
    out = fe.resize_linear_u8(img, 2, 4)
    assert out[0].tolist() == [0, 25, 75, 100]


def test_projection_fixup_bug_compat_and_fixed():
    P = np.array([[718.856, 0, 607.1928, -386.1448], [0, 718.856, 185.2157, 0], [0, 0, 1, 0]])
    img = np.zeros((376, 1241), np.uint8)
    _, Pb = fe.preprocess(img, P, 360, 1176, bug_compat=True)
    _, Pf = fe.preprocess(img, P, 360, 1176, bug_compat=False)
    s = float(np.float32(1176) / np.float32(1228))
    # reference behaviour: cx is NOT shifted (at<float> on a CV_64F matrix), P[0][1] turns into a denormal
    assert Pb[0, 2] == pytest.approx(607.1928 * s, rel=1e-12)
    assert 0 < abs(Pb[0, 1]) < 1e-300
    assert Pf[0, 2] == pytest.approx((607.1928 - 6.0) * s, rel=1e-12)
    assert Pf[0, 1] == 0.0
    assert Pb[0, 0] == pytest.approx(718.856 * s) and Pb[0, 3] == pytest.approx(-386.1448 * s)
    assert Pb[2].tolist() == [0, 0, 1, 0]


# ------------------------------------------------------------------ detector post-processing
def test_heatmap_known_answer():
    det = np.zeros((65, 2, 3), np.float32)
    heat = fe.heatmap(det)
    assert heat.shape == (16, 24)
    assert np.allclose(heat, 1.0 / (65.0 + 1e-5), rtol=1e-6)
    det[8 * 3 + 5, 1, 2] = 10.0                       # channel 8u+v -> pixel (8i+u, 8j+v)
    heat = fe.heatmap(det)
    assert np.unravel_index(np.argmax(heat), heat.shape) == (8 * 1 + 3, 8 * 2 + 5)


def test_nms_greedy_known_answers():
    heat = np.zeros((40, 40), np.float32)
    heat[10, 10], heat[12, 12], heat[10, 15], heat[30, 30] = 0.9, 0.8, 0.7, 0.5
    xy = fe.nms(heat, 0.015, 4, 4, 1000)
    # (12,12) is inside the 9x9 window of (10,10); (15,10) is 5 away -> kept
    assert xy.tolist() == [[10, 10], [15, 10], [30, 30]]
    # border: a strong point at the edge suppresses but is not emitted (nn.cpp:239-254)
    heat2 = np.zeros((40, 40), np.float32)
    heat2[2, 20], heat2[5, 21] = 0.9, 0.8
    assert fe.nms(heat2, 0.015, 4, 4, 1000).tolist() == []
    # threshold is strict (nn.cpp:203)
    heat3 = np.full((16, 16), 0.015, np.float32)
    assert len(fe.nms(heat3, 0.015, 4, 4, 1000)) == 0
    # ties: (confidence desc, column-major index asc) -> smaller x first, then smaller y
    heat4 = np.zeros((40, 40), np.float32)
    heat4[20, 8] = heat4[8, 20] = heat4[8, 8] = 0.5
    assert fe.nms(heat4, 0.015, 4, 4, 1000).tolist() == [[8, 8], [8, 20], [20, 8]]
    # cap
    assert len(fe.nms(heat4, 0.015, 4, 4, 2)) == 2


def test_sample_descriptors_on_grid_nodes():
    rng = np.random.RandomState(1)
    desc = rng.randn(256, 5, 7).astype(np.float32)
    H, W = 40, 56
    # pixel (col=0,row=0) maps to the coarse node (0,0) exactly -> the normalised node itself
    out = fe.sample_descriptors(desc, np.array([[0, 0]]), H, W)
    ref = desc[:, 0, 0] / np.linalg.norm(desc[:, 0, 0])
    assert np.allclose(out[0], ref, atol=1e-6)
    assert np.allclose(np.linalg.norm(fe.sample_descriptors(desc, np.array([[13, 9], [50, 30]]), H, W), axis=1), 1, atol=1e-6)


# ------------------------------------------------------------------ matcher
def test_matcher_known_answers():
    a = np.zeros((3, 256), np.float32)
    b = np.zeros((4, 256), np.float32)
    a[0, 0] = a[1, 1] = a[2, 2] = 1
    b[0, 1] = 1
    b[1, 0] = 1
    b[2, 0] = 1                                       # duplicate of b[1]: lowest train index wins
    b[3, 5] = 1
    idx, d = matching.bf_match(a, b, "NN", False)
    assert idx.tolist() == [1, 0, 0] and d[0] == 0 and d[2] == pytest.approx(np.sqrt(2))
    idx, _ = matching.bf_match(a, b, "NN", True)
    assert idx.tolist() == [1, 0, -1]                 # query 2 loses train 0 to query 1 (d=0 < sqrt2)
    idx, _ = matching.bf_match(a, b, "KNN", False, 0.8)
    assert idx.tolist() == [-1, 0, -1]                # a[0]: d0 = d1 = 0 (tie) fails the ratio test
    idx, _ = matching.bf_match(a, b[:1], "KNN", False, 0.8)
    assert idx.tolist() == [-1, -1, -1]               # a single train row has no second neighbour
    idx, _ = matching.bf_match(a, b[:0], "NN", False)
    assert idx.tolist() == [-1, -1, -1]


# ------------------------------------------------------------------ odometry
def _scene(seed=0, n=150, noise=0.0, outliers=0.0):
    rng = np.random.RandomState(seed)
    P_l, P_r = synth.projection_matrices()
    Xc = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n), rng.uniform(5, 40, n)], 1)
    rv, tv = np.array([0.01, -0.02, 0.005]), np.array([0.05, -0.02, 1.0])
    R = od.quat_to_rot(od.rvec_to_quat(rv))
    Xp = Xc @ R.T + tv

    def proj(P, X):
        p = X @ P[:, :3].T + P[:, 3]
        return (p[:, :2] / p[:, 2:] + noise * rng.randn(len(X), 2)).astype(np.float32)

    cl, cr, pl, pr = proj(P_l, Xc), proj(P_r, Xc), proj(P_l, Xp), proj(P_r, Xp)
    bad = rng.rand(n) < outliers
    pl[bad] += rng.uniform(-50, 50, (bad.sum(), 2)).astype(np.float32)
    return P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad


def test_triangulate_recovers_points():
    P_l, P_r, Xc, cl, cr, *_ = _scene()
    pts = od.triangulate(P_l, P_r, cl, cr)
    # float32 pixel coordinates limit the depth accuracy: relative error ~ 1e-4 at 40 m
    assert np.max(np.abs(pts - Xc) / np.abs(Xc).max(1, keepdims=True)) < 2e-3


def test_ransac_and_refine_recover_pose():
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, bad = _scene(noise=0.2, outliers=0.25)
    pts = od.triangulate(P_l, P_r, cl, cr)
    ok, r, t, inl = od.pnp_ransac(P_l[:, :3], pts, pl, np.zeros(3), np.zeros(3), 500, 2.0, 0)
    assert ok and set(inl.tolist()) <= set(np.nonzero(~bad)[0].tolist()) and len(inl) > 0.9 * (~bad).sum()
    assert np.allclose(r, rv, atol=2e-3) and np.allclose(t, tv, atol=2e-2)
    obs = (pts[inl].astype(np.float64), pl[inl].astype(np.float64), np.zeros(len(inl), int), np.zeros(len(inl), int))
    q, t2, s = od.pnp_refine(P_l, P_r, obs, od.rvec_to_quat(r), t)
    assert s.converged and s.usable and s.final_cost <= s.initial_cost
    assert np.allclose(od.quat_to_rvec(q), rv, atol=2e-3)


def test_refine_inverse_blocks_and_nonconvergence_rule():
    P_l, P_r, Xc, cl, cr, pl, pr, rv, tv, _ = _scene(n=60)
    R = od.quat_to_rot(od.rvec_to_quat(rv))
    Xp = Xc @ R.T + tv
    # inverse blocks: prev-frame 3-D points observed in the current frame (cost.hpp:35-37)
    obs = (np.concatenate([Xc, Xp]), np.concatenate([pl, cl]).astype(np.float64),
           np.zeros(120, int), np.concatenate([np.zeros(60, int), np.ones(60, int)]))
    q, t, s = od.pnp_refine(P_l, P_r, obs, np.array([0, 0, 0, 1.0]), np.zeros(3))
    assert s.converged and np.allclose(t, tv, atol=1e-3) and np.allclose(od.quat_to_rvec(q), rv, atol=1e-4)
    # hitting max_num_iterations is NO_CONVERGENCE -> the reference discards the result (base.cpp:366-374)
    _, _, s1 = od.pnp_refine(P_l, P_r, obs, np.array([0, 0, 0, 1.0]), np.zeros(3), max_iterations=1)
    assert not s1.converged and s1.usable


def test_sample_triplets_distinct_and_deterministic():
    for it in range(50):
        s = od.sample_triplet(0, it, 5)
        assert len(set(s)) == 3 and s == od.sample_triplet(0, it, 5)
    assert od.hash32(1) == 0x6C6D9A5F or isinstance(od.hash32(1), int)


# ------------------------------------------------------------------ golden pins
def test_golden_frontend(golden_dir, sample_images, squeeze_plan):
    g = np.load(os.path.join(golden_dir, "oracle_frontend_squeeze_120x392.npz"))
    P_l, _ = synth.projection_matrices()
    r = fe.detect(squeeze_plan, sample_images[0], P_l, 120, 392)
    assert np.array_equal(r["resized"], g["resized"])
    assert np.array_equal(r["P"], g["P"])
    assert np.allclose(r["det"], g["det"].astype(np.float32), atol=2e-2, rtol=1e-2)
    assert float(r["heat"].astype(np.float64).sum()) == pytest.approx(float(g["heat_sum"]), rel=1e-4)
    same = np.mean([tuple(a) in set(map(tuple, g["xy"].tolist())) for a in r["xy"].tolist()])
    assert same > 0.97                                 # torch CPU conv may differ in the last bits across hosts
    assert len(r["xy"]) == pytest.approx(len(g["xy"]), abs=5)


def test_golden_vgg_and_match_and_odometry(golden_dir, vgg_plan):
    g = np.load(os.path.join(golden_dir, "oracle_vgg_64x64.npz"))
    det, desc = net.forward(vgg_plan, g["x"])
    assert np.allclose(det, g["det"], atol=1e-4) and np.allclose(desc, g["desc"], atol=1e-5)
    assert vgg_plan.n_params() == 1300865             # reference report, Table 1
    m = np.load(os.path.join(golden_dir, "oracle_match.npz"))
    idx, d = matching.bf_match(m["a"], m["b"], "KNN", False, 0.8)
    assert np.array_equal(idx, m["knn_idx"]) and np.array_equal(d, m["knn_d"])
    idx, d = matching.bf_match(m["a"], m["b"], "NN", True, 0.8)
    assert np.array_equal(idx, m["nn_idx"])
    o = np.load(os.path.join(golden_dir, "oracle_odometry.npz"))
    pts = od.triangulate(o["P_l"], o["P_r"], o["cl"], o["cr"])
    assert np.allclose(pts, o["pts"], rtol=1e-5)
    ok, r, t, inl = od.pnp_ransac(o["P_l"][:, :3], o["pts"], o["pl"], np.zeros(3), np.zeros(3), 500, 2.0, 0)
    assert ok == bool(o["ok"]) and np.array_equal(inl, o["inliers"])
    assert np.allclose(r, o["rvec"], atol=1e-9) and np.allclose(t, o["tvec"], atol=1e-9)
    assert np.allclose(r, o["true_rvec"], atol=3e-3) and np.allclose(t, o["true_tvec"], atol=3e-2)


def test_hamming_matching_known_answers():
    """bf_match_hamming (cv::BFMatcher(NORM_HAMMING) restated): bit counts, lowest-index ties, the ratio test on integer
    distances, and the crosscheck procedure, on cases small enough to check by hand."""
    from oracle import matching
    a = np.array([[0b00001111, 0], [0xFF, 0xFF], [0, 0]], np.uint8)
    b = np.array([[0b00001110, 0], [0b00001111, 1], [0xFF, 0x7F], [0, 0]], np.uint8)
    d = matching.hamming_distances(a, b)
    assert d.tolist() == [[1, 1, 11, 4], [13, 11, 1, 16], [3, 5, 15, 0]]
    idx, dist = matching.bf_match_hamming(a, b, "NN")
    assert idx.tolist() == [0, 2, 3] and dist.tolist() == [1.0, 1.0, 0.0]          # row 0: tie 1 == 1 -> lower train index
    idx, dist = matching.bf_match_hamming(a, b, "KNN", ratio=0.8)
    assert idx.tolist() == [-1, 2, 3]                                              # row 0: 1 < 0.8 * 1 fails; row 2: 0 < 0.8 * 3
    idx, dist = matching.bf_match_hamming(a, b, "NN", cross_check=True)
    # train rows choose queries: t0 -> q0 (1), t1 -> q0 (1), t2 -> q1 (1), t3 -> q2 (0); q0 keeps the first of its two voters
    assert idx.tolist() == [0, 2, 3] and dist.tolist() == [1.0, 1.0, 0.0]
    assert matching.bf_match_hamming(a[:0], b)[0].shape == (0,) and matching.bf_match_hamming(a, b[:0])[0].tolist() == [-1, -1, -1]
