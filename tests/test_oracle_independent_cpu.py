"""Independent cross-checks of the oracle (CPU).

The reference's own arithmetic lives in TensorRT / OpenCV / Ceres, none of which exists in this image, and the
reference holds no golden vectors: the oracle is "parity unpinned".  What CAN be done here
is to check every restated third-party routine against a second, independently written implementation of the same
published algorithm -- scipy / numpy.linalg / torch -- so that an error would have to be made twice, in two different
code bases, to go unnoticed.  These tests do not replace reference goldens; they bound the risk of a wrong restatement.
"""
import numpy as np
import pytest
import scipy.optimize
import scipy.spatial.distance
import torch
import torch.nn.functional as F

import oracle  # noqa: F401
from oracle import frontend as fe, matching, net, odometry as od
from spvo import synth, weights


def test_matcher_against_scipy_cdist():
    """cv::BFMatcher(NORM_L2) semantics (base.cpp:27-28,462-473) on top of scipy's pairwise distances."""
    rng = np.random.RandomState(5)
    a = rng.randn(150, 256).astype(np.float32)
    b = np.concatenate([a[:90] + 0.08 * rng.randn(90, 256).astype(np.float32), rng.randn(70, 256).astype(np.float32)])
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    D = scipy.spatial.distance.cdist(a.astype(np.float64), b.astype(np.float64))          # independent, float64
    order = np.argsort(D, axis=1, kind="stable")
    # KNN k = 2 + ratio test
    idx, dist = matching.bf_match(a, b, "KNN", False, 0.8)
    best, second = order[:, 0], order[:, 1]
    keep = D[np.arange(len(a)), best] < 0.8 * D[np.arange(len(a)), second]
    margin = np.abs(D[np.arange(len(a)), best] - 0.8 * D[np.arange(len(a)), second]) > 1e-5  # skip float32-vs-float64 boundary cases
    assert np.array_equal(idx[margin] >= 0, keep[margin])
    assert np.array_equal(idx[keep & margin], best[keep & margin])
    assert np.allclose(dist[keep & margin], D[np.arange(len(a)), best][keep & margin], atol=2e-6)
    # NN + crossCheck = cv::batchDistance(crosscheck = true): every train row votes for its nearest query row, a query row
    # keeps the nearest of its voters (first on ties) -- transcribed here on scipy's float64 distances
    idx, dist = matching.bf_match(a, b, "NN", True, 0.8)
    want = np.full(len(a), -1)
    bestd = np.full(len(a), np.inf)
    for t in range(len(b)):
        q = int(np.argmin(D[:, t]))
        if D[q, t] < bestd[q]:
            bestd[q], want[q] = D[q, t], t
    assert np.array_equal(idx, want)
    assert np.allclose(dist[want >= 0], bestd[want >= 0], atol=2e-6) and np.all(dist[want < 0] == 0)
    mutual = (np.argmin(D, axis=0)[best] == np.arange(len(a)))
    assert np.all(idx[mutual] == best[mutual])                                          # every mutual pair is among the results


def test_triangulation_against_numpy_svd():
    """cv::triangulatePoints = DLT: the null vector of the 4x4 system, here from numpy.linalg.svd (LAPACK)."""
    P_l, P_r = synth.projection_matrices()
    rng = np.random.RandomState(2)
    X = np.stack([rng.uniform(-8, 8, 60), rng.uniform(-2, 2, 60), rng.uniform(4, 50, 60)], 1)
    proj = lambda P: ((X @ P[:, :3].T + P[:, 3]) / (X @ P[2, :3] + P[2, 3])[:, None])[:, :2].astype(np.float32)
    xl, xr = proj(P_l) + 0.2 * rng.randn(60, 2).astype(np.float32), proj(P_r) + 0.2 * rng.randn(60, 2).astype(np.float32)
    got = od.triangulate(P_l, P_r, xl, xr)
    for i in range(60):
        A = np.stack([xl[i, 0] * P_l[2] - P_l[0], xl[i, 1] * P_l[2] - P_l[1], xr[i, 0] * P_r[2] - P_r[0], xr[i, 1] * P_r[2] - P_r[1]]).astype(np.float64)
        v = np.linalg.svd(A)[2][-1]
        assert np.allclose(got[i], v[:3] / v[3], rtol=2e-4, atol=1e-4)


def _scene(seed, n=120, outliers=0.0):
    P_l, P_r = synth.projection_matrices()
    rng = np.random.RandomState(seed)
    Xc = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n), rng.uniform(5, 40, n)], 1)
    rv, tv = np.array([0.012, -0.02, 0.004]), np.array([0.04, -0.01, 0.9])
    R = od.quat_to_rot(od.rvec_to_quat(rv))
    Xp = Xc @ R.T + tv
    proj = lambda P, X: (X @ P[:, :3].T + P[:, 3])[:, :2] / (X @ P[2, :3] + P[2, 3])[:, None]
    uvl, uvr = proj(P_l, Xp) + 0.3 * rng.randn(n, 2), proj(P_r, Xp) + 0.3 * rng.randn(n, 2)
    bad = rng.rand(n) < outliers
    uvl[bad] += rng.uniform(-30, 30, (bad.sum(), 2))
    obs = (np.concatenate([Xc, Xc]), np.concatenate([uvl, uvr]), np.concatenate([np.zeros(n, int), np.ones(n, int)]), np.zeros(2 * n, int))
    return P_l, P_r, obs, rv, tv


def test_refinement_minimum_is_confirmed_by_scipy():
    """The Ceres restatement (Huber loss on each 2-d block, LM) against scipy.optimize.least_squares(loss='huber') on
    the same robust cost: from the same start the oracle's cost is not higher than scipy's, and scipy started AT the
    oracle's optimum does not move away from it (first-order optimality by an independent implementation)."""
    P_l, P_r, obs, rv, tv = _scene(11, outliers=0.15)
    q0, t0 = od.rvec_to_quat(rv + 0.01), tv + 0.05

    def make_resid(qref):
        def resid(p):                              # parameters: local rotation (composed with qref) + translation
            q = od.quat_plus(qref, p[:3])
            r, _ = od.residuals_and_jacobian(P_l, P_r, obs, q, p[3:], want_jac=False)
            # scipy applies rho to each squared RESIDUAL, Ceres to each block's squared norm: feed scipy the block norms
            return np.sqrt(r[:, 0] ** 2 + r[:, 1] ** 2)
        return resid

    def cost(q, t):
        r, _ = od.residuals_and_jacobian(P_l, P_r, obs, q, t, want_jac=False)
        return 0.5 * float(np.sum(od._huber(r[:, 0] ** 2 + r[:, 1] ** 2, 1.0)[0]))

    q_o, t_o, summ = od.pnp_refine(P_l, P_r, obs, q0, t0)
    assert summ.usable and np.linalg.norm(t_o - tv) < 0.02                       # and it is the right pose
    kw = dict(loss="huber", f_scale=1.0, xtol=1e-13, ftol=1e-13, gtol=1e-13)
    sol = scipy.optimize.least_squares(make_resid(q0), np.concatenate([np.zeros(3), t0]), **kw)
    assert cost(q_o, t_o) <= cost(od.quat_plus(q0, sol.x[:3]), sol.x[3:]) * (1 + 1e-9)
    sol2 = scipy.optimize.least_squares(make_resid(q_o), np.concatenate([np.zeros(3), t_o]), **kw)
    assert np.linalg.norm(sol2.x[:3]) < 2e-5 and np.linalg.norm(sol2.x[3:] - t_o) < 2e-4
    assert cost(od.quat_plus(q_o, sol2.x[:3]), sol2.x[3:]) >= cost(q_o, t_o) * (1 - 1e-7)   # nothing lower nearby


def test_analytic_jacobian_against_finite_differences():
    P_l, P_r, obs, rv, tv = _scene(4, n=30)
    X, uv, cam, inv = obs
    inv = inv.copy()
    inv[::3] = 1                                   # mix in inverse-transform blocks (refinement degree 3 and 4)
    obs = (X, uv, cam, inv)
    q, t = od.rvec_to_quat(rv), tv
    r0, J = od.residuals_and_jacobian(P_l, P_r, obs, q, t)
    h = 1e-6
    for k in range(6):
        d = np.zeros(6)
        d[k] = h
        rp, _ = od.residuals_and_jacobian(P_l, P_r, obs, od.quat_plus(q, d[:3]), t + d[3:], want_jac=False)
        rm, _ = od.residuals_and_jacobian(P_l, P_r, obs, od.quat_plus(q, -d[:3]), t - d[3:], want_jac=False)
        assert np.allclose((rp - rm) / (2 * h), J[:, :, k], rtol=1e-5, atol=1e-4)


def test_resize_against_float_bilinear():
    """cv::resize(INTER_LINEAR) on 8-bit images is an 11-bit fixed-point version of half-pixel-centre bilinear
    interpolation; torch's float implementation of the same sampling rule must agree to one grey level."""
    rng = np.random.RandomState(0)
    img = (rng.rand(94, 307) * 255).astype(np.uint8)
    img = np.clip(np.cumsum(np.cumsum(img.astype(np.float64) - 127, 0), 1) / 40 + 128, 0, 255).astype(np.uint8)   # smooth texture
    for oh, ow in ((90, 294), (47, 160), (120, 392)):
        got = fe.resize_linear_u8(img, oh, ow).astype(np.float64)
        ref = F.interpolate(torch.from_numpy(img.astype(np.float32))[None, None], size=(oh, ow), mode="bilinear", align_corners=False)[0, 0].numpy()
        assert np.abs(got - ref).max() <= 1.0 + 1e-3


def test_descriptor_sampling_against_grid_sample():
    """bilinearInterpolationDesc (nn.cpp:366-431) = align_corners=True bilinear sampling + L2 normalisation."""
    rng = np.random.RandomState(3)
    H, W = 64, 96
    desc = rng.randn(256, H // 8, W // 8).astype(np.float32)
    xy = np.stack([rng.randint(0, W, 40), rng.randint(0, H, 40)], 1).astype(np.int32)
    got = fe.sample_descriptors(desc, xy, H, W)
    g = torch.tensor(np.stack([xy[:, 0] / (W - 1) * 2 - 1, xy[:, 1] / (H - 1) * 2 - 1], 1), dtype=torch.float32)[None, :, None, :]
    ref = F.grid_sample(torch.from_numpy(desc)[None], g, mode="bilinear", align_corners=True)[0, :, :, 0].T
    ref = F.normalize(ref, dim=1).numpy()
    assert np.abs(got - ref).max() < 2e-5


def test_network_executor_against_direct_numpy_convolution():
    """The torch executor of the plan vs a direct 7-loop-free numpy convolution of the first two VGG layers."""
    plan = weights.vgg_plan(seed=0)
    x = np.random.RandomState(1).rand(1, 1, 12, 20).astype(np.float32)
    _, _, vals = net.forward(plan, x, return_all=True)

    def conv3x3(inp, w, b):
        C, Hh, Ww = inp.shape
        p = np.pad(inp.astype(np.float64), ((0, 0), (1, 1), (1, 1)))
        out = np.zeros((w.shape[0], Hh, Ww))
        for ky in range(3):
            for kx in range(3):
                out += np.einsum("oc,chw->ohw", w[:, :, ky, kx].astype(np.float64), p[:, ky:ky + Hh, kx:kx + Ww])
        return np.maximum(out + b[:, None, None], 0)

    a1 = conv3x3(x[0], plan.ops[0].weight, plan.ops[0].bias)
    assert np.abs(a1 - vals[plan.ops[0].out][0]).max() < 1e-5
    a2 = conv3x3(a1, plan.ops[1].weight, plan.ops[1].bias)
    a2 = a2.reshape(64, 6, 2, 10, 2).max(axis=(2, 4))                                   # conv1b's fused 2x2 max-pool
    assert np.abs(a2 - vals[plan.ops[1].out][0]).max() < 1e-4


def test_nms_against_a_literal_transcription():
    """processOneHeatmap (nn.cpp:188-262) transcribed literally -- grid of suppression flags, visiting order by
    confidence -- against the oracle's vectorised restatement."""
    rng = np.random.RandomState(9)
    H, W, dist, border, cap = 40, 56, 4, 4, 1000
    heat = rng.rand(H, W).astype(np.float32) * 0.05
    ys, xs = np.nonzero(heat > 0.015)
    conf = heat[ys, xs]
    order = np.lexsort((xs * H + ys, -conf))                                            # confidence desc, column-major index asc
    suppressed = np.zeros((H, W), bool)
    out = []
    for i in order:
        y, x = ys[i], xs[i]
        if suppressed[y, x]:
            continue
        if border <= y < H - border and border <= x < W - border:
            out.append((x, y))
        suppressed[max(0, y - dist):y + dist + 1, max(0, x - dist):x + dist + 1] = True
        if len(out) >= cap:
            break
    assert np.array_equal(fe.nms(heat, 0.015, dist, border, cap), np.array(out, np.int32))


@pytest.mark.parametrize("name", ["sp_squeeze", "sp_mbv1", "sp_mbv2"])
def test_plan_evaluations_match_the_direct_onnx_evaluation(name, golden_dir):
    """oracle/net.py and the C++ restatement execute the product's execution PLAN (ReLU / BatchNorm / Add / MaxPool folded into
    the producing convolution, Concat as channel offsets).  The fixtures hold what the reference's ONNX graph gives when it is
    evaluated node by node by an interpreter that shares nothing with the packer (oracle/onnx_direct.py, frozen by
    tests/golden/make_onnx_direct_golden.py): the plan's semantics -- epsilon and position of BatchNorm, pads, Concat order,
    the ReduceL2 / Div tail -- are pinned to the graph's."""
    import os
    from oracle import cpu_backend, net
    from spvo import weights
    g = np.load(os.path.join(golden_dir, f"onnx_direct_{name}_64x96.npz"))
    path = os.path.join(golden_dir, name + ".spvw")
    det, desc = net.forward(weights.load(path), g["x"])
    assert np.abs(det - g["det"]).max() <= 2e-5 * max(1.0, np.abs(g["det"]).max())
    assert np.abs(desc - g["desc"]).max() <= 2e-5
    c = cpu_backend.CpuBackend(net_height=64, net_width=96)
    c.load_weights(path)
    det2, desc2 = c.forward(g["x"])
    c.close()
    assert np.abs(det2 - g["det"]).max() <= 1e-4 * max(1.0, np.abs(g["det"]).max()) and np.abs(desc2 - g["desc"]).max() <= 1e-4


def test_direct_onnx_fixtures_are_current():
    """where the reference tree is present (the build container), the fixtures equal a fresh direct evaluation"""
    import os
    models = "/root/reference/src/odml_visual_odometry/models"
    if not os.path.isdir(models):
        pytest.skip("reference tree not present")
    from oracle import onnx_direct
    from tests.conftest import GOLDEN
    for name in ("sp_squeeze", "sp_mbv1", "sp_mbv2"):
        g = np.load(os.path.join(GOLDEN, f"onnx_direct_{name}_64x96.npz"))
        out = onnx_direct.run(os.path.join(models, name + "_b1.onnx"), g["x"])
        # (torch's CPU convolution picks its blocking by thread count: bitwise repeatability is not guaranteed)
        assert np.abs(out["output_det"] - g["det"]).max() <= 2e-5 and np.abs(out["output_desc"] - g["desc"]).max() <= 2e-6
        out2 = onnx_direct.run(os.path.join(models, name + "_b2.onnx"), np.concatenate([g["x"], g["x"]]))   # b2 graphs: same weights
        assert np.abs(out2["output_det"][1] - g["det"][0]).max() <= 2e-5
