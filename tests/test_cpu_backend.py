"""The C++ CPU restatement (oracle/cpu: the compiled second oracle and bench.py's CPU baseline) against the numpy / torch
restatement (oracle/*.py): integer outputs bit for bit, floats within the bars the GPU tests use."""
import os

import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import cpu_backend, frontend as fe, matching, net, odometry as od
from spvo import synth, weights

H, W = 120, 392


@pytest.fixture(scope="module")
def cpu(tmp_path_factory, vgg_plan):
    p = str(tmp_path_factory.mktemp("cpu") / "vgg.spvw")
    weights.save(vgg_plan, p)
    c = cpu_backend.CpuBackend(net_height=H, net_width=W)
    c.load_weights(p)
    yield c
    c.close()


@pytest.fixture(scope="module")
def pair(golden_dir):
    frames, poses, P_l, P_r = synth.stereo_sequence(3, os.path.join(golden_dir, "images", "0000000000.png"), seed=0)
    return frames, P_l, P_r


@pytest.mark.parametrize("size,compat", [((120, 392), 1), ((360, 1176), 1), ((192, 640), 0), ((376, 1240), 1)])
def test_preprocess_is_bit_exact(pair, size, compat):
    frames, P_l, _ = pair
    c = cpu_backend.CpuBackend(net_height=size[0], net_width=size[1], bug_compat_p=compat)
    got, P = c.preprocess(frames[0][0], P_l)
    ref, Pr = fe.preprocess(frames[0][0], P_l, size[0], size[1], bool(compat))
    assert np.array_equal(got, ref)
    assert np.array_equal(P.view(np.uint64), Pr.view(np.uint64))                  # bit patterns, incl. the bug-compatible denormal
    c.close()


def test_network_agrees_with_the_torch_restatement(cpu, vgg_plan, pair):
    frames, P_l, _ = pair
    x = fe.to_network_input(fe.preprocess(frames[0][0], P_l, H, W)[0])[None, None]
    det, desc = cpu.forward(x)
    rdet, rdesc = net.forward(vgg_plan, x)
    assert np.abs(det - rdet).max() <= 1e-4 * max(1.0, np.abs(rdet).max())
    assert np.abs(desc - rdesc).max() <= 1e-4


@pytest.mark.parametrize("name", ["sp_squeeze", "sp_mbv1", "sp_mbv2"])
def test_reference_graphs_agree_with_the_torch_restatement(golden_dir, name, pair):
    """Concat (squeeze), depthwise + BatchNorm-after-ReLU (mbv1), residual Add (mbv2) through the C++ executor"""
    path = os.path.join(golden_dir, name + ".spvw")
    plan = weights.load(path)
    c = cpu_backend.CpuBackend(net_height=64, net_width=96)
    c.load_weights(path)
    x = np.random.RandomState(3).rand(1, 1, 64, 96).astype(np.float32)
    det, desc = c.forward(x)
    rdet, rdesc = net.forward(plan, x)
    assert np.abs(det - rdet).max() <= 1e-4 * max(1.0, np.abs(rdet).max())
    assert np.abs(desc - rdesc).max() <= 1e-4
    c.close()


def test_heatmap_nms_sampling(cpu, vgg_plan, pair):
    frames, P_l, _ = pair
    ref = fe.detect(vgg_plan, frames[0][0], P_l, H, W)
    heat = cpu.heatmap(ref["det"])
    assert np.abs(heat - ref["heat"]).max() <= 2e-6
    assert np.array_equal(cpu.nms(ref["heat"]), ref["xy"])                         # same heat map -> bit-exact keypoints
    d = cpu.sample_descriptors(ref["desc"], ref["xy"])
    assert np.abs(d - ref["descriptors"]).max() <= 1e-6
    # ties and a ramp: the pinned total order (confidence desc, column-major index asc)
    rng = np.random.RandomState(0)
    h2 = (rng.randint(0, 6, size=(H, W)) * 0.01).astype(np.float32)
    assert np.array_equal(cpu.nms(h2), fe.nms(h2))


def test_matching_is_bit_exact(cpu, golden_dir):
    m = np.load(os.path.join(golden_dir, "oracle_match.npz"))
    for selector, cross in (("KNN", False), ("NN", False), ("NN", True)):
        idx, d = cpu.match(m["a"], m["b"], selector, cross)
        ridx, rd = matching.bf_match(m["a"], m["b"], selector, cross, 0.8)
        assert np.array_equal(idx, ridx) and np.array_equal(d, rd), (selector, cross)
    rng = np.random.RandomState(2)
    a = rng.randn(37, 256).astype(np.float32)
    b = np.concatenate([a[:10], a[:10], rng.randn(5, 256).astype(np.float32)])   # exact duplicates: ties go to the lowest index
    for selector, cross in (("KNN", False), ("NN", True)):
        idx, d = cpu.match(a, b, selector, cross)
        ridx, rd = matching.bf_match(a, b, selector, cross, 0.8)
        assert np.array_equal(idx, ridx) and np.array_equal(d, rd)
    assert np.array_equal(cpu.match(a, b[:1], "KNN")[0], np.full(37, -1))          # nb < 2: nothing passes the ratio test
    assert len(cpu.match(a[:0], b)[0]) == 0


def test_odometry_stages_agree_with_the_oracle(cpu, golden_dir):
    g = np.load(os.path.join(golden_dir, "oracle_odometry.npz"))
    P_l, P_r = g["P_l"], g["P_r"]
    xyz = cpu.triangulate(P_l, P_r, g["cl"], g["cr"])
    assert np.max(np.abs(xyz - g["pts"]) / np.maximum(np.abs(g["pts"]), 1e-3)) <= 2e-6
    K = P_l[:, :3]
    ok, r, t, inl = cpu.pnp_ransac(K, g["pts"], g["pl"], np.zeros(3), np.zeros(3), seed=0)
    assert ok == bool(g["ok"]) and np.array_equal(inl, g["inliers"])              # inlier set bit-exact
    assert np.abs(r - g["rvec"]).max() <= 1e-8 and np.abs(t - g["tvec"]).max() <= 1e-8
    obs = np.zeros(2 * len(inl), cpu_backend.OBS_DTYPE)
    obs["X"][0::2] = obs["X"][1::2] = g["pts"][inl]
    obs["uv"][0::2], obs["uv"][1::2] = g["pl"][inl], g["pr"][inl]
    obs["cam"][1::2] = 1
    q, t2, s = cpu.pnp_refine(P_l, P_r, obs, od.rvec_to_quat(g["rvec"]), g["tvec"])
    assert s.iterations == int(g["iterations"]) and bool(s.converged) == bool(g["converged"])
    assert np.abs(q - g["q"]).max() <= 1e-9 and np.abs(t2 - g["t"]).max() <= 1e-9
    # all four block kinds (inverse transformation, right camera) against the oracle on the same blocks
    rng = np.random.RandomState(5)
    n = 60
    obs2 = np.zeros(n, cpu_backend.OBS_DTYPE)
    obs2["X"] = g["pts"][inl][:n] + rng.randn(n, 3).astype(np.float32) * 0.01
    obs2["uv"] = np.where(rng.rand(n, 1) < 0.5, g["pl"][inl][:n], g["pr"][inl][:n])
    obs2["cam"], obs2["inverse"] = rng.randint(0, 2, n), rng.randint(0, 2, n)
    q0 = od.rvec_to_quat(g["rvec"])
    qa, ta, sa = cpu.pnp_refine(P_l, P_r, obs2, q0, g["tvec"])
    oo = (obs2["X"].astype(np.float64), obs2["uv"].astype(np.float64), obs2["cam"].astype(np.int64), obs2["inverse"].astype(np.int64))
    qb, tb, sb = od.pnp_refine(P_l, P_r, oo, q0, g["tvec"])
    assert sa.iterations == sb.iterations and bool(sa.converged) == sb.converged and bool(sa.usable) == sb.usable
    assert np.abs(qa - qb).max() <= 1e-9 and np.abs(ta - tb).max() <= 1e-9


def test_state_machine_agrees_with_the_oracle(golden_dir, pair):
    """three stereo frames through spvo_cpu_frontend_step and through oracle/odometry.py's FrontEndState on the trained
    sp_squeeze graph: index maps bit for bit, poses to 1e-6"""
    frames, P_l, P_r = pair
    path = os.path.join(golden_dir, "sp_squeeze.spvw")
    plan = weights.load(path)
    c = cpu_backend.CpuBackend(net_height=H, net_width=W)
    c.load_weights(path)
    c.frontend_reset("KNN", True, 2.0, 0.25, 4)
    st = od.FrontEndState()
    for k, (L, R) in enumerate(frames):
        res = c.frontend_step(L, R, P_l, P_r)
        rl, rr = fe.detect(plan, L, P_l, H, W), fe.detect(plan, R, P_r, H, W)
        # the two network restatements agree to ~1e-6, so feed the oracle the SAME features the C++ side found when the
        # keypoint sets coincide (they do on this fixture); otherwise the comparison below would test the threshold, not the logic
        dl, dr = c.detect(L, P_l), c.detect(R, P_r)
        assert np.array_equal(dl["xy"].astype(np.int32), rl["xy"]) and np.array_equal(dr["xy"].astype(np.int32), rr["xy"])
        od.add_features(st, dl["xy"], dl["descriptors"], dr["xy"], dr["descriptors"], dl["P"], dr["P"])
        od.match_descriptors(st, 0)
        assert np.array_equal(c.frontend_map(0), st.maps[0])
        if k == 0:
            continue
        od.match_descriptors(st, 1)
        assert np.array_equal(c.frontend_map(1), st.maps[1]) and np.array_equal(c.frontend_map(2), st.maps[2])
        q, t, dbg = od.solve_stereo_odometry(st)
        assert res.n_joined == len(dbg["join"]["cl"]) and res.n_inliers == len(dbg["inliers"]) and bool(res.pnp_ok) == dbg["ok"]
        assert np.abs(np.array(res.q[:]) - q).max() <= 1e-6 and np.abs(np.array(res.t[:]) - t).max() <= 1e-6
        assert res.t_total_ms > 0 and res.t_detect_ms > 0
    c.close()


# ---------------------------------------------------------------- BASELINE config 1: the classic (ORB) front end on the CPU
def test_orb_restatement_properties(cpu, pair):
    """ORB with the reference's parameters (classic.cpp:13-24: 2000 features, scale 1.2, 8 levels, edge 31, FAST threshold 20): the
    per-level budget of the ORB formula, the border, FAST scores above the threshold, and orientation-steered descriptors -- the
    image turned by 180 degrees gives (nearly) the same descriptors at the mirrored keypoints."""
    frames, _, _ = pair
    img = frames[0][0]
    o = cpu.orb(img)
    assert len(o["xy"]) == 2000
    f = 1 / 1.2
    want = [round(2000 * (1 - f) / (1 - f ** 8) * f ** lv) for lv in range(8)]
    got = np.bincount(o["octave"], minlength=8)
    assert np.all(np.abs(got[:7] - want[:7]) <= 1) and got.sum() == 2000
    scale = 1.2 ** o["octave"]
    rows, cols = img.shape
    assert np.all(o["xy"][:, 0] >= 31 * scale - 1e-3) and np.all(o["xy"][:, 0] <= cols - 31 * scale + 1.5 * scale)
    assert o["response"].min() > 20                                  # FAST score above the threshold
    for lv in range(8):                                              # best-first retention inside a level
        r = o["response"][o["octave"] == lv]
        assert np.all(np.diff(r) <= 0)
    assert 0.35 < np.unpackbits(o["desc"], axis=1).mean() < 0.65     # binary tests are balanced
    o2 = cpu.orb(np.ascontiguousarray(img[::-1, ::-1]))
    # level-0 keypoints: the mirrored position exists in the turned image and its descriptor is close (same tests, steered by an
    # orientation that turned with the image)
    k0 = np.nonzero(o["octave"] == 0)[0]
    pos2 = {(int(x), int(y)): i for i, (x, y) in enumerate(o2["xy"]) if o2["octave"][i] == 0}
    dists = []
    for i in k0:
        j = pos2.get((cols - 1 - int(o["xy"][i, 0]), rows - 1 - int(o["xy"][i, 1])))
        if j is not None:
            dists.append(int(np.unpackbits(o["desc"][i] ^ o2["desc"][j]).sum()))
    assert len(dists) > 150 and np.median(dists) < 40                # unrelated descriptors are ~128 bits apart


def test_classic_front_end_tracks_the_synthetic_motion(golden_dir):
    """ClassicFeatureFrontEnd(ORB, ORB, BF, KNN) at the native resolution through the CPU state machine (BASELINE config 1: plumbing
    and the CPU baseline's workload): Hamming KNN matches, stereo gate with min_disparity = stereo_threshold (hpp:203-206), and a
    pose close to the synthetic ground truth."""
    frames, poses, P_l, P_r = synth.stereo_sequence(4, os.path.join(golden_dir, "images", "0000000000.png"), seed=0)
    c = cpu_backend.CpuBackend(net_height=360, net_width=1176)
    c.frontend_reset_classic("KNN", True, 2.0, 4)
    for k, (L, R) in enumerate(frames):
        r = c.frontend_step(L, R, P_l, P_r)
        assert r.n_kp_l == 2000 and r.n_stereo > 1000
        m = c.frontend_map(0)
        assert len(m) == 2000 and (m >= 0).sum() == r.n_stereo and m.max() < 2000
        if k == 0:
            continue
        assert r.pnp_ok and r.n_inliers > 300 and r.n_temporal > 700
        Rt, tt = synth.relative_pose(poses[k - 1], poses[k])
        assert np.abs(np.array(r.t[:]) - tt).max() < 0.1             # metres, per 0.8 m step
    c.close()
