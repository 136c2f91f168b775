"""The C++ host mirror of FeatureFrontEnd (host/feature_detection.cpp), driven with the call
sequence of visual_odometry_node.cpp:150-262, against the oracle's restatement of the same state
machine fed with identical upstream features (so that poses must agree to 1e-4)."""
import os
import shutil

import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import odometry as od
from spvo import host, synth, weights

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def models_dir(tmp_path_factory, squeeze_weights_path):
    d = tmp_path_factory.mktemp("models")
    os.makedirs(d / "laptop")
    shutil.copyfile(squeeze_weights_path, d / "laptop" / weights.engine_name("sp_squeeze", 2, 360, 1176, "FP32"))
    return str(d)


@pytest.fixture(scope="module")
def sequence(golden_dir):
    return synth.stereo_sequence(5, os.path.join(golden_dir, "images", "0000000000.png"), seed=0)


def _angle(q):
    q = q / np.linalg.norm(q)
    return 2 * np.arctan2(np.linalg.norm(q[:3]), abs(q[3]))


@pytest.mark.parametrize("selector,cross", [("KNN", True), ("NN", True)])
def test_host_sequence_matches_oracle_state_machine(models_dir, sequence, selector, cross):
    frames, poses, P_l, P_r = sequence
    fe = host.FrontEnd(models_dir, prefix="sp_squeeze", selector=selector, cross_check=cross)
    assert fe.engine_loaded, fe.last_error
    st = od.FrontEndState()
    errs = []
    traj_gpu, traj_cpu = [np.eye(4)], [np.eye(4)]
    for k, (L, R) in enumerate(frames):
        res = fe.step(L, R, P_l, P_r)
        # identical upstream for the oracle: the GPU's own keypoints / descriptors / P
        from oracle import frontend as ofe
        _, Pl2 = ofe.preprocess(L, P_l, 360, 1176, True)
        _, Pr2 = ofe.preprocess(R, P_r, 360, 1176, True)
        od.add_features(st, fe.keypoints(host.CURR_LEFT), fe.descriptors(host.CURR_LEFT),
                        fe.keypoints(host.CURR_RIGHT), fe.descriptors(host.CURR_RIGHT), Pl2, Pr2)
        ocross = cross and selector != "KNN"                       # base.cpp:27-28
        idx0, _ = od.match_descriptors(st, 0, selector, ocross)
        assert np.array_equal(fe.map_of_indices(0), idx0)          # maps_of_indices: bit-exact
        q, tr, d = fe.matches(0)
        assert np.array_equal(q, np.nonzero(idx0 >= 0)[0]) and np.array_equal(tr, idx0[idx0 >= 0])
        if k == 0:
            assert res is None and fe.dq_size() == 2
            continue
        idx1, _ = od.match_descriptors(st, 1, selector, ocross)
        assert np.array_equal(fe.map_of_indices(1), idx1)
        assert np.array_equal(fe.map_of_indices(2), st.maps[2])    # PREV_LEFT_PREV_RIGHT roll (base.cpp:475-481)
        oq, ot, dbg = od.solve_stereo_odometry(st)
        gq, gt = res
        assert np.array_equal(fe.inliers("post"), dbg["join"]["post"])
        assert np.array_equal(fe.inliers("pnp"), dbg["inliers"])   # integer outputs: bit-exact
        assert np.abs(gt - ot).max() <= 1e-4 and _angle(od.quat_mul(gq, np.array([-oq[0], -oq[1], -oq[2], oq[3]]))) <= 1e-4
        assert fe.dq_size() == 4 and fe.frame_count() == k
        # ground truth (reported): relative pose error of this synthetic step
        Rgt, tgt = synth.relative_pose(poses[k - 1], poses[k])
        errs.append(np.linalg.norm(gt - tgt))
        for traj, (q_, t_) in ((traj_gpu, (gq, gt)), (traj_cpu, (oq, ot))):   # integrate world_T_curr = world_T_prev * (curr_T_prev)^-1
            S = np.eye(4)
            S[:3, :3], S[:3, 3] = od.quat_to_rot(np.asarray(q_)), t_
            traj.append(traj[-1] @ np.linalg.inv(S))
    assert max(errs) < 0.05, errs                                   # metres per ~0.8 m step
    # SURVEY.md section 8d parity gate: ATE(GPU trajectory, CPU trajectory) <= 1e-3 m
    ate = np.sqrt(np.mean([np.sum((a[:3, 3] - b[:3, 3]) ** 2) for a, b in zip(traj_gpu, traj_cpu)]))
    assert ate <= 1e-3, ate
    fe.clear()
    assert fe.dq_size() == 0 and fe.frame_count() == 0
    fe.close()


def test_host_error_conventions(models_dir, tmp_path):
    # missing engine file: logged, object left half-initialised, calls return without throwing (nn.cpp:53-55)
    fe = host.FrontEnd(str(tmp_path), prefix="does_not_exist")
    assert not fe.engine_loaded and "no such engine file" in fe.last_error
    img = np.zeros((376, 1241), np.uint8)
    P_l, P_r = synth.projection_matrices()
    fe.add_stereo_image_pair(img, img, P_l, P_r)
    assert fe.dq_size() == 0
    fe.close()
    # wrong batch size (nn.cpp:489-491)
    fe = host.FrontEnd(models_dir, prefix="sp_squeeze", batch=3)
    assert not fe.engine_loaded and "Wrong batch size" in fe.last_error
    fe.close()
    # blank images: no keypoints, no matches, no crash; pose falls back to the prior (base.cpp:244-250)
    fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
    assert fe.step(img, img, P_l, P_r) is None
    q, t = fe.step(img, img, P_l, P_r)
    assert np.allclose(q, [0, 0, 0, 1]) and np.allclose(t, 0)
    fe.close()


def test_prefetch_pipeline_is_transparent(models_dir, sequence):
    """Handing the next one or two pairs over early (their detector overlapped with this frame's matching
    and solving, the tail of one submission overlapped with the network of the next) changes nothing."""
    import torch
    frames, poses, P_l, P_r = sequence
    dev = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in frames]
    rows, cols = frames[0][0].shape
    out = {}
    for depth in (0, 1, 2, 3, 4):     # 4: the front end switches trunk pairing on (two pairs per set of network launches)
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        res = []
        for k, (dl, dr) in enumerate(dev):
            ahead = [(dev[k + d][0].data_ptr(), dev[k + d][1].data_ptr()) if d <= depth and k + d < len(dev) else None for d in (1, 2, 3, 4)]
            r = fe.step_device(dl.data_ptr(), dr.data_ptr(), rows, cols, dl.stride(0), P_l, P_r, ahead[0], ahead[1], next3_pair=ahead[2], next4_pair=ahead[3])
            res.append((r, fe.keypoints(host.CURR_LEFT), fe.map_of_indices(0), fe.map_of_indices(1) if k else None, fe.inliers("pnp")))
        out[depth] = res
        fe.close()
    for depth in (1, 2, 3, 4):
        for a, b in zip(out[0], out[depth]):
            assert np.array_equal(a[1], b[1]) and np.array_equal(a[2], b[2]) and np.array_equal(a[4], b[4])
            if a[3] is not None:
                assert np.array_equal(a[3], b[3])
            if a[0] is not None:
                assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1])      # poses: bit-identical


@pytest.mark.parametrize("arrangement", ["default", "deferred_copies", "heads_on_tail", "heads_on_net"])
def test_the_reference_entry_point_with_host_images_equals_the_device_entry(models_dir, sequence, arrangement, tuning):
    """addStereoImagePair(cv::Mat&, ...) -- the reference's own interface (node.cpp:175): host images in, resized images and
    descriptors back in images_dq / descriptors_dq -- without look-ahead, with one and with two pairs announced through
    prefetchStereoImagePair, and the device-resident entry: the same keypoints, descriptors, index maps, inliers and poses,
    bit for bit.  Arrangements: the bulk copies into the deques made inside addStereoImagePair (default, as the reference) or
    deferred behind the solve (setDeferredHostCopies(true)); the heads on the tail / on the network stream (diagnostic switch)."""
    import torch
    frames, poses, P_l, P_r = sequence
    rows, cols = frames[0][0].shape
    out = {}
    if arrangement.startswith("heads_on"):
        tuning(heads_on_net=1 if arrangement == "heads_on_net" else 0)
    for mode in ("device", 0, 1, 2, 4):
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        if arrangement == "deferred_copies":
            fe.set_deferred_copies(True)
        res = []
        if mode == "device":
            dev = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in frames]
            for k, (dl, dr) in enumerate(dev):
                fe.add_stereo_image_pair_device(dl.data_ptr(), dr.data_ptr(), rows, cols, dl.stride(0), P_l, P_r, host_descriptors=True)
                fe.match_descriptors(host.CURR_LEFT_CURR_RIGHT)
                r = None
                if k:
                    fe.match_descriptors(host.CURR_LEFT_PREV_LEFT)
                    r = fe.solve_stereo_odometry()
                res.append((r, fe.keypoints(host.CURR_LEFT), fe.descriptors(host.CURR_RIGHT), fe.map_of_indices(0), fe.map_of_indices(1) if k else None,
                            fe.inliers("pnp"), None))
        else:
            mats = [(fe.make_image(L), fe.make_image(R)) for L, R in frames]
            for k in range(len(frames)):
                ahead = [mats[k + d] if d <= mode and k + d < len(mats) else None for d in (1, 2, 3, 4)]
                r = fe.step_host(mats[k][0], mats[k][1], P_l, P_r, ahead[0], ahead[1], next3_pair=ahead[2], next4_pair=ahead[3])
                res.append((r, fe.keypoints(host.CURR_LEFT), fe.descriptors(host.CURR_RIGHT), fe.map_of_indices(0), fe.map_of_indices(1) if k else None,
                            fe.inliers("pnp"), fe.image(host.CURR_LEFT)))
            for m in mats:
                fe.free_image(m[0]); fe.free_image(m[1])
        out[mode] = res
        fe.close()
    from oracle import frontend as ofe
    for mode in (0, 1, 2, 4):
        for k, (a, b) in enumerate(zip(out["device"], out[mode])):
            for i in (1, 2, 3, 5):
                assert np.array_equal(a[i], b[i]), (mode, k, i)
            if a[4] is not None:
                assert np.array_equal(a[4], b[4])
            if a[0] is not None:
                assert np.array_equal(a[0][0], b[0][0]) and np.array_equal(a[0][1], b[0][1])      # poses: bit-identical
            assert np.array_equal(b[6], ofe.preprocess(frames[k][0], P_l, 360, 1176)[0])           # images_dq holds the resized u8 image (nn.cpp:154)


def test_public_deques_are_filled_when_add_stereo_image_pair_returns(models_dir, sequence):
    """nn.cpp:154, 494-498: the reference fills images_dq and descriptors_dq INSIDE addStereoImagePair.  Read the public members
    immediately after the call -- no completeHostCopies(), no match, no solve in between -- on the default configuration: the resized
    u8 images are the oracle's, bit for bit, and the descriptors are those the device slots hold (the ones matchDescriptors reads);
    both also one call later, when the pair has rolled to the PREV positions.  With the opt-in deferral the same reads agree once
    completeHostCopies() has run (the accessor that calls it first)."""
    from oracle import frontend as ofe
    frames, poses, P_l, P_r = sequence
    fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
    mats = [(fe.make_image(L), fe.make_image(R)) for L, R in frames[:3]]
    prev = None
    for k in range(3):
        import ctypes as C
        Pl, Pr = np.ascontiguousarray(P_l, np.float64), np.ascontiguousarray(P_r, np.float64)
        fe.lib.spvo_host_add_stereo_pair_mat(fe.h, C.c_void_p(mats[k][0]), C.c_void_p(mats[k][1]), host._p(Pl), host._p(Pr))
        raw = [(fe.image_raw(pos), fe.descriptors_raw(pos)) for pos in (host.CURR_LEFT, host.CURR_RIGHT)]          # straight from the deques
        for (img, desc), src, pos in zip(raw, frames[k], (host.CURR_LEFT, host.CURR_RIGHT)):
            assert np.array_equal(img, ofe.preprocess(src, P_l, 360, 1176)[0])
            n = len(fe.keypoints(pos))
            assert desc.shape == (n, 256) and n > 100
            assert np.abs(np.linalg.norm(desc, axis=1) - 1.0).max() < 1e-5                                         # normalised rows: not uninitialised memory
            assert np.array_equal(desc, fe.descriptors(pos))                                                       # = after completeHostCopies()
        if prev is not None:                                                                                      # rolled to PREV_LEFT / PREV_RIGHT
            for (img, desc), pos in zip(prev, (host.PREV_LEFT, host.PREV_RIGHT)):
                assert np.array_equal(img, fe.image_raw(pos)) and np.array_equal(desc, fe.descriptors_raw(pos))
        prev = raw
        fe.match_descriptors(host.CURR_LEFT_CURR_RIGHT)
        if k:
            fe.match_descriptors(host.CURR_LEFT_PREV_LEFT)
            fe.solve_stereo_odometry()
    for m in mats:
        fe.free_image(m[0]); fe.free_image(m[1])
    fe.close()


def test_fp16_engine_through_the_host_class(tmp_path, squeeze_weights_path, sequence):
    """TensorRtPrecision::FP16 (hpp:124-126): the front end loads `<prefix>_<B>_<H>_<W>_FP16.spvw`, refuses a file
    of the other precision, and tracks the synthetic ego-motion like the FP32 engine (the trained squeeze weights;
    < 5 cm per 0.8 m step), with most keypoints in common."""
    frames, poses, P_l, P_r = sequence
    d = tmp_path / "models"
    os.makedirs(d / "laptop")
    plan = weights.load(squeeze_weights_path)
    weights.save(plan, str(d / "laptop" / weights.engine_name("sp_squeeze", 2, 360, 1176, "FP16")), precision="FP16")
    weights.save(plan, str(d / "laptop" / weights.engine_name("sp_squeeze", 2, 360, 1176, "FP32")))
    weights.save(plan, str(d / "laptop" / weights.engine_name("mislabelled", 2, 360, 1176, "FP16")))     # an FP32 engine under an FP16 name
    bad = host.FrontEnd(str(d), prefix="mislabelled", precision="FP16")
    assert not bad.engine_loaded and "was not built for FP16" in bad.last_error
    bad.close()
    out = {}
    for prec in ("FP32", "FP16"):
        fe = host.FrontEnd(str(d), prefix="sp_squeeze", precision=prec)
        assert fe.engine_loaded, fe.last_error
        res = []
        for L, R in frames:
            r = fe.step(L, R, P_l, P_r)
            res.append((r, fe.keypoints(host.CURR_LEFT)))
        out[prec] = res
        fe.close()
    for k in range(1, len(frames)):
        (q16, t16), kp16 = out["FP16"][k]
        (q32, t32), kp32 = out["FP32"][k]
        q_true, t_true = synth.relative_pose(poses[k - 1], poses[k])
        assert np.linalg.norm(t16 - t_true) < 0.05 and np.linalg.norm(t16 - t32) < 0.05
        a, b = set(map(tuple, kp16.tolist())), set(map(tuple, kp32.tolist()))
        assert len(a & b) / len(a | b) > 0.8


def test_int8_engine_through_the_host_class(tmp_path, sequence):
    """BASELINE config 5 end to end: an INT8 sp_mbv1 engine, calibrated on the device (spvo/quant.py) with the first
    stereo pair, loaded by the host class under the reference's engine naming rule, tracks the synthetic ego-motion;
    and the device calibration agrees with the oracle's."""
    from oracle import frontend as ofe, net_int8
    from spvo import quant
    from tests.conftest import GOLDEN
    frames, poses, P_l, P_r = sequence
    plan = weights.load(os.path.join(GOLDEN, "sp_mbv1.spvw"))
    x = np.stack([ofe.to_network_input(ofe.preprocess(img, P_l, 360, 1176, True)[0]) for img in frames[0]])[:, None]
    scales = quant.calibrate(plan, [x], 360, 1176)
    ref = net_int8.calibrate(plan, [x])
    assert np.allclose(scales, ref, rtol=2e-3)                       # percentiles of activations that agree to 1e-4
    plan.act_scales = scales
    d = tmp_path / "models"
    os.makedirs(d / "laptop")
    weights.save(plan, str(d / "laptop" / weights.engine_name("sp_mbv1", 2, 360, 1176, "INT8")), precision="INT8")
    weights.save(plan, str(d / "laptop" / weights.engine_name("sp_mbv1", 2, 360, 1176, "FP32")), precision="FP32")
    out = {}
    for prec in ("FP32", "INT8"):
        fe = host.FrontEnd(str(d), prefix="sp_mbv1", precision=prec)
        assert fe.engine_loaded, fe.last_error
        res = []
        for L, R in frames:
            r = fe.step(L, R, P_l, P_r)
            res.append((r, fe.keypoints(host.CURR_LEFT)))
        out[prec] = res
        fe.close()
    for k in range(1, len(frames)):
        (q8, t8), kp8 = out["INT8"][k]
        (q32, t32), kp32 = out["FP32"][k]
        _, t_true = synth.relative_pose(poses[k - 1], poses[k])
        assert np.linalg.norm(t32 - t_true) < 0.05
        assert np.linalg.norm(t8 - t_true) < 0.10                    # naive post-training quantisation: still tracks
        a, b = set(map(tuple, kp8.tolist())), set(map(tuple, kp32.tolist()))
        assert len(a & b) / len(a | b) > 0.4


def test_result_changing_options_have_setters(models_dir, sequence, monkeypatch):
    """SuperPointFeatureFrontEnd::setMaxKeypoints / ::setMatchFp8 and FeatureFrontEnd::setDevice (through harness_capi's
    spvo_host_set_options): the options of the front ends constructed afterwards.  A setter outranks the environment variable of the same
    name, a negative value hands the decision back to it."""
    frames, _, P_l, P_r = sequence
    L, R = frames[0]
    try:
        host.set_options(max_keypoints=300)
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        assert fe.engine_loaded, fe.last_error
        fe.add_stereo_image_pair(L, R, P_l, P_r)
        assert len(fe.keypoints(host.CURR_LEFT)) == 300 and len(fe.keypoints(host.CURR_RIGHT)) == 300      # the cap bites (the frames give ~1000)
        fe.close()
        monkeypatch.setenv("SPVO_MAX_KEYPOINTS", "200")
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        fe.add_stereo_image_pair(L, R, P_l, P_r)
        assert len(fe.keypoints(host.CURR_LEFT)) == 300                      # the setter outranks the environment
        fe.close()
        host.set_options(max_keypoints=-1)
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        fe.add_stereo_image_pair(L, R, P_l, P_r)
        assert len(fe.keypoints(host.CURR_LEFT)) == 200                      # ... and hands the decision back to it
        fe.close()
        monkeypatch.delenv("SPVO_MAX_KEYPOINTS")
        # fp8 shortlist: the GEMM only prunes, the re-rank is exact -- the matches are the fp32 matcher's, bit for bit
        got = {}
        for fp8 in (0, 1):
            host.set_options(match_fp8=fp8)
            fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
            fe.add_stereo_image_pair(L, R, P_l, P_r)
            fe.match_descriptors(host.CURR_LEFT_CURR_RIGHT)
            got[fp8] = fe.matches(host.CURR_LEFT_CURR_RIGHT)
            assert fe.context().match_fp8() == bool(fp8)
            fe.close()
        for x, y in zip(got[0], got[1]):
            assert np.array_equal(x, y)
        # device: an index that does not exist is refused when the context is created (no silent fall-back to device 0)
        host.set_options(device=63)
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        assert not fe.engine_loaded and "device 63" in fe.last_error
        fe.close()
        monkeypatch.setenv("SPVO_DEVICE", "62")
        host.set_options(device=0)
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        assert fe.engine_loaded, fe.last_error                              # setDevice(0) outranks SPVO_DEVICE=62
        fe.close()
    finally:
        host.set_options()


@pytest.mark.parametrize("knn,cross", [(1, 0), (0, 0), (0, 1)])
def test_classic_front_end_matches_binary_descriptors_on_the_gpu(knn, cross):
    """matchDescriptors of a ClassicFeatureFrontEnd(ORB, ORB, BF, ...) -- base.cpp:434-500 with the NORM_HAMMING matcher of
    base.cpp:17-21 -- runs spvo_match_hamming on the host descriptor matrices of descriptors_dq: the index maps equal the
    oracle's for both match types of a frame (the detectors themselves need OpenCV and are not part of this test)."""
    import ctypes
    from oracle import matching
    from spvo import host
    lib = host.load()
    lib.spvo_host_classic_match.restype = ctypes.c_int
    lib.spvo_host_classic_match.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_int), ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_int]
    rng = np.random.RandomState(3)
    sets = [rng.randint(0, 256, (n, 32)).astype(np.uint8) for n in (1500, 1400, 1600, 1550)]   # prevL, prevR, currL, currR
    sets[3][:400] = sets[2][100:500]                                                             # stereo partners
    sets[0][200:700] = sets[2][300:800]                                                          # temporal partners
    sets[0][200:300, 1] ^= 3
    ptrs = (ctypes.c_void_p * 4)(*[s.ctypes.data for s in sets])
    ns = (ctypes.c_int * 4)(*[len(s) for s in sets])
    for match_type, (qa, tb) in ((0, (2, 3)), (1, (2, 0))):                                      # CURR_LEFT_CURR_RIGHT, CURR_LEFT_PREV_LEFT
        out = np.full(len(sets[qa]), -7, np.int32)
        n = lib.spvo_host_classic_match(knn, cross, match_type, ptrs, ns, 32, out.ctypes.data, len(out))
        ref, _ = matching.bf_match_hamming(sets[qa], sets[tb], "KNN" if knn else "NN", bool(cross))
        assert n == len(ref) and np.array_equal(out, ref)
        assert (ref >= 0).sum() > 300


def test_deferred_solve_gives_the_same_poses_one_step_later(models_dir, sequence):
    """solveStereoOdometrySubmit / Collect (the solve handed over, its pose collected during the next step) against the one-piece
    solveStereoOdometry on the same sequence: identical poses and state, delivered one call later."""
    import torch
    frames, _, P_l, P_r = sequence
    outs = {}
    for deferred in (False, True):
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze", selector="KNN", cross_check=True)
        assert fe.engine_loaded, fe.last_error
        dev = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in frames]
        torch.cuda.synchronize()
        poses = []
        for k, (L, R) in enumerate(dev):
            nxt = (dev[k + 1][0].data_ptr(), dev[k + 1][1].data_ptr()) if k + 1 < len(dev) else None
            r = fe.step_device(L.data_ptr(), R.data_ptr(), L.shape[0], L.shape[1], L.stride(0), P_l, P_r, nxt, None, deferred_solve=deferred)
            if r is not None:
                poses.append((r[0].copy(), r[1].copy()))
        last = fe.finish_solve()
        if deferred:
            assert last is not None
            poses.append((last[0].copy(), last[1].copy()))
        else:
            assert last is None
        outs[deferred] = (poses, fe.frame_count())
        fe.close()
    n = len(frames) - 1
    assert len(outs[False][0]) == len(outs[True][0]) == n and outs[False][1] == outs[True][1] == n
    for (qa, ta), (qb, tb) in zip(outs[False][0], outs[True][0]):
        assert np.array_equal(qa, qb) and np.array_equal(ta, tb)


def test_the_block_loop_equals_the_call_sequence(models_dir, sequence):
    """spvo_host_run_device_block (host/harness_capi.cpp: the C loop bench.py times -- n stereoCallbacks per call, pairs announced `depth`
    ahead, the solve deferred by a frame) against the plain call sequence of the node on the same frames: every pose identical bit for
    bit, one record per frame, whatever the depth and however the sequence is cut into blocks (the pairs a block announces last are
    collected by the next block); the records carry a latency and the solver's outcome."""
    import torch
    frames, _, P_l, P_r = sequence
    dev = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in frames]
    torch.cuda.synchronize()
    rows, cols = frames[0][0].shape
    stride = dev[0][0].stride(0)
    n = len(dev)
    fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
    assert fe.engine_loaded, fe.last_error
    ref = []
    for k, (L, R) in enumerate(dev):
        r = fe.step_device(L.data_ptr(), R.data_ptr(), rows, cols, stride, P_l, P_r)
        ref.append(None if r is None else (r[0].copy(), r[1].copy()))
    fe.close()
    dl, dr = [d[0].data_ptr() for d in dev], [d[1].data_ptr() for d in dev]
    for depth, cuts in ((4, (n,)), (4, (3, n - 3)), (2, (1, 2, n - 3)), (0, (n,))):
        fe = host.FrontEnd(models_dir, prefix="sp_squeeze")
        recs, first = [], 0
        for m in cuts:
            # (the last block of a run must not announce pairs beyond the sequence: the cycle wraps, as in bench.py, so it simply would)
            recs.append(fe.run_device_block(dl, dr, rows, cols, stride, P_l, P_r, first, m, depth=depth, deferred=depth > 0))
            first += m
        rec = np.concatenate(recs)
        fe.close()
        assert len(rec) == n and not rec["has_pose"][0] and rec["has_pose"][1:].all()
        for k in range(1, n):
            assert np.array_equal(rec["q"][k], ref[k][0]) and np.array_equal(rec["t"][k], ref[k][1]), (depth, cuts, k)
        assert (rec["latency_ms"] > 0).all() and (rec["keypoints_left"] > 100).all() and (rec["pnp_inliers"][1:] > 10).all()
        assert rec["accepted"][1:].mean() > 0.5 and (rec["lm_iterations"][rec["refined"] == 1] > 0).all()


def test_classic_front_end_on_the_gpu_equals_the_cpu_state_machine(sequence):
    """ClassicFeatureFrontEnd(ORB, ORB, BF, KNN) -- BASELINE config 1's front end, node.cpp:353-360 -- through the mirror class:
    ORB on the GPU (spvo_orb_detect), Hamming matching on the GPU (spvo_match_hamming), the solver through the C ABI, at the
    native resolution.  Against the CPU restatement's state machine on the same frames: keypoint counts, stereo matches and PnP
    inliers identical (bit-exact features and matches upstream), poses within 1e-6; and the poses track the synthetic motion."""
    from oracle import cpu_backend
    frames, gt, P_l, P_r = sequence
    frames = frames[:4]
    poses, stats, _ = host.classic_sequence(frames, P_l, P_r, "KNN", True, 2.0, 4)
    c = cpu_backend.CpuBackend(net_height=360, net_width=1176)
    c.frontend_reset_classic("KNN", True, 2.0, 4)
    for k, (L, R) in enumerate(frames):
        r = c.frontend_step(L, R, P_l, P_r)
        assert stats[k, 0] == r.n_kp_l == 2000 and stats[k, 1] == r.n_kp_r and stats[k, 2] == r.n_stereo
        if k == 0:
            continue
        assert stats[k, 3] == r.n_inliers and r.n_inliers > 300
        Rc, Rg = od.quat_to_rot(np.array(r.q[:])), od.quat_to_rot(poses[k, :4])       # both: cam0_curr_T_cam0_prev
        assert np.abs(Rg - Rc).max() <= 1e-6 and np.abs(poses[k, 4:] - np.array(r.t[:])).max() <= 1e-6
        Rt, tt = synth.relative_pose(gt[k - 1], gt[k])
        assert np.abs(np.array(r.t[:]) - tt).max() < 0.1
    c.close()


def test_a_node_like_caller_linked_against_the_host_library_runs(tmp_path, squeeze_weights_path, sequence, monkeypatch):
    """tests/boundary_node_caller.cpp -- construction from launch parameters, stereoCallback, the goal callback's reset, as
    visual_odometry_node.cpp:150-262, 316, 330-403 write them -- is compiled, LINKED against libspvo_host.so and EXECUTED on
    the GPU, with both constructors (is_classic false / true).  The poses solveStereoOdometry(tf2::Transform&) hands it equal,
    bit for bit, those of the same frames through spvo.host (FrontEnd.step / classic_sequence): the caller's path through the
    class interface and the harness's C wrappers are the same code underneath, and this is the run that proves the first one."""
    import ctypes as C
    import subprocess
    from spvo import capi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.dirname(capi.LIB_PATH)
    so = str(tmp_path / "libboundary_node_caller.so")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-Wall", "-Werror", "-I" + os.path.join(pkg, "host"),
                        os.path.join(root, "tests", "boundary_node_caller.cpp"), "-o", so, "-L" + pkg, "-lspvo_host", "-lspvo", "-Wl,-rpath," + pkg],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    models = tmp_path / "models"
    os.makedirs(models / "laptop")
    shutil.copyfile(squeeze_weights_path, models / "laptop" / weights.engine_name("superpoint_pretrained", 2, 360, 1176, "FP32"))
    monkeypatch.setenv("SPVO_MODELS_DIR", str(models))
    lib = C.CDLL(so)
    lib.boundary_node_run.restype = C.c_int
    lib.boundary_node_run.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_double, C.c_int, C.c_int,
                                      C.c_void_p, C.c_void_p]
    frames, _, P_l, P_r = sequence
    frames = frames[:3]
    Ls = np.ascontiguousarray(np.stack([f[0] for f in frames]), np.uint8)
    Rs = np.ascontiguousarray(np.stack([f[1] for f in frames]), np.uint8)
    rows, cols = Ls.shape[1:]
    Pl = np.ascontiguousarray(P_l, np.float64).reshape(12)
    Pr = np.ascontiguousarray(P_r, np.float64).reshape(12)

    def run(is_classic, min_disparity, h, w):
        poses = np.zeros((len(frames), 7), np.float64)
        check = C.c_double(0)
        n = lib.boundary_node_run(int(is_classic), len(frames), Ls.ctypes.data, Rs.ctypes.data, rows, cols, Pl.ctypes.data, Pr.ctypes.data, min_disparity, h, w,
                                  poses.ctypes.data, C.byref(check))
        assert n == len(frames) - 1 and np.isnan(poses[0]).all() and check.value > 0
        return poses

    # ---- SuperPointFeatureFrontEnd (node.cpp:396-403).  (The harness's constructor call also points the class at this test's
    # models directory -- SuperPointFeatureFrontEnd::setModelsDir outranks SPVO_MODELS_DIR and earlier tests have set it.)
    fe = host.FrontEnd(str(models), prefix="superpoint_pretrained", selector="KNN", cross_check=True, min_disparity=0.25)
    assert fe.engine_loaded, fe.last_error
    ref = [fe.step(L, R, P_l, P_r) for L, R in frames]
    fe.close()
    got = run(False, 0.25, 360, 1176)
    for k in range(1, len(frames)):
        assert np.array_equal(got[k, :4], ref[k][0]) and np.array_equal(got[k, 4:], ref[k][1]), k
        assert np.linalg.norm(ref[k][1]) > 0.3                                   # a real step of the synthetic motion, not the prior
    # ---- ClassicFeatureFrontEnd (node.cpp:353-360), native resolution
    got_c = run(True, 2.0, 0, 0)
    ref_c, stats, _ = host.classic_sequence(frames, P_l, P_r, "KNN", True, 2.0, 4)
    assert np.array_equal(got_c[1:], ref_c[1:]) and (stats[1:, 3] > 300).all()


@pytest.mark.parametrize("precision", ["FP16", "INT8"])
def test_launch_segments_replayed_from_graphs_do_not_change_results(tmp_path, squeeze_weights_path, sequence, precision, tuning):
    """Round 6: for FP16 / INT8 engines the runs of kernel launches between two event operations of a submission -- a group's trunk, its heads,
    a pair's heat map + NMS + sampling, its two matches -- are recorded and, from the third time the same run (same buffer set, slots, batch,
    engine and tuning generation) comes by, replayed from a captured HIP graph: one hipGraphLaunch instead of up to eleven launches
    (csrc/launch_segments.hip.h; tuning "graphs" = 0 keeps plain launches).  Over 48 frames of the block loop with four pairs announced
    ahead and trunk pairing: every pose, every solver outcome and the keypoint / match counts of every frame are identical bit for bit with
    and without, and most segments do go out as replays."""
    import torch
    from spvo import quant
    from oracle import frontend as ofe
    frames, poses, P_l, P_r = sequence
    d = tmp_path / "models"
    os.makedirs(d / "laptop")
    plan = weights.load(squeeze_weights_path) if precision == "FP16" else weights.load(os.path.join(os.path.dirname(squeeze_weights_path), "sp_mbv1.spvw"))
    prefix = "sp_squeeze" if precision == "FP16" else "sp_mbv1"
    if precision == "INT8":
        x = np.stack([ofe.to_network_input(ofe.preprocess(img, P_l, 360, 1176, True)[0]) for img in frames[0]])[:, None]
        plan.act_scales = quant.calibrate(plan, [x], 360, 1176)
    weights.save(plan, str(d / "laptop" / weights.engine_name(prefix, 2, 360, 1176, precision)), precision=precision)
    order = [0, 1, 2, 3, 4, 3, 2, 1]
    dev = [(torch.from_numpy(frames[f][0]).cuda().clone(), torch.from_numpy(frames[f][1]).cuda().clone()) for f in order]
    dl, dr = [a.data_ptr() for a, _ in dev], [b.data_ptr() for _, b in dev]
    rows, cols, stride = frames[0][0].shape[0], frames[0][0].shape[1], dev[0][0].stride(0)
    out = {}
    for graphs in (0, 1):
        tuning(graphs=graphs)
        fe = host.FrontEnd(str(d), prefix=prefix, precision=precision)
        assert fe.engine_loaded, fe.last_error
        recs = [fe.run_device_block(dl, dr, rows, cols, stride, P_l, P_r, first, 16, depth=4, deferred=True) for first in (0, 16, 32)]
        prof = fe.context().profile()
        out[graphs] = (np.concatenate(recs), prof.get("segment_graph_launch", {}).get("calls", 0), prof.get("segment_plain_launch", {}).get("calls", 0))
        fe.close()
    a, b = out[0][0], out[1][0]
    for field in ("q", "t", "has_pose", "pnp_ok", "accepted", "refined", "lm_iterations", "pnp_inliers", "stereo_matches", "keypoints_left"):
        assert np.array_equal(a[field], b[field]), field
    assert out[0][1] == 0 and out[0][2] == 0                       # tuning "graphs" = 0: no segment is ever opened
    assert out[1][2] > 0 and out[1][1] >= 0.4 * (out[1][1] + out[1][2]), out[1][1:]   # 48 frames = six rounds of the eight buffer sets: the first two of every key go out as plain launches (and
    #                                                                                       the first frames' allocations start new key generations), the rest as replays (a long loop: 82 %)
