"""End-to-end GPU checks: spvo_detect / spvo_match_slots vs the oracle on the same stereo pair,
and full-size properties that do not depend on the oracle finishing quickly."""
import os

import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import frontend as fe, matching
from spvo import synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def stereo_pair(golden_dir):
    import os
    frames, poses, P_l, P_r = synth.stereo_sequence(2, os.path.join(golden_dir, "images", "0000000000.png"), seed=0)
    return frames, poses, P_l, P_r


def test_detect_matches_oracle_end_to_end(ctx_squeeze, squeeze_plan, stereo_pair):
    frames, _, P_l, P_r = stereo_pair
    L, R = frames[0]
    out = ctx_squeeze.detect(L, R, P_l, P_r, 2, 3, want_resized=True)
    rl = fe.detect(squeeze_plan, L, P_l, 360, 1176)
    rr = fe.detect(squeeze_plan, R, P_r, 360, 1176)
    assert np.array_equal(out["resized_l"], rl["resized"]) and np.array_equal(out["resized_r"], rr["resized"])
    assert np.array_equal(out["P_l"], rl["P"]) and np.array_equal(out["P_r"], rr["P"])
    for side, ref in (("l", rl), ("r", rr)):
        xy = out["xy_" + side].astype(np.int32)
        # keypoints are threshold/sort outcomes of floats that agree to ~1e-6: report set agreement
        a, b = set(map(tuple, xy.tolist())), set(map(tuple, ref["xy"].tolist()))
        assert len(a & b) / len(a | b) > 0.97, len(a & b)
        assert len(xy) == len(ref["xy"]) == 1000
        # descriptors of the common keypoints agree to 1e-4 (north_star tolerance)
        pos = {tuple(p): i for i, p in enumerate(ref["xy"].tolist())}
        common = [(i, pos[tuple(p)]) for i, p in enumerate(xy.tolist()) if tuple(p) in pos]
        gi, ri = map(np.array, zip(*common))
        assert np.abs(out["desc_" + side][gi] - ref["descriptors"][ri]).max() <= 1e-4
    # device-resident slots give the same matches as host descriptors through the oracle matcher
    idx, d = ctx_squeeze.match_slots(2, 3, len(out["xy_l"]))
    ridx, rd = matching.bf_match(out["desc_l"], out["desc_r"], "KNN", False, 0.8)
    assert np.array_equal(idx, ridx) and np.array_equal(d, rd)
    assert (idx >= 0).sum() > 500


@pytest.mark.parametrize("graph", ["vgg", "squeeze"])
def test_detect_at_the_second_engine_size_matches_oracle(graph, vgg_weights_path, vgg_plan, squeeze_weights_path, squeeze_plan, stereo_pair):
    """240 x 784 -- the other size engine_generation.py:20 builds engines for (feature_detection.hpp:296 picks the crop / scale by
    it) -- through the whole GPU path: preprocess (bit-exact image and P), network (another mix of kernels at 30 x 98 cells:
    tests/test_gpu_network.py pins it), NMS, descriptor sampling, matching."""
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    L, R = frames[0]
    path, plan = (vgg_weights_path, vgg_plan) if graph == "vgg" else (squeeze_weights_path, squeeze_plan)
    ctx = capi.Context(net_height=240, net_width=784)
    ctx.load_weights(path)
    out = ctx.detect(L, R, P_l, P_r, 2, 3, want_resized=True)
    for side, img, P in (("l", L, P_l), ("r", R, P_r)):
        ref = fe.detect(plan, img, P, 240, 784)
        assert np.array_equal(out["resized_" + side], ref["resized"]) and np.array_equal(out["P_" + side], ref["P"])
        xy = out["xy_" + side].astype(np.int32)
        a, b = set(map(tuple, xy.tolist())), set(map(tuple, ref["xy"].tolist()))
        assert len(a & b) / len(a | b) > 0.97, (len(a & b), len(a), len(b))
        assert len(xy) == len(ref["xy"])
        pos = {tuple(p): i for i, p in enumerate(ref["xy"].tolist())}
        common = [(i, pos[tuple(p)]) for i, p in enumerate(xy.tolist()) if tuple(p) in pos]
        gi, ri = map(np.array, zip(*common))
        assert np.abs(out["desc_" + side][gi] - ref["descriptors"][ri]).max() <= 1e-4
    idx, d = ctx.match_slots(2, 3, len(out["xy_l"]))
    ridx, rd = matching.bf_match(out["desc_l"], out["desc_r"], "KNN", False, 0.8)
    assert np.array_equal(idx, ridx) and np.array_equal(d, rd)
    ctx.close()


def test_detect_properties_full_size(ctx_vgg, stereo_pair):
    frames, _, P_l, P_r = stereo_pair
    L, R = frames[1]
    a = ctx_vgg.detect(L, R, P_l, P_r, 0, 1)
    b = ctx_vgg.detect(L, R, P_l, P_r, 2, 3)
    for k in ("xy_l", "xy_r", "desc_l", "desc_r"):
        assert np.array_equal(a[k], b[k])                                    # idempotent, slot-independent
    xy = a["xy_l"]
    assert len(xy) <= 1000 and np.all(xy == np.round(xy))                    # integer coordinates (nn.cpp:243)
    assert xy[:, 0].min() >= 4 and xy[:, 0].max() < 1176 - 4 and xy[:, 1].min() >= 4 and xy[:, 1].max() < 360 - 4
    d = np.abs(xy[:, None, :] - xy[None, :, :]).max(-1)
    np.fill_diagonal(d, 99)
    assert d.min() > 4                                                       # NMS: Chebyshev distance > dist_thresh
    assert np.allclose(np.linalg.norm(a["desc_l"], axis=1), 1, atol=1e-5)
    # swapping left and right swaps the outputs
    c = ctx_vgg.detect(R, L, P_r, P_l, 0, 1)
    assert np.array_equal(c["xy_l"], a["xy_r"]) and np.array_equal(c["desc_r"], a["desc_l"])
    # a slot matched against itself, every selector: the brute-force result, bit for bit, on EVERY row.  The seeded,
    # untrained network produces clusters of (near-)duplicate descriptors (squared distances of 1e-10 and below, far
    # under what the fp32 distance GEMM resolves): the re-rank re-scores whole clusters exactly, so ties go to the
    # lowest index as BFMatcher's scan does (base.cpp:462-473)
    n = len(c["xy_l"])
    for selector, cross in (("NN", True), ("NN", False), ("KNN", False)):
        idx, dist = ctx_vgg.match_slots(0, 0, n, selector, cross)
        ridx, rdist = matching.bf_match(c["desc_l"], c["desc_l"], selector, cross, 0.8)
        assert np.array_equal(idx, ridx) and np.array_equal(dist, rdist), (selector, cross)
    idx, dist = ctx_vgg.match_slots(0, 0, n, "NN", False)
    assert np.all(dist == 0)                                                 # every row finds itself or an exact duplicate


def test_detect_in_fp32_split_mode_matches_oracle(vgg_weights_path, vgg_plan, stereo_pair):
    """the whole detector with the FP32 engine evaluated on the bf16x3 split kernels (spvo_set_fp32_split): same bars as
    the native engine -- keypoint sets agree with the oracle (threshold / sort outcomes of floats that agree to ~1e-6),
    descriptors of the common keypoints to 1e-4"""
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    L, R = frames[0]
    ctx = capi.Context()
    ctx.set_fp32_split(True)
    ctx.load_weights(vgg_weights_path)
    out = ctx.detect(L, R, P_l, P_r, 2, 3)
    for side, img, P in (("l", L, P_l), ("r", R, P_r)):
        ref = fe.detect(vgg_plan, img, P, 360, 1176)
        xy = out["xy_" + side].astype(np.int32)
        a, b = set(map(tuple, xy.tolist())), set(map(tuple, ref["xy"].tolist()))
        assert len(a & b) / len(a | b) > 0.97, len(a & b)
        pos = {tuple(p): i for i, p in enumerate(ref["xy"].tolist())}
        common = [(i, pos[tuple(p)]) for i, p in enumerate(xy.tolist()) if tuple(p) in pos]
        gi, ri = map(np.array, zip(*common))
        assert np.abs(out["desc_" + side][gi] - ref["descriptors"][ri]).max() <= 1e-4
    ctx.close()


@pytest.mark.parametrize("switches", [{"winograd": 0}, {"wino4": 0}, {"heads_split": 0}, {"heads_fused": 0}, {"merge_siblings": 0}, {"wino_narrow": 0}, {"wino_dynamic": 0},
                                      {"winograd": 0, "heads_split": 0, "merge_siblings": 0}])
def test_kernel_selection_switches_do_not_change_the_detector(vgg_weights_path, stereo_pair, switches, tuning):
    """The diagnostic switches (spvo_set_tuning, INTEGRATION.md: direct instead of Winograd 3x3 kernels, F(2x2) only, heads as ordinary
    trunk layers / unfused, head siblings launched separately, static tile assignment) select other kernels / streams for the same
    arithmetic: keypoints are identical up to threshold outcomes of heat-map values that differ in the last bits, descriptors agree
    to 1e-5.  The library reads NO environment variable for any of this: an unknown name is refused."""
    from spvo import capi
    with pytest.raises(capi.SpvoError):
        capi.set_tuning("no_such_switch", 1)
    frames, _, P_l, P_r = stereo_pair
    L, R = frames[1]
    outs = []
    for e in ({}, switches):
        tuning(**e)
        ctx = capi.Context()
        ctx.load_weights(vgg_weights_path)
        outs.append(ctx.detect(L, R, P_l, P_r, 2, 3))
        ctx.close()
    a, b = outs
    for side in ("l", "r"):
        sa, sb = set(map(tuple, a["xy_" + side].astype(int).tolist())), set(map(tuple, b["xy_" + side].astype(int).tolist()))
        assert len(sa & sb) / len(sa | sb) > 0.99
        pos = {tuple(p): i for i, p in enumerate(b["xy_" + side].astype(int).tolist())}
        common = [(i, pos[tuple(p)]) for i, p in enumerate(a["xy_" + side].astype(int).tolist()) if tuple(p) in pos]
        gi, ri = map(np.array, zip(*common))
        assert np.abs(a["desc_" + side][gi] - b["desc_" + side][ri]).max() <= 1e-5


def test_prematch_is_transparent(ctx_squeeze, stereo_pair):
    """spvo_set_prematch only moves the two standard matches into the detector's submission."""
    frames, _, P_l, P_r = stereo_pair
    res = {}
    for on in (False, True):
        ctx_squeeze.set_prematch(on, "KNN", False, 0.8)
        a = ctx_squeeze.detect(frames[0][0], frames[0][1], P_l, P_r, 0, 1)
        b = ctx_squeeze.detect(frames[1][0], frames[1][1], P_l, P_r, 2, 3)
        n = len(b["xy_l"])
        res[on] = (ctx_squeeze.match_slots(2, 3, n), ctx_squeeze.match_slots(2, 0, n),
                   ctx_squeeze.match_slots(2, 3, n, "NN", True))       # different parameters: computed on demand
    ctx_squeeze.set_prematch(False)
    for x, y in zip(res[False], res[True]):
        assert np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])
    assert (res[True][1][0] >= 0).sum() > 300                            # temporal matches exist


def test_submissions_in_flight(ctx_squeeze, stereo_pair):
    """spvo_detect_dev_submit x6 / spvo_detect_wait x6 through the C ABI: same keypoints and matches as the
    synchronous calls, oldest-first completion, and the documented SPVO_ERR_STATE refusals."""
    import torch
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    dev = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in frames]
    rows, cols = frames[0][0].shape
    args = lambda k: (dev[k][0].data_ptr(), dev[k][1].data_ptr(), rows, cols, dev[k][0].stride(0))
    ctx_squeeze.set_prematch(True, "KNN", False, 0.8)
    ref = [ctx_squeeze.detect_dev(*args(0), P_l, P_r, 0, 1), ctx_squeeze.detect_dev(*args(1), P_l, P_r, 2, 3)]
    ref = [{k: (None if v is None else v.copy()) for k, v in r.items()} for r in ref]
    ref_m = (ctx_squeeze.match_slots(2, 3, len(ref[1]["xy_l"])), ctx_squeeze.match_slots(2, 0, len(ref[1]["xy_l"])))
    with pytest.raises(capi.SpvoError) as e:
        ctx_squeeze.detect_wait(P_l, P_r)                                     # nothing in flight
    assert e.value.code == -4
    ctx_squeeze.detect_dev_submit(*args(0), 4, 5)
    ctx_squeeze.detect_dev_submit(*args(1), 6, 7)
    for k, sl in enumerate(((8, 9), (10, 11), (12, 13), (14, 15))):          # ... up to six: the limit
        ctx_squeeze.detect_dev_submit(*args(k & 1), *sl)
    with pytest.raises(capi.SpvoError) as e:
        ctx_squeeze.detect_dev_submit(*args(1), 0, 1)                         # a seventh one
    assert e.value.code == -4
    with pytest.raises(capi.SpvoError) as e:
        ctx_squeeze.forward(np.zeros((1, 1, 360, 1176), np.float32))          # would overwrite the activations
    assert e.value.code == -4
    a = ctx_squeeze.detect_wait(P_l, P_r)
    with pytest.raises(capi.SpvoError) as e:
        ctx_squeeze.detect_dev_submit(*args(0), 6, 7)                         # slots of the submission still in flight
    assert e.value.code == -4
    b = ctx_squeeze.detect_wait(P_l, P_r)
    c3 = ctx_squeeze.detect_wait(P_l, P_r)
    rest = [ctx_squeeze.detect_wait(P_l, P_r) for _ in range(3)]
    for k, r in enumerate(rest):
        assert np.array_equal(r["xy_l"], ref[(k + 1) & 1]["xy_l"]) and np.array_equal(r["xy_r"], ref[(k + 1) & 1]["xy_r"])
    assert np.array_equal(c3["xy_l"], ref[0]["xy_l"]) and np.array_equal(c3["xy_r"], ref[0]["xy_r"])
    for got, want in ((a, ref[0]), (b, ref[1])):
        assert np.array_equal(got["xy_l"], want["xy_l"]) and np.array_equal(got["xy_r"], want["xy_r"])
        assert np.array_equal(got["P_l"], want["P_l"])
    n = len(b["xy_l"])
    got_m = (ctx_squeeze.match_slots(6, 7, n), ctx_squeeze.match_slots(6, 4, n))    # stereo, temporal: from the submission's cache
    for (gi, gd), (ri, rd) in zip(got_m, ref_m):
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    ctx_squeeze.set_prematch(False, "KNN", False, 0.8)


def _mbv1_int8_engine(tmp_path, frames):
    """sp_mbv1 as an INT8 engine at 360x1176, calibrated on the device on the first frame (spvo/quant.py)"""
    from spvo import quant, weights
    from tests.conftest import GOLDEN
    plan = weights.load(os.path.join(GOLDEN, "sp_mbv1.spvw"))
    plan.act_scales = quant.calibrate(plan, [quant.calibration_inputs(plan, frames[0], 360, 1176)], 360, 1176)
    path = str(tmp_path / weights.engine_name("sp_mbv1", 2, 360, 1176, "INT8"))
    weights.save(plan, path, precision="INT8")
    return path


def test_int8_blocks_in_one_launch_equal_the_layerwise_engine(stereo_pair, tmp_path, tuning):
    """INT8 engines run a MobileNet block (depthwise 3x3 -> pointwise 1x1, the first one with the fp32 stem in front) as ONE launch
    (csrc/conv_i8_fused.hip.h; tuning "int8_fused" = 0 keeps one launch per layer).  Same arithmetic in the same order: the network
    outputs, keypoints and descriptors of a detector pass are bit-identical either way -- in the pipelined entry points too, where the
    fused kernels do not store the tensors they skip (tests/test_gpu_network.py::test_int8_engine_is_bit_exact covers every tensor
    through the synchronous entry points, where they do)."""
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    path = _mbv1_int8_engine(tmp_path, frames)
    out = {}
    for fused in (0, 1):
        tuning(int8_fused=fused)
        ctx = capi.Context()
        ctx.load_weights(path)
        assert ctx.engine_precision() == "INT8"
        r = ctx.detect(frames[0][0], frames[0][1], P_l, P_r, 0, 1)
        x = np.stack([fe.to_network_input(fe.preprocess(img, P_l, 360, 1176, True)[0]) for img in frames[1]])[:, None]
        det, desc = ctx.forward(x)
        out[fused] = ({k: np.array(r[k]) for k in ("xy_l", "xy_r", "desc_l", "desc_r")}, det.copy(), desc.copy())
        ctx.close()
    for key in out[0][0]:
        assert np.array_equal(out[0][0][key], out[1][0][key]), key
    assert len(out[1][0]["xy_l"]) > 300
    assert np.array_equal(out[0][1], out[1][1]) and np.array_equal(out[0][2], out[1][2])


@pytest.mark.parametrize("graph", ["vgg", "squeeze", "vgg_fp16", "mbv1_int8"])
def test_trunk_pairing_does_not_change_results(graph, vgg_weights_path, squeeze_weights_path, stereo_pair, vgg_plan, tmp_path):
    """spvo_set_trunk_pairing: a submission whose network would only queue is held until the next one arrives and the two pairs
    run through every layer in ONE launch (four images).  Keypoints, descriptors (through the pinned mirrors) and both matches of
    every pair are bit-identical to the unpaired run -- the kernels are the ones selected for two images and every tile is
    computed the same way wherever it runs -- whether a pair ends up grouped (2 + 2), alone because nothing was queued, or alone
    because it was waited for while held."""
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    seq = [frames[k & 1] for k in range(7)]
    path = vgg_weights_path if graph == "vgg" else squeeze_weights_path
    if graph == "vgg_fp16":          # the FP16 engine's kernels take the batch from the launch as well
        import copy
        from spvo import weights
        p16 = copy.copy(vgg_plan)
        p16.precision = "FP16"
        path = str(tmp_path / weights.engine_name("superpoint_pretrained", 2, 360, 1176, "FP16"))
        weights.save(p16, path)
    if graph == "mbv1_int8":         # ... and so do the INT8 engine's, the fused MobileNet blocks included
        path = _mbv1_int8_engine(tmp_path, frames)
    out = {}
    for pairing in (False, True):
        ctx = capi.Context()
        ctx.load_weights(path)
        ctx.set_prematch(True, "KNN", False, 0.8)
        ctx.set_trunk_pairing(pairing)
        res = []

        def collect(slot_l):
            v = ctx.detect_collect_mirrors(P_l, P_r)
            n = len(v["xy_l"])
            m = [ctx.match_slots(slot_l, slot_l + 1, n)] + ([ctx.match_slots(slot_l, (slot_l - 2) % 16, n)] if res else [])
            res.append(({k: np.array(v[k]) for k in ("xy_l", "xy_r", "desc_l", "desc_r", "resized_l")}, [(i.copy(), d.copy()) for i, d in m]))
        # pairs 0..3 handed over in one go (pair 0 runs alone, 1 is held, 2 joins it, 3 is held and launched by its own wait), then one
        # at a time behind a collect (each finds its predecessor's trunk queued or done), then a last one
        for k in range(4):
            ctx.detect_submit(seq[k][0], seq[k][1], 2 * k, 2 * k + 1)
        collect(0); collect(2)
        ctx.detect_submit(seq[4][0], seq[4][1], 8, 9)
        collect(4)
        ctx.detect_submit(seq[5][0], seq[5][1], 10, 11)
        ctx.detect_submit(seq[6][0], seq[6][1], 12, 13)
        collect(6); collect(8); collect(10); collect(12)
        out[pairing] = res
        ctx.close()
    for k, ((fa, ma), (fb, mb)) in enumerate(zip(out[False], out[True])):
        for key in fa:
            assert np.array_equal(fa[key], fb[key]), (k, key)
        for (ia, da), (ib, db) in zip(ma, mb):
            assert np.array_equal(ia, ib) and np.array_equal(da, db), k


def test_preprocess_inside_the_first_layer_is_bit_identical(vgg_weights_path, stereo_pair, tuning):
    """Round 5: a submission's crop / resize / normalise runs inside its group's first layer (csrc/conv_first_pre.hip.h) instead of in a
    launch of its own (tuning "preprocess_fused" = 0).  Same integer arithmetic, same fp32 planes: every output -- resized u8 images,
    keypoints, descriptors, matches -- is identical bit for bit, for pairs grouped two per launch, for a pair alone, and when a pair of
    ANOTHER image size arrives while one is held (one launch has one crop geometry: the held pair then goes first, alone)."""
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    small = [(np.ascontiguousarray(L[4:-4, 8:-9]), np.ascontiguousarray(R[4:-4, 8:-9])) for L, R in frames]     # another size -> another crop
    order = [frames[0], frames[1], small[0], frames[0], small[1], small[0]]
    out = {}
    for fused in (1, 0):
        tuning(preprocess_fused=fused)
        ctx = capi.Context()
        ctx.load_weights(vgg_weights_path)
        ctx.set_prematch(True, "KNN", False, 0.8)
        ctx.set_trunk_pairing(True)
        res = []
        for k, (L, R) in enumerate(order):
            ctx.detect_submit(L, R, 2 * k, 2 * k + 1)
        for k in range(len(order)):
            v = ctx.detect_collect_mirrors(P_l, P_r)
            n = len(v["xy_l"])
            m = ctx.match_slots(2 * k, 2 * k + 1, n)
            res.append({**{key: np.array(v[key]) for key in ("xy_l", "xy_r", "desc_l", "desc_r", "resized_l", "resized_r")}, "mi": m[0].copy(), "md": m[1].copy()})
        out[fused] = res
        ctx.close()
    for k, (a, b) in enumerate(zip(out[1], out[0])):
        for key in a:
            assert np.array_equal(a[key], b[key]), (k, key)
    assert len(out[1][2]["xy_l"]) > 100 and len(out[1][0]["xy_l"]) > 100


@pytest.mark.parametrize("graph", ["squeeze", "vgg_fp16", "mbv1_int8"])
def test_two_tail_streams_do_not_change_results(graph, squeeze_weights_path, stereo_pair, vgg_plan, tmp_path, tuning):
    """spvo_set_tuning("tail_streams", 2): consecutive submissions' tails (heat map, NMS, sampling, the two matches) alternate between
    two streams instead of queueing on one -- what crosses between them is ordered by events (the temporal partner's features, the heads
    of a paired group) and by hand-over on the own stream (the clean NMS counter block, two sets ahead).  Keypoints, descriptors, resized
    images and both matches of every pair are bit-identical to the one-stream run, with trunk pairing and without."""
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    seq = [frames[k & 1] for k in range(8)]
    path = squeeze_weights_path
    if graph == "vgg_fp16":
        import copy
        from spvo import weights
        p16 = copy.copy(vgg_plan)
        p16.precision = "FP16"
        path = str(tmp_path / weights.engine_name("superpoint_pretrained", 2, 360, 1176, "FP16"))
        weights.save(p16, path)
    if graph == "mbv1_int8":
        path = _mbv1_int8_engine(tmp_path, frames)
    out = {}
    for tails, pairing in ((1, False), (2, False), (2, True)):
        tuning(tail_streams=tails)
        ctx = capi.Context()
        ctx.load_weights(path)
        ctx.set_prematch(True, "KNN", False, 0.8)
        ctx.set_trunk_pairing(pairing)
        res = []

        def collect(slot_l):
            v = ctx.detect_collect_mirrors(P_l, P_r)
            n = len(v["xy_l"])
            m = [ctx.match_slots(slot_l, slot_l + 1, n)] + ([ctx.match_slots(slot_l, (slot_l - 2) % 16, n)] if res else [])
            res.append(({k: np.array(v[k]) for k in ("xy_l", "xy_r", "desc_l", "desc_r", "resized_l")}, [(i.copy(), d.copy()) for i, d in m]))
        for k in range(5):
            ctx.detect_submit(seq[k][0], seq[k][1], 2 * k, 2 * k + 1)
        collect(0); collect(2)
        ctx.detect_submit(seq[5][0], seq[5][1], 10, 11)
        ctx.detect_submit(seq[6][0], seq[6][1], 12, 13)
        collect(4); collect(6); collect(8)
        ctx.detect_submit(seq[7][0], seq[7][1], 14, 15)
        collect(10); collect(12); collect(14)
        out[(tails, pairing)] = res
        ctx.close()
    ref = out[(1, False)]
    for key in ((2, False), (2, True)):
        for k, ((fa, ma), (fb, mb)) in enumerate(zip(ref, out[key])):
            for name in fa:
                assert np.array_equal(fa[name], fb[name]), (key, k, name)
            for (ia, da), (ib, db) in zip(ma, mb):
                assert np.array_equal(ia, ib) and np.array_equal(da, db), (key, k)
    assert len(ref) == 8 and all(len(f["xy_l"]) > 100 for f, _ in ref)


def test_a_failed_group_launch_keeps_the_queue_consistent(squeeze_weights_path, stereo_pair, tuning):
    """Trunk pairing, error path: the launch of a held group fails (injected: the context's fourth group launch) while one of its two
    members had already been accepted.  The submit that triggers the launch returns the error and is NOT queued; the member accepted
    earlier stays queued and its spvo_detect_wait reports SPVO_ERR_STATE instead of waiting for events that were never recorded (and
    handing out another submission's stale keypoints); nothing is left in flight, and the context keeps working afterwards."""
    import torch
    from spvo import capi
    frames, _, P_l, P_r = stereo_pair
    dev = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in frames]
    rows, cols = frames[0][0].shape
    args = lambda k: (dev[k][0].data_ptr(), dev[k][1].data_ptr(), rows, cols, dev[k][0].stride(0))
    tuning(inject_launch_failure=4)                       # read at spvo_create: launches 1, 2 = the reference pairs, 3 = the group (A, B), 4 = the group (C, D)
    ctx = capi.Context()
    ctx.load_weights(squeeze_weights_path)
    ref = [ctx.detect_dev(*args(k), P_l, P_r, 0, 1)["xy_l"].copy() for k in (0, 1)]   # launches 1 and 2
    ctx.set_trunk_pairing(True)
    ctx.detect_dev_submit(*args(0), 2, 3)                 # A: held for a partner
    ctx.detect_dev_submit(*args(1), 4, 5)                 # B: completes the group, launch 3
    ctx.detect_dev_submit(*args(0), 6, 7)                 # C: held
    with pytest.raises(capi.SpvoError) as e:
        ctx.detect_dev_submit(*args(1), 8, 9)             # D: completes the group, whose launch fails
    assert "injected" in str(e.value)
    assert np.array_equal(ctx.detect_wait(P_l, P_r)["xy_l"], ref[0])      # A and B are untouched
    assert np.array_equal(ctx.detect_wait(P_l, P_r)["xy_l"], ref[1])
    with pytest.raises(capi.SpvoError) as e:
        ctx.detect_wait(P_l, P_r)                         # C: accepted earlier, never launched
    assert e.value.code == -4 and "launch had failed" in str(e.value)
    with pytest.raises(capi.SpvoError) as e:
        ctx.detect_wait(P_l, P_r)                         # D was never queued
    assert e.value.code == -4
    ctx.detect_dev_submit(*args(0), 10, 11)               # the context keeps working (the failed members' slots are free again)
    ctx.detect_dev_submit(*args(1), 6, 7)
    assert np.array_equal(ctx.detect_wait(P_l, P_r)["xy_l"], ref[0])
    assert np.array_equal(ctx.detect_wait(P_l, P_r)["xy_l"], ref[1])
    # ... and a held pair that is asked for before its partner arrives is launched alone, by the wait
    ctx.detect_dev_submit(*args(0), 12, 13)
    assert np.array_equal(ctx.detect_wait(P_l, P_r)["xy_l"], ref[0])
    ctx.close()


def test_keypoint_cap_2048(vgg_weights_path, vgg_plan, stereo_pair):
    """BASELINE config 5 raises the reference's static cap of 1000 keypoints (hpp:368) to 2048: the cap is a
    context parameter; NMS (on the same heat map) stays bit-exact against the oracle and the matcher handles 2048 x 2048."""
    from tests.conftest import make_ctx
    frames, _, P_l, P_r = stereo_pair
    L, R = frames[0]
    ctx = make_ctx(vgg_weights_path, max_keypoints=2048)
    out = ctx.detect(L, R, P_l, P_r, 0, 1)
    assert len(out["xy_l"]) == 2048 and len(out["xy_r"]) == 2048            # the seeded VGG yields ~13 k candidates
    x = fe.preprocess(L, P_l, 360, 1176, True)[0].astype(np.float32)[None, None] / 255.0
    det, _ = ctx.forward(x)
    heat = ctx.heatmap(det[0])
    assert np.array_equal(ctx.nms(heat), fe.nms(heat, 0.015, 4, 4, 2048))   # same heat map -> bit-exact keypoints, 2048 of them
    # the headline workload's descriptors (seeded VGG: clusters of numerically identical rows) at the raised cap, every
    # selector, every row: indices and distances equal the brute-force oracle's (base.cpp:462-473)
    for selector, cross in (("KNN", False), ("NN", False), ("NN", True)):
        idx, d = ctx.match_slots(0, 1, 2048, selector, cross)
        ridx, rd = matching.bf_match(out["desc_l"], out["desc_r"], selector, cross, 0.8)
        assert np.array_equal(idx, ridx) and np.array_equal(d, rd), (selector, cross)
    ctx.close()


def test_matches_on_the_headline_workload_are_exact(ctx_vgg, stereo_pair):
    """What bench.py times: seeded VGG, 360x1176, cap 1000, the stereo and the temporal match.  Every row of every
    selector equals the oracle's brute-force matcher bit for bit (indices and distances)."""
    frames, _, P_l, P_r = stereo_pair
    prev = ctx_vgg.detect(frames[0][0], frames[0][1], P_l, P_r, 0, 1)
    cur = ctx_vgg.detect(frames[1][0], frames[1][1], P_l, P_r, 2, 3)
    n = len(cur["xy_l"])
    assert n == 1000
    for (sa, sb, da, db) in ((2, 3, cur["desc_l"], cur["desc_r"]), (2, 0, cur["desc_l"], prev["desc_l"])):
        for selector, cross in (("KNN", False), ("NN", False), ("NN", True)):
            idx, d = ctx_vgg.match_slots(sa, sb, n, selector, cross)
            ridx, rd = matching.bf_match(da, db, selector, cross, 0.8)
            assert np.array_equal(idx, ridx) and np.array_equal(d, rd), (sa, sb, selector, cross)


@pytest.mark.parametrize("tails", [1, 2])
def test_nms_redo_with_two_submissions_in_flight(vgg_plan, tmp_path, tails, tuning):
    """(with one tail stream and with two: the continuation, the re-sampling and the rematch run on the submission's OWN tail stream)
    The rare path of spvo_detect_wait: the first batch of NMS rounds leaves candidates undecided, so the host
    enqueues more rounds, re-samples the descriptors and redoes the matches of that submission AND the temporal
    match of the younger submission that had already matched against the replaced keypoints.  Forced here with
    all-zero images and a threshold below the uniform response (every pixel is a candidate and ties with its
    neighbours: one decision chain across the whole picture)."""
    import torch
    from spvo import weights
    from tests.conftest import make_ctx
    H, W = 120, 392
    path = str(tmp_path / weights.engine_name("superpoint_pretrained", 2, H, W, "FP32"))
    weights.save(vgg_plan, path)
    tuning(tail_streams=tails)
    ctx = make_ctx(path, net_height=H, net_width=W, conf_thresh=0.001)
    P_l, P_r = synth.projection_matrices()
    z = torch.zeros((H, W), dtype=torch.uint8, device="cuda")
    args = (z.data_ptr(), z.data_ptr(), H, W, z.stride(0))
    ctx.set_prematch(True, "NN", False, 0.8)
    a = ctx.detect_dev(*args, P_l, P_r, 0, 1)
    n = len(a["xy_l"])
    assert n > 100                                                          # one keypoint per 5x5 cell of the interior
    ref_xy = a["xy_l"].copy()
    ref_b = ctx.detect_dev(*args, P_l, P_r, 2, 3)
    ref_m = (ctx.match_slots(2, 3, n, "NN", False), ctx.match_slots(2, 0, n, "NN", False))
    ref_m = tuple((i.copy(), d.copy()) for i, d in ref_m)
    assert np.array_equal(ref_b["xy_l"], ref_xy)
    ctx.detect_dev_submit(*args, 4, 5)
    ctx.detect_dev_submit(*args, 6, 7)
    r1 = ctx.detect_wait(P_l, P_r)
    r2 = ctx.detect_wait(P_l, P_r)
    assert np.array_equal(r1["xy_l"], ref_xy) and np.array_equal(r2["xy_l"], ref_xy) and np.array_equal(r2["xy_r"], ref_xy)
    got = (ctx.match_slots(6, 7, n, "NN", False), ctx.match_slots(6, 4, n, "NN", False))
    for (gi, gd), (ri, rd) in zip(got, ref_m):
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    prof = ctx.profile()
    assert prof["nms_redo"]["calls"] >= 4 and prof["rematch"]["calls"] >= 1      # the path under test really ran
    ctx.close()


def test_fp8_shortlist_matcher(ctx_squeeze, stereo_pair):
    """BASELINE config 5: the shortlist of the matcher from an fp8 (e4m3) distance GEMM.  The GEMM only prunes: pass 1 re-scores
    a statistical window canonically, pass 2 every column whose RIGOROUS lower bound (per-row norms of the fp8 rounding residuals,
    csrc/match.hip.h: MATCH_ERR_REL_FP8) does not exceed the second canonical distance found.  So indices AND distances equal
    the brute-force oracle's on every row, confident or borderline -- KNN ratio test and plain NN."""
    frames, _, P_l, P_r = stereo_pair
    L, R = frames[0]
    out = ctx_squeeze.detect(L, R, P_l, P_r, 0, 1)
    n = len(out["xy_l"])
    ridx, rd = matching.bf_match(out["desc_l"], out["desc_r"], "KNN", False, 0.8)
    r_nn, r_nnd = matching.bf_match(out["desc_l"], out["desc_r"], "NN", False, 0.8)
    try:
        ctx_squeeze.set_match_fp8(True)
        idx, d = ctx_squeeze.match_slots(0, 1, n)
        nn_idx, nn_d = ctx_squeeze.match_slots(0, 1, n, "NN", False)
        # descriptors handed over by the caller (spvo_match) take the same path; a near-duplicate cluster widens every window
        rng = np.random.RandomState(3)
        a = rng.randn(700, 256).astype(np.float32)
        a /= np.linalg.norm(a, axis=1, keepdims=True)
        b = np.concatenate([a[:300] + 0.02 * rng.randn(300, 256).astype(np.float32), np.repeat(a[300:301], 200, 0) + 1e-3 * rng.randn(200, 256).astype(np.float32),
                            rng.randn(150, 256).astype(np.float32)])
        b /= np.linalg.norm(b, axis=1, keepdims=True)
        b = b.astype(np.float32)
        sidx, sd = ctx_squeeze.match(a, b)
    finally:
        ctx_squeeze.set_match_fp8(False)
    assert (ridx >= 0).sum() > 300
    assert np.array_equal(idx, ridx) and np.array_equal(d, rd)
    assert np.array_equal(nn_idx, r_nn) and np.array_equal(nn_d, r_nnd)
    s_ridx, s_rd = matching.bf_match(a, b, "KNN", False, 0.8)
    assert np.array_equal(sidx, s_ridx) and np.array_equal(sd, s_rd)
    idx2, d2_ = ctx_squeeze.match_slots(0, 1, n)                              # back to the fp32 shortlist: the same again
    assert np.array_equal(idx2, ridx) and np.array_equal(d2_, rd)


def test_host_image_submissions_are_bit_identical_to_the_synchronous_entry(ctx_squeeze, stereo_pair):
    """spvo_detect_submit / spvo_detect_collect (host images through pinned staging, two submissions in flight, resized
    images and descriptors in the submissions' pinned mirrors) against spvo_detect on the same pairs: every output
    bit for bit -- keypoints, descriptors, the resized u8 images, the projection matrices -- and the same matches."""
    frames, _, P_l, P_r = stereo_pair
    ref = [ctx_squeeze.detect(L, R, P_l, P_r, 2 * k, 2 * k + 1, want_resized=True) for k, (L, R) in enumerate(frames[:2])]
    n = len(ref[1]["xy_l"])
    ref_m = (ctx_squeeze.match_slots(2, 3, n), ctx_squeeze.match_slots(2, 0, n))
    ref_m = tuple((i.copy(), d.copy()) for i, d in ref_m)
    ctx_squeeze.set_prematch(True, "KNN", False, 0.8)
    ctx_squeeze.detect_submit(frames[0][0].copy(), frames[0][1].copy(), 4, 5)     # temporaries: the call copies them before it returns
    ctx_squeeze.detect_submit(frames[1][0].copy(), frames[1][1].copy(), 6, 7)
    got = [ctx_squeeze.detect_collect(P_l, P_r), ctx_squeeze.detect_collect(P_l, P_r)]
    for g, r in zip(got, ref):
        for k in ("xy_l", "xy_r", "desc_l", "desc_r", "resized_l", "resized_r", "P_l", "P_r"):
            assert np.array_equal(g[k], r[k]), k
    got_m = (ctx_squeeze.match_slots(6, 7, n), ctx_squeeze.match_slots(6, 4, n))
    for (gi, gd), (ri, rd) in zip(got_m, ref_m):
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd)
    ctx_squeeze.set_prematch(False, "KNN", False, 0.8)
    # the same results as VIEWS of the submissions' pinned mirrors (spvo_detect_collect_mirrors: what the host class copies once into
    # images_dq / descriptors_dq); the first pair's views stay intact while the second pair is collected
    ctx_squeeze.detect_submit(frames[0][0].copy(), frames[0][1].copy(), 4, 5)       # (slot 6 is this submission's temporal partner: not reusable yet)
    ctx_squeeze.detect_submit(frames[1][0].copy(), frames[1][1].copy(), 8, 9)
    views = [ctx_squeeze.detect_collect_mirrors(P_l, P_r), ctx_squeeze.detect_collect_mirrors(P_l, P_r)]
    for g, r in zip(views, ref):
        for k in ("xy_l", "xy_r", "desc_l", "desc_r", "resized_l", "resized_r", "P_l", "P_r"):
            assert np.array_equal(g[k], r[k]), k
    ctx_squeeze.detect_submit(frames[0][0], frames[0][1], 0, 1, extras=1)          # descriptors not requested: no view of them
    v = ctx_squeeze.detect_collect_mirrors(P_l, P_r)
    assert v["desc_l"] is None and np.array_equal(v["resized_r"], ref[0]["resized_r"]) and np.array_equal(v["xy_l"], ref[0]["xy_l"])
    # extras not requested at submit time cannot be fetched while a younger submission is in flight -- and the refusal
    # leaves the queue intact (the oldest submission is still collectable)
    from spvo import capi
    ctx_squeeze.detect_submit(frames[0][0], frames[0][1], 0, 1, extras=0)
    ctx_squeeze.detect_submit(frames[1][0], frames[1][1], 2, 3, extras=0)
    with pytest.raises(capi.SpvoError) as e:
        ctx_squeeze.detect_collect(P_l, P_r)
    assert e.value.code == -4
    a = ctx_squeeze.detect_collect(P_l, P_r, want_desc=False, want_resized=False)
    b = ctx_squeeze.detect_collect(P_l, P_r, want_desc=True, want_resized=True)   # the last one in flight may fetch anything
    assert np.array_equal(a["xy_l"], ref[0]["xy_l"]) and np.array_equal(b["desc_r"], ref[1]["desc_r"]) and np.array_equal(b["resized_l"], ref[1]["resized_l"])
