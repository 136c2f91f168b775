"""GPU parity of K0, K7-K11 against the oracle.  Integer/byte outputs bit-exact."""
import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import frontend as fe, net
from tests.conftest import make_ctx

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("H,W", [(360, 1176), (240, 784), (120, 392), (376, 1240), (192, 640)])
@pytest.mark.parametrize("bug", [1, 0])
def test_preprocess_bit_exact(vgg_weights_path, kitti_P, H, W, bug):
    rng = np.random.RandomState(H + bug)
    img = rng.randint(0, 256, (376, 1241)).astype(np.uint8)
    ctx = make_ctx(vgg_weights_path, net_height=H, net_width=W, bug_compat_p=bug)
    got, P = ctx.preprocess(img, kitti_P[1])
    ref, Pref = fe.preprocess(img, kitti_P[1], H, W, bool(bug))
    assert np.array_equal(got, ref)
    assert np.array_equal(P.view(np.uint64), Pref.view(np.uint64))       # f64 bit patterns, incl. the denormal
    ctx.close()


def test_preprocess_strided_rows_and_sample_size(vgg_weights_path, kitti_P, sample_images):
    ctx = make_ctx(vgg_weights_path)
    big = np.zeros((375, 1300), np.uint8)
    big[:, :1242] = sample_images[0]
    view = big[:, :1242]                                                   # stride 1300 != cols
    got, _ = ctx.preprocess(view, kitti_P[0])
    ref, _ = fe.preprocess(sample_images[0], kitti_P[0], 360, 1176)
    assert np.array_equal(got, ref)
    ctx.close()


def test_heatmap(ctx_squeeze, squeeze_plan, sample_images):
    x = (sample_images[0][:360, :1176].astype(np.float32) / 255)[None, None]
    det, _ = net.forward(squeeze_plan, x)
    heat = ctx_squeeze.heatmap(det[0])
    ref = fe.heatmap(det[0])
    # expf (ocml) vs numpy exp: <= 2 ulp each; heat values are <= 1
    assert np.abs(heat - ref).max() <= 2e-6
    assert np.all((heat > 0.015) == (ref > 0.015)) or np.sum((heat > 0.015) != (ref > 0.015)) <= 3


def _nms_both(ctx, heat):
    got = ctx.nms(heat)
    ref = fe.nms(heat, ctx.cfg.conf_thresh, ctx.cfg.dist_thresh, ctx.cfg.border_remove, ctx.cfg.max_keypoints)
    return got, ref


def test_nms_bit_exact_on_real_heatmap(ctx_squeeze, squeeze_plan, sample_images):
    for k in range(2):
        x = (sample_images[k][:360, :1176].astype(np.float32) / 255)[None, None]
        det, _ = net.forward(squeeze_plan, x)
        heat = fe.heatmap(det[0])
        got, ref = _nms_both(ctx_squeeze, heat)
        assert len(ref) == 1000 and np.array_equal(got, ref)


def test_nms_edge_cases(vgg_weights_path):
    ctx = make_ctx(vgg_weights_path, net_height=120, net_width=392, max_keypoints=50)
    H, W = 120, 392
    # empty
    got, ref = _nms_both(ctx, np.zeros((H, W), np.float32))
    assert len(got) == 0 and len(ref) == 0
    # everything a candidate with identical confidence: pure tie-break order, long suppression chains
    got, ref = _nms_both(ctx, np.full((H, W), 0.5, np.float32))
    assert np.array_equal(got, ref) and len(ref) == 50
    # monotone ramp: the worst case for parallel rounds (each decision depends on its neighbour)
    ramp = (np.arange(H * W, dtype=np.float32).reshape(H, W) + 1) / (H * W)
    got, ref = _nms_both(ctx, ramp)
    assert np.array_equal(got, ref)
    # random field with many exact ties, threshold strictness, border points
    rng = np.random.RandomState(0)
    heat = (rng.randint(0, 40, (H, W)) / 40.0).astype(np.float32)
    heat[0, :] = 1.0
    heat[:, W - 1] = 1.0
    got, ref = _nms_both(ctx, heat)
    assert np.array_equal(got, ref)
    ctx.close()
    # uncapped, other radius/border
    ctx = make_ctx(vgg_weights_path, net_height=120, net_width=392, max_keypoints=4000, dist_thresh=2, border_remove=0)
    heat = rng.rand(H, W).astype(np.float32)
    got, ref = _nms_both(ctx, heat)
    assert np.array_equal(got, ref) and len(ref) > 1000
    ctx.close()


@pytest.mark.parametrize("nms_first", [1, 2, 3, 6])
def test_nms_result_does_not_depend_on_the_launches_enqueued_with_a_submission(vgg_weights_path, tuning, nms_first):
    """`nms_first` (diagnostic switch; default 4 = three round launches + the finishing kernel): a single round launch, one round
    launch + the finishing kernel, two + it, five + it -- what the first batch leaves undecided is continued by the host, and the
    keypoints are the oracle's in every case (dense map: every pixel a candidate; clustered map: blobs that need several rounds)."""
    tuning(nms_first=nms_first)
    H, W = 120, 392
    ctx = make_ctx(vgg_weights_path, net_height=H, net_width=W, max_keypoints=1000)
    rng = np.random.RandomState(nms_first)
    dense = (0.02 + 0.9 * rng.rand(H, W)).astype(np.float32)
    clustered = np.full((H, W), 0.001, np.float32)
    for _ in range(100):
        cy, cx = rng.randint(0, H), rng.randint(0, W)
        y0, y1, x0, x1 = max(cy - 6, 0), min(cy + 7, H), max(cx - 6, 0), min(cx + 7, W)
        clustered[y0:y1, x0:x1] = 0.1 + 0.8 * rng.rand(y1 - y0, x1 - x0).astype(np.float32)
    for heat in (dense, clustered):
        got, ref = _nms_both(ctx, heat)
        assert np.array_equal(got, ref)
    ctx.close()


def test_nms_long_chains_are_finished_on_the_device(vgg_weights_path):
    """Decision chains far longer than the three round launches' twelve rounds -- rows of candidates with confidences falling along
    the row: every fifth one is kept, and each decision waits for the one before it -- are settled by nms_finish_kernel (one workgroup
    per image, rounds separated by workgroup barriers) without the host's continuation; a map whose stragglers do not fit that
    kernel's list (every pixel a link of one chain) still takes the host path.  Both bit-exact against the sequential oracle."""
    H, W = 120, 392
    ctx = make_ctx(vgg_weights_path, net_height=H, net_width=W, max_keypoints=1000)
    heat = np.full((H, W), 0.001, np.float32)
    for r, y in enumerate(range(8, H - 8, 30)):                      # four rows, far enough apart not to interact (well inside the kernel's budget of window evaluations)
        heat[y, :] = (0.9 - 0.002 * np.arange(W) - 0.0001 * r).astype(np.float32)   # ~78 links per chain
    got, ref = _nms_both(ctx, heat)
    assert np.array_equal(got, ref) and len(ref) > 250
    assert ctx.profile().get("nms_redo", {"calls": 0})["calls"] == 0                   # no host continuation: the finishing kernel settled it
    ramp = (np.arange(H * W, dtype=np.float32).reshape(H, W) + 1) / (H * W)
    got, ref = _nms_both(ctx, ramp)
    assert np.array_equal(got, ref)
    assert ctx.profile()["nms_redo"]["calls"] >= 1                   # 47 k undecided candidates: the host's continuation
    ctx.close()


@pytest.mark.parametrize("H,W,dist,border", [(360, 1176, 4, 4), (120, 392, 4, 4), (120, 392, 2, 0), (120, 392, 8, 3), (192, 640, 3, 4)])
def test_nms_on_sparse_tied_clustered_and_dense_heat_maps(vgg_weights_path, H, W, dist, border):
    """processOneHeatmap on synthetic heat maps of four kinds -- sparse with distinct confidences, many exact ties and near-ties,
    blobs that need several rounds of decisions per neighbourhood, every pixel a candidate -- at several sizes, radii and
    borders: bit-exact keypoints against the oracle's sequential greedy suppression."""
    ctx = make_ctx(vgg_weights_path, net_height=H, net_width=W, max_keypoints=1000, dist_thresh=dist, border_remove=border)
    rng = np.random.RandomState(H + dist)
    heats = []
    sparse = np.where(rng.rand(H, W) < 0.02, 0.02 + 0.9 * rng.rand(H, W), 0.001).astype(np.float32)
    heats.append(sparse)                                            # ~2 % candidates, distinct confidences
    coarse = sparse.copy()
    m = coarse > 0.015
    coarse[m] = (np.round(coarse[m] * 64) / 64 + 1e-6 * rng.randint(0, 3, m.sum())).astype(np.float32)   # many ties / near-ties
    heats.append(coarse)
    clustered = np.full((H, W), 0.001, np.float32)
    for _ in range(300):                                            # blobs: several rounds of decisions per neighbourhood
        cy, cx = rng.randint(0, H), rng.randint(0, W)
        y0, y1, x0, x1 = max(cy - 6, 0), min(cy + 7, H), max(cx - 6, 0), min(cx + 7, W)
        clustered[y0:y1, x0:x1] = 0.1 + 0.8 * rng.rand(y1 - y0, x1 - x0).astype(np.float32)
    heats.append(clustered)
    heats.append((0.02 + 0.9 * rng.rand(H, W)).astype(np.float32))  # every pixel a candidate
    for heat in heats:
        got, ref = _nms_both(ctx, heat)
        assert np.array_equal(got, ref)
    ctx.close()


def test_sample_descriptors(ctx_squeeze, squeeze_plan, sample_images):
    x = (sample_images[0][:360, :1176].astype(np.float32) / 255)[None, None]
    det, desc = net.forward(squeeze_plan, x)
    xy = fe.nms(fe.heatmap(det[0]))
    # add the extreme in-bounds pixels (nn.cpp:385-386 asserts are compiled out)
    xy = np.concatenate([xy[:300], np.array([[0, 0], [1175, 359], [1175, 0], [0, 359], [4, 4]], np.int32)])
    got = ctx_squeeze.sample_descriptors(np.ascontiguousarray(desc[0].transpose(1, 2, 0)), xy)
    ref = fe.sample_descriptors(desc[0], xy, 360, 1176)
    assert np.abs(got - ref).max() <= 1e-6
    assert len(ctx_squeeze.sample_descriptors(np.ascontiguousarray(desc[0].transpose(1, 2, 0)), xy[:0])) == 0
