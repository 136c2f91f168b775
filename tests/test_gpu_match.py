"""GPU parity of K12/K13 (MFMA distance GEMM + exact re-rank) against the oracle: indices bit-exact."""
import os

import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import matching

pytestmark = pytest.mark.gpu


def _unit(rng, n):
    a = rng.randn(n, 256).astype(np.float32)
    return a / np.linalg.norm(a, axis=1, keepdims=True)


def _check(ctx, a, b, selector, cross, ratio=0.8):
    idx, d = ctx.match(a, b, selector, cross, ratio)
    ridx, rd = matching.bf_match(a, b, selector, cross, ratio)
    assert np.array_equal(idx, ridx), (selector, cross, np.nonzero(idx != ridx)[0][:10])
    if len(b):
        assert np.array_equal(d, rd)                                     # canonical summation order: bit-exact floats


@pytest.mark.parametrize("na,nb", [(1000, 1000), (997, 613), (1, 1000), (130, 1), (33, 129), (64, 128), (5, 0)])
def test_match_sizes(ctx_vgg, na, nb):
    rng = np.random.RandomState(na * 7 + nb)
    a = _unit(rng, na)
    b = _unit(rng, nb)
    if nb > 10 and na > 10:                                              # plant true correspondences
        m = min(na, nb) // 2
        b[:m] = a[:m] + 0.03 * rng.randn(m, 256).astype(np.float32)
        b /= np.linalg.norm(b, axis=1, keepdims=True)
    for selector, cross in (("KNN", False), ("NN", False), ("NN", True)):
        _check(ctx_vgg, a, b, selector, cross)


def test_match_ties_and_duplicates(ctx_vgg):
    rng = np.random.RandomState(5)
    a = _unit(rng, 200)
    b = np.concatenate([a[:50], a[:50], a[:50], a[:50], a[:50], _unit(rng, 300)])   # 5 exact copies
    rng.shuffle(b[250:])
    for selector, cross in (("KNN", False), ("NN", False), ("NN", True)):
        _check(ctx_vgg, a, b, selector, cross)
    # all-zero descriptors: every distance ties at 0
    z = np.zeros((40, 256), np.float32)
    _check(ctx_vgg, z, z, "NN", True)
    _check(ctx_vgg, z, z, "KNN", False)


def test_match_golden_fixture(ctx_vgg, golden_dir):
    m = np.load(os.path.join(golden_dir, "oracle_match.npz"))
    idx, d = ctx_vgg.match(m["a"], m["b"], "KNN", False, 0.8)
    assert np.array_equal(idx, m["knn_idx"]) and np.array_equal(d, m["knn_d"])
    idx, _ = ctx_vgg.match(m["a"], m["b"], "NN", True, 0.8)
    assert np.array_equal(idx, m["nn_idx"])


def test_match_unnormalised_large_values(ctx_vgg):
    rng = np.random.RandomState(9)
    a = (rng.randn(300, 256) * 10).astype(np.float32)
    b = (rng.randn(400, 256) * 10).astype(np.float32)
    b[:100] = a[:100] + 0.5 * rng.randn(100, 256).astype(np.float32)
    _check(ctx_vgg, a, b, "KNN", False)
    _check(ctx_vgg, a, b, "NN", True)


def test_match_near_duplicate_clusters(ctx_vgg):
    """Clusters of rows that differ by a few ulps (what an untrained network produces): dozens of train rows sit
    inside the error of the distance GEMM, so the pruning must hand ALL of them to the exact re-rank; ties and
    near-ties resolve under (canonical distance, index) order exactly as the oracle's scan does."""
    rng = np.random.RandomState(11)
    base = _unit(rng, 40)
    rows = []
    for i in range(40):
        reps = 5 + (i * 7) % 90                                            # cluster sizes 5 .. 94: more than one re-score batch
        pert = (rng.randint(-2, 3, size=(reps, 256)) * 2.0 ** -26).astype(np.float32)
        rows.append(base[i] + pert)
    b = np.concatenate(rows + [_unit(rng, 300)])
    b = b[rng.permutation(len(b))]
    a = np.concatenate([base, base + (rng.randint(-1, 2, size=base.shape) * 2.0 ** -25).astype(np.float32), _unit(rng, 100)])
    for selector, cross in (("KNN", False), ("NN", False), ("NN", True)):
        _check(ctx_vgg, a, b, selector, cross)
        _check(ctx_vgg, b, a, selector, cross)
    _check(ctx_vgg, b, b, "NN", True)
    _check(ctx_vgg, b, b, "KNN", False)


def test_match_one_big_cluster_2048(ctx_vgg):
    """2048 x 2048 with every row within 1e-6 of one point: every pair is inside the error window, every row is
    re-scored (32 batches per query)."""
    rng = np.random.RandomState(13)
    c = _unit(rng, 1)
    a = (c + rng.randn(2048, 256) * 3e-8).astype(np.float32)
    b = (c + rng.randn(2048, 256) * 3e-8).astype(np.float32)
    b[100:200] = a[300:400]
    for selector, cross in (("KNN", False), ("NN", True)):
        _check(ctx_vgg, a, b, selector, cross)


@pytest.mark.parametrize("scale", [1e-3, 1.0, 37.0, 1e4])
def test_match_scaled_descriptors(ctx_vgg, scale):
    """the error window scales with |a|^2 + |b|^2: exactness does not depend on unit norms"""
    rng = np.random.RandomState(17)
    a = _unit(rng, 500) * np.float32(scale)
    b = np.concatenate([a[:250] * (1 + rng.randn(250, 1).astype(np.float32) * 1e-7), _unit(rng, 380) * np.float32(scale)])
    for selector, cross in (("KNN", False), ("NN", False), ("NN", True)):
        _check(ctx_vgg, a, b, selector, cross)


@pytest.mark.parametrize("nbytes", [32, 61, 64])
@pytest.mark.parametrize("selector,cross", [("KNN", False), ("NN", False), ("NN", True)])
def test_hamming_matcher_is_bit_exact(ctx_vgg, nbytes, selector, cross):
    """cv::BFMatcher(NORM_HAMMING) for the classic front end's binary descriptors (ORB 32 bytes, AKAZE 61, BRISK 64) through
    spvo_match_hamming against oracle/matching.py: indices and distances exact, for random bit strings, planted duplicates
    (ties between train rows), near-duplicates one bit apart, and the degenerate sizes."""
    rng = np.random.RandomState(nbytes)
    a = rng.randint(0, 256, (1900, nbytes)).astype(np.uint8)
    b = rng.randint(0, 256, (2000, nbytes)).astype(np.uint8)
    b[100:200] = a[300:400]                                             # exact matches
    b[250:300] = a[300:350]                                             # ... twice: the lower train index has to win
    b[400:500] = a[500:600]
    b[400:500, 0] ^= 1                                                  # one bit apart
    a[700:720] = a[699]                                                 # several queries with the same nearest train row (cross-check)
    for aa, bb in ((a, b), (a[:5], b[:1]), (a[:3], b[:2]), (a[:0], b), (a[:7], b[:0])):
        gi, gd = ctx_vgg.match_hamming(aa, bb, selector, cross)
        ri, rd = matching.bf_match_hamming(aa, bb, selector, cross)
        assert np.array_equal(gi, ri) and np.array_equal(gd, rd), (len(aa), len(bb))
    gi, _ = ctx_vgg.match_hamming(a, b, selector, cross)
    if selector == "NN" and not cross:
        assert (gi[300:350] == np.arange(100, 150)).all()              # two identical train rows: the lower index
    if selector == "KNN":
        assert (gi[300:350] == -1).all() and (gi[350:400] == np.arange(150, 200)).all()   # 0 < 0.8 * 0 fails; a single exact partner passes
