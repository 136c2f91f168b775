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
