"""Oracle for descriptor matching.

TEST INFRASTRUCTURE ONLY.  Restates FeatureFrontEnd::initMatcher (base.cpp:10-33)
and FeatureFrontEnd::matchDescriptors (base.cpp:434-491) with
cv::BFMatcher(NORM_L2) semantics [OpenCV 4.5.4, not installed here: restated from
its published behaviour, "parity unpinned"]:
  * distance = sqrt(sum_k (a_k - b_k)^2) in float32.  OpenCV's SIMD partial-sum
    order is build-dependent; the canonical order pinned here is a sequential
    sum over k with separately rounded multiply and add (the scalar loop of
    normL2Sqr_: `s += t*t`).
  * best match = strict '<' scan over train rows, so the lowest train index wins
    ties; knnMatch(k=2) returns the two best in ascending order.
  * crossCheck (batchDistance, crosscheck=true): for every train row, of the
    queries whose best is that row, only the one with the smallest distance
    (lowest query index on ties) keeps its match.
"""
from __future__ import annotations

import numpy as np


def sq_distances(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """float32 [na, nb], canonical summation order (sequential over k)."""
    a = np.ascontiguousarray(a, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    acc = np.zeros((a.shape[0], b.shape[0]), np.float32)
    for k in range(a.shape[1]):
        t = a[:, k, None] - b[None, :, k]
        acc = acc + t * t
    return acc


def best_two(d2: np.ndarray):
    """Two smallest per row under (distance, index) order.  Returns d2_0, d2_1, i0, i1."""
    na, nb = d2.shape
    i0 = np.argmin(d2, axis=1)                        # first occurrence = lowest index on ties
    r = np.arange(na)
    v0 = d2[r, i0]
    if nb < 2:
        return v0, np.full(na, np.inf, np.float32), i0, np.full(na, -1, np.int64)
    masked = d2.copy()
    masked[r, i0] = np.inf
    i1 = np.argmin(masked, axis=1)
    return v0, masked[r, i1], i0, i1


def bf_match(desc_a: np.ndarray, desc_b: np.ndarray, selector: str = "KNN", cross_check: bool = False,
             ratio: float = 0.8):
    """Returns (train_idx int32 [na] with -1 = no match, distance f32 [na]: the matched pair's distance where
    train_idx >= 0; for NN without cross-check and KNN the nearest neighbour's distance everywhere; 0 for rows a
    cross-check left unmatched).

    train_idx is exactly maps_of_indices[match_type] of base.cpp:483-491; the
    DMatch list of the reference is {(i, train_idx[i], distance[i]) : train_idx[i] >= 0}
    in query order.
    """
    na, nb = len(desc_a), len(desc_b)
    if na == 0 or nb == 0:
        return np.full(na, -1, np.int32), np.zeros(na, np.float32)
    d2 = sq_distances(desc_a, desc_b)
    v0, v1, i0, i1 = best_two(d2)
    d0 = np.sqrt(v0.astype(np.float32))
    d1 = np.sqrt(v1.astype(np.float32))
    out = np.full(na, -1, np.int32)
    if selector == "NN" and cross_check:
        # cv::batchDistance(..., K = 1, crosscheck = true) as BFMatcher::knnMatchImpl calls it (base.cpp:27-28, 463):
        # every TRAIN row looks up its nearest query row (lowest index on ties); a query row keeps, among the train rows
        # that chose it, the nearest one (the first such train row on ties: `if (d < d0)` while i runs upwards) and is
        # unmatched (-1) if no train row chose it.  [third-party semantics restated from memory of
        # modules/core/src/batch_distance.cpp; OpenCV's documentation describes the result as "mutual nearest
        # neighbours", which this procedure contains but does not equal.]
        v0t, _, q_of_t, _ = best_two(d2.T.copy())
        dt = np.sqrt(v0t.astype(np.float32))
        best = np.full(na, np.inf, np.float32)
        d0 = np.zeros(na, np.float32)                  # distance of the kept pair; 0 where unmatched
        for t in range(nb):
            q = int(q_of_t[t])
            if dt[t] < best[q]:
                best[q] = dt[t]
                out[q] = t
                d0[q] = dt[t]
    elif selector == "NN":
        out[:] = i0
    elif selector == "KNN":
        if nb >= 2:
            keep = d0 < np.float32(ratio) * d1          # base.cpp:469
            out[keep] = i0[keep]
    else:
        raise ValueError(selector)
    return out, d0


# ------------------------------------------------------------------ binary descriptors (ORB / BRISK / AKAZE: base.cpp:17-21)
_POPCOUNT8 = np.array([bin(i).count("1") for i in range(256)], np.int32)


def hamming_distances(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    """int32 [na, nb]: cv::NORM_HAMMING = number of differing bits of the two byte strings."""
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    out = np.zeros((a.shape[0], b.shape[0]), np.int32)
    for k in range(a.shape[1]):                       # one byte column at a time keeps the temporary at na x nb bytes
        out += _POPCOUNT8[a[:, k, None] ^ b[None, :, k]]
    return out


def bf_match_hamming(desc_a: np.ndarray, desc_b: np.ndarray, selector: str = "KNN", cross_check: bool = False, ratio: float = 0.8):
    """bf_match for cv::BFMatcher(NORM_HAMMING) (initMatcher, base.cpp:17-21, 27-28): the same three selection procedures on
    integer distances; DMatch::distance is the bit count as a float, the ratio test of base.cpp:469 runs in float32."""
    na, nb = len(desc_a), len(desc_b)
    if na == 0 or nb == 0:
        return np.full(na, -1, np.int32), np.zeros(na, np.float32)
    d = hamming_distances(desc_a, desc_b).astype(np.float32)     # bit counts <= 2040 are exact in float32
    v0, v1, i0, i1 = best_two(d)
    out = np.full(na, -1, np.int32)
    d0 = v0.astype(np.float32)
    if selector == "NN" and cross_check:
        v0t, _, q_of_t, _ = best_two(d.T.copy())
        best = np.full(na, np.inf, np.float32)
        d0 = np.zeros(na, np.float32)
        for t in range(nb):
            q = int(q_of_t[t])
            if v0t[t] < best[q]:
                best[q] = v0t[t]
                out[q] = t
                d0[q] = v0t[t]
    elif selector == "NN":
        out[:] = i0
    elif selector == "KNN":
        if nb >= 2:
            keep = d0 < np.float32(ratio) * v1.astype(np.float32)
            out[keep] = i0[keep]
    else:
        raise ValueError(selector)
    return out, d0
