"""Oracle for solveStereoOdometry.

TEST INFRASTRUCTURE ONLY.  numpy restatement of
  * FeatureFrontEnd::solveStereoOdometry         base.cpp:125-399
  * CostFunctor32::operator()                     cost.hpp:27-58
("base.cpp" = src/odml_visual_odometry/src/feature_detection_base.cpp,
 "cost.hpp" = src/odml_visual_odometry/include/odml_visual_odometry/ceres_cost_function.hpp)

Third-party pieces, restated from published behaviour ("parity unpinned": OpenCV
4.5.4 and Ceres are not installed here and the reference pins nothing):
  * cv::triangulatePoints: rows x*P[2]-P[0], y*P[2]-P[1] per view, SVD of the 4x4
    in double, last right-singular vector, stored as float32; then
    cv::convertPointsFromHomogeneous: scale = w != 0 ? 1/w : 1 in float32.
  * cv::solvePnPRansac(..., USAC_ACCURATE) CANNOT be reproduced (GC-RANSAC + LO +
    polishing inside OpenCV).  The stand-in pinned here is a deterministic RANSAC:
    500 three-point samples from a counter-based hash, each solved by Newton
    iterations from the motion prior (useExtrinsicGuess=true in the reference),
    scored by squared reprojection error <= 2^2, best = most inliers (earliest
    sample on ties), Gauss-Newton refit on the inliers.
  * ceres::Solve: Levenberg-Marquardt trust region (initial radius 1e4, Jacobi
    scaling, eta 1e-3, function/gradient/parameter tolerances 1e-6/1e-10/1e-8),
    HuberLoss(1.0) through Ceres' corrector (rho'' <= 0 => plain sqrt(rho')
    re-weighting), EigenQuaternionParameterization (x_plus = [sin|d|/|d| d, cos|d|] * x).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

# reference constants: hpp:145-147
TIME_INTERVAL = 0.1
MAX_ACCELERATION = 8.0
IGNORE_FRAME_COUNT = 10


# ----------------------------------------------------------------------------
# quaternion helpers (Eigen coefficient order x, y, z, w)
# ----------------------------------------------------------------------------
def quat_mul(a, b):
    ax, ay, az, aw = a
    bx, by, bz, bw = b
    return np.array([aw * bx + ax * bw + ay * bz - az * by,
                     aw * by - ax * bz + ay * bw + az * bx,
                     aw * bz + ax * by - ay * bx + az * bw,
                     aw * bw - ax * bx - ay * by - az * bz])


def quat_to_rot(q):
    """Eigen::Quaternion::toRotationMatrix (no normalisation)."""
    x, y, z, w = q
    tx, ty, tz = 2 * x, 2 * y, 2 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1 - (txx + tyy)]])


def rvec_to_quat(rvec):
    """base.cpp:274-278: AngleAxis(|r|, r.normalized()) -> Quaterniond."""
    r = np.asarray(rvec, np.float64).reshape(3)
    angle = np.sqrt(r[0] * r[0] + r[1] * r[1] + r[2] * r[2])
    axis = r / angle if angle > 0 else r
    s = np.sin(angle / 2)
    return np.array([axis[0] * s, axis[1] * s, axis[2] * s, np.cos(angle / 2)])


def quat_to_rvec(q):
    x, y, z, w = q
    if w < 0:
        x, y, z, w = -x, -y, -z, -w
    n = np.sqrt(x * x + y * y + z * z)
    if n < 1e-300:
        return np.zeros(3)
    angle = 2 * np.arctan2(n, w)
    return np.array([x, y, z]) / n * angle


# ----------------------------------------------------------------------------
# row M: triangulation  (base.cpp:211-223)
# ----------------------------------------------------------------------------
def triangulate(P_l, P_r, xy_l, xy_r) -> np.ndarray:
    P = [np.asarray(P_l, np.float64).reshape(3, 4), np.asarray(P_r, np.float64).reshape(3, 4)]
    pts = [np.asarray(xy_l, np.float32).reshape(-1, 2), np.asarray(xy_r, np.float32).reshape(-1, 2)]
    n = len(pts[0])
    out = np.zeros((n, 3), np.float32)
    for i in range(n):
        A = np.zeros((4, 4))
        for j in range(2):
            x, y = float(pts[j][i, 0]), float(pts[j][i, 1])
            A[2 * j + 0] = x * P[j][2] - P[j][0]
            A[2 * j + 1] = y * P[j][2] - P[j][1]
        _, _, vt = np.linalg.svd(A)
        h = vt[3].astype(np.float32)                  # points4D is CV_32F for Point2f input
        scale = np.float32(1.0) / h[3] if h[3] != 0 else np.float32(1.0)
        out[i] = h[:3] * scale
    return out


# ----------------------------------------------------------------------------
# row O: deterministic RANSAC stand-in for solvePnPRansac  (base.cpp:227-239)
# ----------------------------------------------------------------------------
def hash32(x: int) -> int:
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def sample_triplet(seed: int, it: int, n: int):
    """Three distinct indices in [0, n); re-draw (attempt counter) on duplicates."""
    idx = []
    for k in range(3):
        attempt = 0
        while True:
            r = hash32(seed * 0x9E3779B9 + it * 0x85EBCA6B + k * 0xC2B2AE35 + attempt * 0x27D4EB2F) % n
            if r not in idx:
                idx.append(r)
                break
            attempt += 1
    return idx


def _project_jac(K, q, t, X, uv):
    """Residual (2,) and Jacobian (2,6) wrt a left-multiplied small rotation and t."""
    R = quat_to_rot(q)
    Y = R @ X
    Xc = Y + t
    p = K @ Xc
    u, v = p[0] / p[2], p[1] / p[2]
    du = (K[0] - u * K[2]) / p[2]
    dv = (K[1] - v * K[2]) / p[2]
    # d Xc / d theta = -[Y]x
    S = np.array([[0, -Y[2], Y[1]], [Y[2], 0, -Y[0]], [-Y[1], Y[0], 0]])
    J = np.zeros((2, 6))
    J[0, :3] = -(du @ S)
    J[1, :3] = -(dv @ S)
    J[0, 3:] = du
    J[1, 3:] = dv
    return np.array([u - uv[0], v - uv[1]]), J


def _apply_delta(q, t, d):
    dq = np.array([d[0] / 2, d[1] / 2, d[2] / 2, 1.0])
    qn = quat_mul(dq, q)
    qn = qn / np.sqrt(qn @ qn)
    return qn, t + d[3:]


def minimal_solve(K, X3, uv3, q0, t0, max_iter=10):
    """Newton iterations on the 6x6 system of a 3-point sample, from (q0, t0) -- pnp_ransac starts every sample at the identity."""
    q, t = q0.copy(), t0.copy()
    ok = False
    for _ in range(max_iter):
        f = np.zeros(6)
        J = np.zeros((6, 6))
        for i in range(3):
            r, Ji = _project_jac(K, q, t, X3[i], uv3[i])
            f[2 * i:2 * i + 2] = r
            J[2 * i:2 * i + 2] = Ji
        if not np.all(np.isfinite(f)) or not np.all(np.isfinite(J)):
            return False, q, t
        if np.max(np.abs(f)) < 1e-9:
            ok = True
            break
        try:
            d = np.linalg.solve(J, -f)
        except np.linalg.LinAlgError:
            return False, q, t
        if not np.all(np.isfinite(d)) or np.max(np.abs(d)) > 1e3:
            return False, q, t
        q, t = _apply_delta(q, t, d)
    if not ok:
        f = np.concatenate([_project_jac(K, q, t, X3[i], uv3[i])[0] for i in range(3)])
        ok = bool(np.all(np.isfinite(f)) and np.max(np.abs(f)) < 1e-6)
    return ok, q, t


def reproj_inliers(K, q, t, X, uv, thr):
    R = quat_to_rot(q)
    Xc = X @ R.T + t
    p = Xc @ K.T
    with np.errstate(divide="ignore", invalid="ignore"):
        du = p[:, 0] / p[:, 2] - uv[:, 0]
        dv = p[:, 1] / p[:, 2] - uv[:, 1]
        e2 = du * du + dv * dv
    return (p[:, 2] > 0) & (e2 <= thr * thr)


def gn_refit(K, q, t, X, uv, max_iter=10):
    for _ in range(max_iter):
        A = np.zeros((6, 6))
        g = np.zeros(6)
        for i in range(len(X)):
            r, J = _project_jac(K, q, t, X[i], uv[i])
            A += J.T @ J
            g += J.T @ r
        try:
            d = np.linalg.solve(A, -g)
        except np.linalg.LinAlgError:
            break
        if not np.all(np.isfinite(d)):
            break
        q, t = _apply_delta(q, t, d)
        if np.max(np.abs(d)) < 1e-10:
            break
    return q, t


def pnp_ransac(K, xyz, xy, rvec0, tvec0, iterations=500, reproj_error=2.0, seed=0):
    """Returns ok, rvec, tvec, inliers (ascending int32)."""
    K = np.asarray(K, np.float64).reshape(3, 3)
    X = np.asarray(xyz, np.float32).astype(np.float64).reshape(-1, 3)
    uv = np.asarray(xy, np.float32).astype(np.float64).reshape(-1, 2)
    n = len(X)
    t0 = np.asarray(tvec0, np.float64).reshape(3).copy()
    if n < 4:
        return False, np.asarray(rvec0, float).reshape(3), t0, np.zeros(0, np.int32)
    # The minimal solver is PRIOR-FREE, as cv::solvePnPRansac's is (base.cpp:237-239: a closed-form P3P per sample; `useExtrinsicGuess` only
    # seeds the model that is handed back when nothing better is found): every sample's Newton iteration starts at the identity -- "no
    # motion", the P3P root a frame-to-frame step wants -- so hypotheses, scores, selection and refit of a frame do not depend on the
    # previous frame's pose (round 6; rounds 1-5 started at the motion prior, which chained every frame's RANSAC to the solve before it).
    # The prior (rvec0, tvec0) is what comes back when no model is found.
    q_id, t_id = np.array([0.0, 0.0, 0.0, 1.0]), np.zeros(3)
    best = (-1, None, None, None)
    for it in range(iterations):
        s = sample_triplet(seed, it, n)
        ok, q, t = minimal_solve(K, X[s], uv[s], q_id, t_id)
        if not ok:
            continue
        mask = reproj_inliers(K, q, t, X, uv, reproj_error)
        c = int(mask.sum())
        if c > best[0]:
            best = (c, q, t, mask)
    if best[0] < 4:
        return False, np.asarray(rvec0, float).reshape(3), t0, np.zeros(0, np.int32)
    _, q, t, mask = best
    inl = np.nonzero(mask)[0].astype(np.int32)
    q, t = gn_refit(K, q, t, X[inl], uv[inl])
    return True, quat_to_rvec(q), t, inl


# ----------------------------------------------------------------------------
# rows Q, S: CostFunctor32 + Ceres-style LM  (base.cpp:282-375, cost.hpp:27-58)
# ----------------------------------------------------------------------------
def _drot(q):
    """dR/dq_k for Eigen's toRotationMatrix polynomial, k = x, y, z, w."""
    x, y, z, w = q
    dx = np.array([[0, 2 * y, 2 * z], [2 * y, -4 * x, -2 * w], [2 * z, 2 * w, -4 * x]])
    dy = np.array([[-4 * y, 2 * x, 2 * w], [2 * x, 0, 2 * z], [-2 * w, 2 * z, -4 * y]])
    dz = np.array([[-4 * z, -2 * w, 2 * x], [2 * w, -4 * z, 2 * y], [2 * x, 2 * y, 0]])
    dw = np.array([[0, -2 * z, 2 * y], [2 * z, 0, -2 * x], [-2 * y, 2 * x, 0]])
    return [dx, dy, dz, dw]


def _plus_jacobian(q):
    x, y, z, w = q
    return np.array([[w, z, -y], [-z, w, x], [y, -x, w], [-x, -y, -z]])


def quat_plus(q, d):
    """EigenQuaternionParameterization::Plus."""
    nd = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    if nd > 0:
        s = np.sin(nd) / nd
        dq = np.array([s * d[0], s * d[1], s * d[2], np.cos(nd)])
        return quat_mul(dq, q)
    return q.copy()


def residuals_and_jacobian(P_l, P_r, obs, q, t, want_jac=True):
    """obs: structured (X[n,3] f64, uv[n,2] f64, cam[n], inverse[n]).  cost.hpp:27-58.

    Returns r [n,2], J [n,2,6] (local: 3 rotation + 3 translation) or None.
    """
    X, uv, cam, inv = obs
    n = len(X)
    Ps = np.stack([np.asarray(P_l, np.float64).reshape(3, 4), np.asarray(P_r, np.float64).reshape(3, 4)])
    P = Ps[cam]                                             # [n,3,4]
    R = quat_to_rot(q)
    fwd = ~inv.astype(bool)
    Xt = np.where(fwd[:, None], X @ R.T + t, (X - t) @ R)   # R^T (X - t) == (X - t) @ R
    p = np.einsum("nij,nj->ni", P[:, :, :3], Xt) + P[:, :, 3]
    u = p[:, 0] / p[:, 2]
    v = p[:, 1] / p[:, 2]
    r = np.stack([u - uv[:, 0], v - uv[:, 1]], axis=1)
    if not want_jac:
        return r, None
    du = (P[:, 0, :3] - u[:, None] * P[:, 2, :3]) / p[:, 2:3]   # [n,3]
    dv = (P[:, 1, :3] - v[:, None] * P[:, 2, :3]) / p[:, 2:3]
    dR = _drot(q)
    G = _plus_jacobian(q)
    Jq = np.zeros((n, 2, 4))
    wv = np.where(fwd[:, None], X, X - t)
    for k in range(4):
        dXt = np.where(fwd[:, None], wv @ dR[k].T, wv @ dR[k])  # (dR) w   or (dR)^T w
        Jq[:, 0, k] = np.einsum("ni,ni->n", du, dXt)
        Jq[:, 1, k] = np.einsum("ni,ni->n", dv, dXt)
    J = np.zeros((n, 2, 6))
    J[:, :, :3] = Jq @ G
    dt_f_u, dt_f_v = du, dv
    dt_i_u, dt_i_v = -(du @ R.T), -(dv @ R.T)                    # d/dt of R^T (X - t) = -R^T
    J[:, 0, 3:] = np.where(fwd[:, None], dt_f_u, dt_i_u)
    J[:, 1, 3:] = np.where(fwd[:, None], dt_f_v, dt_i_v)
    return r, J


def _huber(s, delta):
    """rho(s), rho'(s) for ceres::HuberLoss(delta); s = squared norm."""
    b = delta * delta
    big = s > b
    sq = np.sqrt(np.where(big, s, 1.0))
    rho = np.where(big, 2 * delta * sq - b, s)
    rho1 = np.where(big, delta / sq, 1.0)
    return rho, rho1


@dataclass
class RefineSummary:
    iterations: int = 0
    converged: bool = False
    usable: bool = False
    initial_cost: float = 0.0
    final_cost: float = 0.0
    message: str = ""


def pnp_refine(P_l, P_r, obs, q0, t0, max_iterations=40, huber_delta=1.0):
    """Ceres TrustRegionMinimizer + LevenbergMarquardtStrategy restatement.  Returns q, t, summary."""
    q = np.asarray(q0, np.float64).copy()
    t = np.asarray(t0, np.float64).copy()
    summ = RefineSummary()
    if len(obs[0]) == 0:
        summ.converged = summ.usable = True
        return q, t, summ

    def evaluate(q, t, want_jac):
        r, J = residuals_and_jacobian(P_l, P_r, obs, q, t, want_jac)
        s = r[:, 0] ** 2 + r[:, 1] ** 2
        rho, rho1 = _huber(s, huber_delta)
        cost = 0.5 * float(np.sum(rho))
        if not want_jac:
            return cost, None, None
        w = np.sqrt(rho1)
        rw = (r * w[:, None]).reshape(-1)
        Jw = (J * w[:, None, None]).reshape(-1, 6)
        return cost, Jw.T @ Jw, Jw.T @ rw

    cost, A, g = evaluate(q, t, True)
    summ.initial_cost = summ.final_cost = cost
    if not np.isfinite(cost):
        return q, t, summ
    scale = 1.0 / (1.0 + np.sqrt(np.diag(A)))               # Jacobi scaling, fixed at iteration 0
    radius = 1e4
    decrease_factor = 2.0
    invalid = 0
    x_norm = np.sqrt(q @ q + t @ t)
    summ.usable = True
    if np.max(np.abs(g)) <= 1e-10:
        summ.converged = True
        summ.message = "gradient tolerance"
        return q, t, summ
    it = 0
    while True:
        if it >= max_iterations:
            summ.message = "max iterations"
            break
        it += 1
        summ.iterations = it
        As = A * scale[:, None] * scale[None, :]
        gs = g * scale
        D2 = np.clip(np.diag(As), 1e-6, 1e32) / radius
        try:
            L = np.linalg.cholesky(As + np.diag(D2))
            ds = -np.linalg.solve(L.T, np.linalg.solve(L, gs))
        except np.linalg.LinAlgError:
            ds = np.full(6, np.nan)
        model_change = -(gs @ ds + 0.5 * ds @ (As @ ds))
        if not np.all(np.isfinite(ds)) or not (model_change > 0):
            invalid += 1
            if invalid >= 5:
                summ.usable = False
                summ.message = "too many invalid steps"
                break
            radius /= decrease_factor
            decrease_factor *= 2
            continue
        invalid = 0
        d = ds * scale
        qc = quat_plus(q, d[:3])
        tc = t + d[3:]
        cand_cost, _, _ = evaluate(qc, tc, False)
        step_norm = np.sqrt(np.sum((qc - q) ** 2) + np.sum((tc - t) ** 2))
        if step_norm <= 1e-8 * (x_norm + 1e-8):
            summ.converged = True
            summ.message = "parameter tolerance"
            break
        cost_change = cost - cand_cost
        if abs(cost_change) <= 1e-6 * cost:
            summ.converged = True
            summ.message = "function tolerance"
            break
        rel = cost_change / model_change
        if np.isfinite(cand_cost) and rel > 1e-3:
            q, t = qc, tc
            cost, A, g = evaluate(q, t, True)
            x_norm = np.sqrt(q @ q + t @ t)
            radius = min(1e16, radius / max(1.0 / 3.0, 1.0 - (2.0 * rel - 1.0) ** 3))
            decrease_factor = 2.0
            summ.final_cost = cost
            if np.max(np.abs(g)) <= 1e-10:
                summ.converged = True
                summ.message = "gradient tolerance"
                break
            if radius < 1e-32:
                summ.converged = True
                summ.message = "min trust region radius"
                break
        else:
            radius /= decrease_factor
            decrease_factor *= 2
    return q, t, summ


# ----------------------------------------------------------------------------
# the whole of FeatureFrontEnd's odometry state machine (rows A, K-bookkeeping, L-R)
# ----------------------------------------------------------------------------
@dataclass
class FrontEndState:
    """FeatureFrontEnd members used by matchDescriptors/solveStereoOdometry (hpp:123-177)."""
    keypoints: List[np.ndarray] = field(default_factory=list)     # deque of [n,2] float32, last 4
    descriptors: List[np.ndarray] = field(default_factory=list)
    maps: List[Optional[np.ndarray]] = field(default_factory=lambda: [np.zeros(0, np.int32)] * 3)
    matches: List[Optional[np.ndarray]] = field(default_factory=lambda: [None, None, None])
    P_l: Optional[np.ndarray] = None
    P_r: Optional[np.ndarray] = None
    r_pred: np.ndarray = field(default_factory=lambda: np.zeros(3))
    t_pred: np.ndarray = field(default_factory=lambda: np.zeros(3))
    frame_count: int = 0
    prev_pts3d: Optional[np.ndarray] = None
    prev_matched_to_valid: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    inliers_pnp: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))
    inliers_postmatching: np.ndarray = field(default_factory=lambda: np.zeros(0, np.int32))


CURR_LEFT_CURR_RIGHT, CURR_LEFT_PREV_LEFT, PREV_LEFT_PREV_RIGHT = 0, 1, 2
_POS = {0: (-2, -1), 1: (-2, -4), 2: (-4, -3)}     # hpp:87-90


def add_features(st: FrontEndState, xy_l, desc_l, xy_r, desc_r, P_l, P_r):
    """Deque bookkeeping of addStereoImagePair (nn.cpp:465-498)."""
    st.P_l, st.P_r = np.array(P_l, float).reshape(3, 4), np.array(P_r, float).reshape(3, 4)
    st.keypoints += [np.asarray(xy_l, np.float32), np.asarray(xy_r, np.float32)]
    st.descriptors += [np.asarray(desc_l, np.float32), np.asarray(desc_r, np.float32)]
    while len(st.keypoints) > 4:
        st.keypoints.pop(0)
        st.descriptors.pop(0)


def match_descriptors(st: FrontEndState, match_type: int, selector="KNN", cross_check=False, ratio=0.8):
    """base.cpp:434-491."""
    from .matching import bf_match
    a, b = _POS[match_type]
    idx, dist = bf_match(st.descriptors[a], st.descriptors[b], selector, cross_check, ratio)
    if match_type == CURR_LEFT_CURR_RIGHT:
        st.maps[PREV_LEFT_PREV_RIGHT] = st.maps[CURR_LEFT_CURR_RIGHT]        # base.cpp:475-481
    st.maps[match_type] = idx
    st.matches[match_type] = (idx, dist)
    return idx, dist


def join(st: FrontEndState, stereo_threshold: float, min_disparity: float, refinement_degree: int):
    """The 4-way correspondence join, base.cpp:127-207.  Returns dict of aligned lists."""
    kp = st.keypoints
    cl, cr, pl, pr = kp[-2], kp[-1], kp[-4], kp[-3]
    m_stereo, m_temporal, m_prev = st.maps[0], st.maps[1], st.maps[2]
    out = dict(cl=[], cr=[], pl=[], pr=[], post=[], valid_to_prev=[])
    cur_matched_to_valid = np.full(len(cl), -1, np.int32)
    for qi in range(len(cl)):                        # cv_Dmatches are in query order
        ti = m_stereo[qi]
        if ti < 0:
            continue
        if m_temporal[qi] == -1:
            continue
        a, b = cl[qi], cr[ti]
        if abs(np.float32(a[1]) - np.float32(b[1])) > np.float32(stereo_threshold) or \
                abs(np.float32(a[0]) - np.float32(b[0])) < np.float32(min_disparity):
            continue
        pi = m_temporal[qi]
        if m_prev[pi] == -1:
            continue
        out["cl"].append(a)
        out["cr"].append(b)
        out["post"].append(qi)
        out["pl"].append(pl[pi])
        out["pr"].append(pr[m_prev[pi]])
        if refinement_degree >= 3:
            cur_matched_to_valid[qi] = len(out["cl"]) - 1
            out["valid_to_prev"].append(pi)
    for k in ("cl", "cr", "pl", "pr"):
        out[k] = np.asarray(out[k], np.float32).reshape(-1, 2)
    out["post"] = np.asarray(out["post"], np.int32)
    out["valid_to_prev"] = np.asarray(out["valid_to_prev"], np.int32)
    out["cur_matched_to_valid"] = cur_matched_to_valid
    return out


def build_observations(j, pts3d, inliers, st: FrontEndState, refinement_degree: int):
    """Residual blocks in the order base.cpp:291-356 adds them."""
    X, uv, cam, inv = [], [], [], []
    for vi in inliers.tolist():
        X.append(pts3d[vi]); uv.append(j["pl"][vi]); cam.append(0); inv.append(0)
        if refinement_degree <= 1:
            continue
        X.append(pts3d[vi]); uv.append(j["pr"][vi]); cam.append(1); inv.append(0)
        if refinement_degree <= 2:
            continue
        if st.prev_pts3d is None:
            continue
        pm = j["valid_to_prev"][vi]
        pv = st.prev_matched_to_valid[pm]
        if pv == -1:
            continue
        X.append(st.prev_pts3d[pv]); uv.append(j["cl"][vi]); cam.append(0); inv.append(1)
        if refinement_degree <= 3:
            continue
        X.append(st.prev_pts3d[pv]); uv.append(j["cr"][vi]); cam.append(1); inv.append(1)
    return (np.asarray(X, np.float32).astype(np.float64).reshape(-1, 3),
            np.asarray(uv, np.float32).astype(np.float64).reshape(-1, 2),
            np.asarray(cam, np.int64), np.asarray(inv, np.int64))


def solve_stereo_odometry(st: FrontEndState, stereo_threshold=2.0, min_disparity=0.25,
                          refinement_degree=4, seed=0):
    """base.cpp:125-399.  Returns (q_xyzw, t) of cam0_curr_T_cam0_prev plus a debug dict."""
    j = join(st, stereo_threshold, min_disparity, refinement_degree)
    st.inliers_postmatching = j["post"]
    pts3d = triangulate(st.P_l, st.P_r, j["cl"], j["cr"])
    K = st.P_l[:, :3].copy()
    ok, rvec, tvec, inl = pnp_ransac(K, pts3d, j["pl"], st.r_pred, st.t_pred, 500, 2.0, seed)
    st.inliers_pnp = inl
    acc = np.linalg.norm(tvec - st.t_pred) / TIME_INTERVAL          # base.cpp:241-242
    do_opt = False
    if not ok:
        rvec, tvec = st.r_pred.copy(), st.t_pred.copy()
    elif st.frame_count > IGNORE_FRAME_COUNT and acc > MAX_ACCELERATION:
        rvec, tvec = st.r_pred.copy(), st.t_pred.copy()
    else:
        st.r_pred, st.t_pred = rvec.copy(), tvec.copy()
        do_opt = True
    q = rvec_to_quat(rvec)
    t = np.asarray(tvec, float).copy()
    summ = None
    if do_opt and refinement_degree > 0:
        obs = build_observations(j, pts3d, inl, st, refinement_degree)
        q2, t2, summ = pnp_refine(st.P_l, st.P_r, obs, q, t)
        if summ.usable and summ.converged:                          # base.cpp:366-374
            q, t = q2, t2
    # cam0_curr_T_cam0_prev = (q, t)^-1                              base.cpp:377-385
    R = quat_to_rot(q / np.sqrt(q @ q))
    q_inv = np.array([-q[0], -q[1], -q[2], q[3]]) / np.sqrt(q @ q)
    t_inv = -(R.T @ t)
    if refinement_degree >= 3:                                       # base.cpp:388-394
        st.prev_matched_to_valid = j["cur_matched_to_valid"]
        st.prev_pts3d = pts3d
    st.frame_count += 1
    return q_inv, t_inv, dict(join=j, pts3d=pts3d, ok=ok, rvec=rvec, tvec=tvec, inliers=inl,
                              q_prev_T_curr=q, t_prev_T_curr=t, summary=summ, do_opt=do_opt)
