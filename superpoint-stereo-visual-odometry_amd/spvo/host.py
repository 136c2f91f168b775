"""ctypes binding of host/harness_capi.cpp: drives the C++ SuperPointFeatureFrontEnd
(the mirror of the reference's class) with the call sequence of visual_odometry_node.cpp."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.environ.get("SPVO_LIB_DIR") or os.path.dirname(_HERE), "libspvo_host.so")   # SPVO_LIB_DIR: a side build (make BUILD=... OUT=variants/x) for A/B measurements
_lib = None

CURR_LEFT_CURR_RIGHT, CURR_LEFT_PREV_LEFT, PREV_LEFT_PREV_RIGHT = 0, 1, 2
PREV_LEFT, PREV_RIGHT, CURR_LEFT, CURR_RIGHT = -4, -3, -2, -1


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(f"{LIB_PATH} is missing: run `python __graft_entry__.py` first")
        lib = C.CDLL(LIB_PATH)
        vp = C.c_void_p
        lib.spvo_host_create.restype = vp
        lib.spvo_host_create.argtypes = [C.c_char_p, C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int,
                                         C.c_float, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int, C.c_int]
        lib.spvo_host_destroy.argtypes = [vp]
        lib.spvo_host_set_options.argtypes = [C.c_int, C.c_int, C.c_int]
        lib.spvo_host_set_options.restype = None
        lib.spvo_host_max_keypoints.argtypes = [vp]
        lib.spvo_host_set_deferred_copies.argtypes = [vp, C.c_int]
        lib.spvo_host_set_deferred_copies.restype = None
        lib.spvo_host_destroy.restype = None
        lib.spvo_host_last_error.argtypes = [vp]
        lib.spvo_host_last_error.restype = C.c_char_p
        for name in ("spvo_host_engine_loaded", "spvo_host_dq_size", "spvo_host_frame_count", "spvo_host_last_solve"):
            getattr(lib, name).argtypes = [vp]
        lib.spvo_host_set_seed.argtypes = [vp, C.c_uint]
        lib.spvo_host_set_seed.restype = None
        lib.spvo_host_add_stereo_pair.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp]
        lib.spvo_host_add_stereo_pair.restype = None
        lib.spvo_host_add_stereo_pair_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_int]
        lib.spvo_host_add_stereo_pair_dev.restype = None
        lib.spvo_host_prefetch_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t]
        lib.spvo_host_prefetch_dev.restype = None
        lib.spvo_host_run_device_block.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_int, C.c_size_t, vp, vp, C.c_long, C.c_int, C.c_int, C.c_int, vp]
        lib.spvo_host_make_image.argtypes = [vp, C.c_int, C.c_int]
        lib.spvo_host_make_image.restype = vp
        lib.spvo_host_free_image.argtypes = [vp]
        lib.spvo_host_free_image.restype = None
        lib.spvo_host_add_stereo_pair_mat.argtypes = [vp, vp, vp, vp, vp]
        lib.spvo_host_add_stereo_pair_mat.restype = None
        lib.spvo_host_prefetch_mat.argtypes = [vp, vp, vp]
        lib.spvo_host_prefetch_mat.restype = None
        lib.spvo_host_ctx.argtypes = [vp]
        lib.spvo_host_ctx.restype = vp
        lib.spvo_host_match.argtypes = [vp, C.c_int]
        lib.spvo_host_match.restype = None
        lib.spvo_host_solve.argtypes = [vp, vp, vp]
        lib.spvo_host_solve.restype = None
        lib.spvo_host_solve_submit.argtypes = [vp]
        lib.spvo_host_solve_collect.argtypes = [vp, vp, vp]
        lib.spvo_host_solve_pending.argtypes = [vp]
        lib.spvo_host_clear.argtypes = [vp]
        lib.spvo_host_clear.restype = None
        lib.spvo_host_keypoints.argtypes = [vp, C.c_int, vp, C.c_int]
        lib.spvo_host_descriptors.argtypes = [vp, C.c_int, vp, C.c_int]
        lib.spvo_host_image.argtypes = [vp, C.c_int, vp, C.c_int]
        lib.spvo_host_descriptors_raw.argtypes = [vp, C.c_int, vp, C.c_int]
        lib.spvo_host_image_raw.argtypes = [vp, C.c_int, vp, C.c_int]
        lib.spvo_host_matches.argtypes = [vp, C.c_int, vp, vp, vp, C.c_int]
        lib.spvo_host_map.argtypes = [vp, C.c_int, vp, C.c_int]
        lib.spvo_host_inliers.argtypes = [vp, C.c_int, vp, C.c_int]
        lib.spvo_host_write_kitti.argtypes = [C.c_char_p, C.c_int, C.c_int, vp, vp, vp, vp, C.c_int, vp]
        lib.spvo_host_write_latency.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p, C.c_int, vp, C.c_int,
                                                C.c_char_p, C.c_int]
        _lib = lib
    return _lib


def write_kitti_poses(directory, kitti_eval_id, rel_poses, base_T_cam0=((0, 0, 0, 1), (0, 0, 0)), seq_start=0):
    """Integrate front-end outputs (q xyzw, t of cam0_curr_T_cam0_prev) and write <id>_pred.txt.
    Returns (lines written, final world_T_base as (q, t))."""
    lib = load()
    q = np.ascontiguousarray([p[0] for p in rel_poses], np.float64).reshape(-1, 4)
    t = np.ascontiguousarray([p[1] for p in rel_poses], np.float64).reshape(-1, 3)
    bq = np.ascontiguousarray(base_T_cam0[0], np.float64)
    bt = np.ascontiguousarray(base_T_cam0[1], np.float64)
    final = np.zeros(7)
    n = lib.spvo_host_write_kitti(str(directory).encode(), kitti_eval_id, seq_start, _p(bq), _p(bt), _p(q), _p(t), len(q), _p(final))
    return n, (final[:4].copy(), final[4:].copy())


def write_latency_csv(directory, prefix, batch, height, width, precision, kitti_eval_id, rows):
    lib = load()
    r = np.ascontiguousarray(rows, np.float32).reshape(-1, 4)
    name = C.create_string_buffer(256)
    n = lib.spvo_host_write_latency(str(directory).encode(), prefix.encode(), batch, height, width, precision.encode(), kitti_eval_id,
                                    _p(r), len(r), name, 256)
    return n, name.value.decode()


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class FrameRecord(C.Structure):
    """host/harness_capi.cpp: SpvoFrameRecord -- what one frame of spvo_host_run_device_block leaves behind"""
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("latency_ms", C.c_double), ("has_pose", C.c_int), ("pnp_ok", C.c_int), ("accepted", C.c_int),
                ("refined", C.c_int), ("lm_iterations", C.c_int), ("pnp_inliers", C.c_int), ("stereo_matches", C.c_int), ("keypoints_left", C.c_int)]

RECORD_DTYPE = np.dtype([("q", np.float64, 4), ("t", np.float64, 3), ("latency_ms", np.float64), ("has_pose", np.int32), ("pnp_ok", np.int32), ("accepted", np.int32),
                         ("refined", np.int32), ("lm_iterations", np.int32), ("pnp_inliers", np.int32), ("stereo_matches", np.int32), ("keypoints_left", np.int32)])
assert RECORD_DTYPE.itemsize == C.sizeof(FrameRecord)


def set_options(device=-1, max_keypoints=-1, match_fp8=-1):
    """FeatureFrontEnd::setDevice / SuperPointFeatureFrontEnd::setMaxKeypoints / ::setMatchFp8 (host/feature_detection.hpp): options of
    the front ends constructed AFTERWARDS; a negative value hands the decision back to the environment variable (SPVO_DEVICE,
    SPVO_MAX_KEYPOINTS, SPVO_MATCH_FP8)."""
    load().spvo_host_set_options(int(device), int(max_keypoints), int(match_fp8))


class FrontEnd:
    """SuperPointFeatureFrontEnd with the reference launch-file defaults
    (launch/visual_odometry_superpoint.launch:3-26)."""

    def __init__(self, models_dir, prefix="superpoint_pretrained", machine="laptop", selector="KNN", cross_check=True,
                 batch=2, height=360, width=1176, conf_thresh=0.015, dist_thresh=4, border_remove=4,
                 stereo_threshold=2.0, min_disparity=0.25, refinement_degree=4, verbose=False, precision="FP32"):
        self.lib = load()
        self.h = self.lib.spvo_host_create(models_dir.encode(), prefix.encode(), machine.encode(),
                                           1 if selector == "KNN" else 0, int(cross_check), batch, height, width,
                                           conf_thresh, dist_thresh, border_remove, stereo_threshold, min_disparity,
                                           refinement_degree, int(verbose), {"FP32": 0, "FP16": 1, "INT8": 2}[precision])
        self.H, self.W = height, width
        self.cap = max(1000, int(self.lib.spvo_host_max_keypoints(self.h)))

    def close(self):
        if self.h:
            self.lib.spvo_host_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def engine_loaded(self):
        return bool(self.lib.spvo_host_engine_loaded(self.h))

    @property
    def last_error(self):
        return self.lib.spvo_host_last_error(self.h).decode()

    def set_seed(self, seed):
        self.lib.spvo_host_set_seed(self.h, seed)

    def add_stereo_image_pair(self, img_l, img_r, P_l, P_r):
        img_l = np.ascontiguousarray(img_l, np.uint8)
        img_r = np.ascontiguousarray(img_r, np.uint8)
        Pl = np.ascontiguousarray(P_l, np.float64)
        Pr = np.ascontiguousarray(P_r, np.float64)
        self.lib.spvo_host_add_stereo_pair(self.h, _p(img_l), _p(img_r), img_l.shape[0], img_l.shape[1], _p(Pl), _p(Pr))

    def add_stereo_image_pair_device(self, d_l: int, d_r: int, rows: int, cols: int, stride: int, P_l, P_r, host_descriptors=False):
        Pl = np.ascontiguousarray(P_l, np.float64)
        Pr = np.ascontiguousarray(P_r, np.float64)
        self.lib.spvo_host_add_stereo_pair_dev(self.h, C.c_void_p(d_l), C.c_void_p(d_r), rows, cols, stride, _p(Pl), _p(Pr),
                                               int(host_descriptors))

    def set_deferred_copies(self, on):
        """SuperPointFeatureFrontEnd::setDeferredHostCopies: False (the default, the reference's behaviour) = images_dq / descriptors_dq are
        filled inside addStereoImagePair; True = the two bulk copies are made while the solver's kernels run"""
        self.lib.spvo_host_set_deferred_copies(self.h, int(on))

    def context(self):
        """The spvo_ctx of this front end wrapped for the profiling calls of spvo.capi."""
        from . import capi
        c = capi.Context.__new__(capi.Context)
        c.lib = capi.load()
        c.h = C.c_void_p(self.lib.spvo_host_ctx(self.h))
        c.close = lambda: None                         # owned by the C++ object
        return c

    def run_device_block(self, d_l, d_r, rows, cols, stride, P_l, P_r, first, n, depth=4, deferred=True):
        """n stereoCallbacks on device-resident pairs in ONE call (host/harness_capi.cpp: spvo_host_run_device_block): frame k of the block is
        pair (first + k) % len(d_l); `depth` pairs announced ahead, the solve deferred by a frame (the block's last pose is collected before
        the call returns).  d_l / d_r: device pointers of the cycle.  Returns a structured array of n records (RECORD_DTYPE): pose,
        first-call -> pose latency in ms, solver outcome, keypoint and match counts."""
        cyc = len(d_l)
        pl = (C.c_void_p * cyc)(*d_l)
        pr = (C.c_void_p * cyc)(*d_r)
        rec = np.zeros(n, RECORD_DTYPE)
        Pl, Pr = np.ascontiguousarray(P_l, np.float64), np.ascontiguousarray(P_r, np.float64)
        got = self.lib.spvo_host_run_device_block(self.h, pl, pr, cyc, rows, cols, stride, _p(Pl), _p(Pr), first, n, depth, int(deferred), _p(rec))
        if got != n:
            raise RuntimeError("spvo_host_run_device_block failed: " + self.last_error)
        return rec

    def prefetch_device(self, d_l: int, d_r: int, rows: int, cols: int, stride: int):
        self.lib.spvo_host_prefetch_dev(self.h, C.c_void_p(d_l), C.c_void_p(d_r), rows, cols, stride)

    def step_device(self, d_l, d_r, rows, cols, stride, P_l, P_r, next_pair=None, next2_pair=None, deferred_solve=False, next3_pair=None, next4_pair=None):
        """One stereoCallback on a device-resident pair; `next_pair` / `next2_pair` = (d_l, d_r) of the
        following two frames, handed over early: their detector runs while this frame is matched and
        solved, and the post-processing of one overlaps with the network of the other.
        deferred_solve: this frame's solve is only handed over (solveStereoOdometrySubmit); its pose is what the NEXT call -- or
        finish_solve() -- returns, so the solver never keeps the host from handing the next images over."""
        # the pairs ahead are announced BEFORE this pair is collected (a node learns of them when they arrive, not when it has time):
        # their submissions -- and, with trunk pairing, the launch of a completed group -- do not wait for this pair's tail
        if next_pair is not None:
            self.prefetch_device(d_l, d_r, rows, cols, stride)            # (this pair first, if it has not been announced: the queue is in call order)
        for nxt in (next_pair, next2_pair, next3_pair, next4_pair):
            if nxt is not None:
                self.prefetch_device(nxt[0], nxt[1], rows, cols, stride)   # no-op if already announced
        self.add_stereo_image_pair_device(d_l, d_r, rows, cols, stride, P_l, P_r)
        return self._match_and_solve(deferred_solve)

    def _match_and_solve(self, deferred_solve):
        if self.dq_size() < 4:
            self.match_descriptors(CURR_LEFT_CURR_RIGHT)
            return None
        self.match_descriptors(CURR_LEFT_CURR_RIGHT)
        self.match_descriptors(CURR_LEFT_PREV_LEFT)
        if not deferred_solve:
            return self.solve_stereo_odometry()
        # this frame's chain first (it needs nothing of the previous frame's result), then the previous frame's pose
        submitted = bool(self.lib.spvo_host_solve_submit(self.h))
        if self.lib.spvo_host_solve_pending(self.h) > (1 if submitted else 0):
            return self.finish_solve()
        return None

    def finish_solve(self):
        """Collects a deferred solve, if one is pending: (q, t) or None."""
        if not self.lib.spvo_host_solve_pending(self.h):
            return None
        q = np.zeros(4)
        t = np.zeros(3)
        return (q, t) if self.lib.spvo_host_solve_collect(self.h, _p(q), _p(t)) else None

    # ---- host images as cv::Mat objects (what cv_bridge hands to the node): the reference's own entry point
    def make_image(self, img) -> int:
        img = np.ascontiguousarray(img, np.uint8)
        return self.lib.spvo_host_make_image(_p(img), img.shape[0], img.shape[1])

    def free_image(self, handle: int):
        self.lib.spvo_host_free_image(C.c_void_p(handle))

    def step_host(self, mat_l: int, mat_r: int, P_l, P_r, next_pair=None, next2_pair=None, deferred_solve=False, next3_pair=None, next4_pair=None):
        """One stereoCallback through addStereoImagePair(cv::Mat&, ...) (node.cpp:175) on HOST images; `next_pair` /
        `next2_pair` = (mat_l, mat_r) handles of the following frames, announced with prefetchStereoImagePair."""
        Pl = np.ascontiguousarray(P_l, np.float64)
        Pr = np.ascontiguousarray(P_r, np.float64)
        if next_pair is not None:
            self.lib.spvo_host_prefetch_mat(self.h, C.c_void_p(mat_l), C.c_void_p(mat_r))
        for nxt in (next_pair, next2_pair, next3_pair, next4_pair):     # (announced before this pair is collected: see step_device)
            if nxt is not None:
                self.lib.spvo_host_prefetch_mat(self.h, C.c_void_p(nxt[0]), C.c_void_p(nxt[1]))
        self.lib.spvo_host_add_stereo_pair_mat(self.h, C.c_void_p(mat_l), C.c_void_p(mat_r), _p(Pl), _p(Pr))
        return self._match_and_solve(deferred_solve)

    def match_descriptors(self, match_type):
        self.lib.spvo_host_match(self.h, match_type)

    def solve_stereo_odometry(self):
        q = np.zeros(4)
        t = np.zeros(3)
        self.lib.spvo_host_solve(self.h, _p(q), _p(t))
        return q, t

    def clear(self):
        self.lib.spvo_host_clear(self.h)

    def dq_size(self):
        return self.lib.spvo_host_dq_size(self.h)

    def frame_count(self):
        return self.lib.spvo_host_frame_count(self.h)

    def last_solve(self):
        """Outcome of the last solveStereoOdometry: dict(pnp_ok, accepted, refined, lm_iterations)."""
        v = self.lib.spvo_host_last_solve(self.h)
        return dict(pnp_ok=bool(v & 1), accepted=bool(v & 2), refined=bool(v & 4), lm_iterations=v >> 8)

    def keypoints(self, position):
        xy = np.zeros((self.cap, 2), np.float32)
        n = self.lib.spvo_host_keypoints(self.h, position, _p(xy), self.cap)
        return xy[:max(n, 0)].copy()

    def descriptors(self, position):
        d = np.zeros((self.cap, 256), np.float32)
        n = self.lib.spvo_host_descriptors(self.h, position, _p(d), self.cap)
        return d[:max(n, 0)].copy()

    def image(self, position):
        out = np.zeros((self.H, self.W), np.uint8)
        n = self.lib.spvo_host_image(self.h, position, _p(out), out.size)
        return out if n == out.size else None

    def descriptors_raw(self, position):
        """descriptors_dq.end()[position] as the public member holds it NOW (no completeHostCopies first)"""
        d = np.zeros((self.cap, 256), np.float32)
        n = self.lib.spvo_host_descriptors_raw(self.h, position, _p(d), self.cap)
        return d[:max(n, 0)].copy()

    def image_raw(self, position):
        """images_dq.end()[position] as the public member holds it NOW (no completeHostCopies first)"""
        out = np.zeros((self.H, self.W), np.uint8)
        n = self.lib.spvo_host_image_raw(self.h, position, _p(out), out.size)
        return out if n == out.size else None

    def matches(self, match_type):
        q = np.zeros(self.cap, np.int32)
        t = np.zeros(self.cap, np.int32)
        d = np.zeros(self.cap, np.float32)
        n = self.lib.spvo_host_matches(self.h, match_type, _p(q), _p(t), _p(d), self.cap)
        return q[:n].copy(), t[:n].copy(), d[:n].copy()

    def map_of_indices(self, match_type):
        m = np.zeros(4096, np.int32)
        n = self.lib.spvo_host_map(self.h, match_type, _p(m), 4096)
        return m[:n].copy()

    def inliers(self, which="pnp"):
        v = np.zeros(4096, np.int32)
        n = self.lib.spvo_host_inliers(self.h, 0 if which == "pnp" else 1, _p(v), 4096)
        return v[:n].copy()

    def step(self, img_l, img_r, P_l, P_r):
        """One stereoCallback (node.cpp:150-262).  Returns (q, t) or None on the first frame."""
        self.add_stereo_image_pair(img_l, img_r, P_l, P_r)
        if self.dq_size() < 4:
            self.match_descriptors(CURR_LEFT_CURR_RIGHT)          # node.cpp:188-193
            return None
        self.match_descriptors(CURR_LEFT_CURR_RIGHT)              # node.cpp:196-198
        self.match_descriptors(CURR_LEFT_PREV_LEFT)
        return self.solve_stereo_odometry()                      # node.cpp:218


def classic_sequence(frames, P_l, P_r, selector="KNN", cross_check=True, stereo_threshold=2.0, refinement_degree=4, warm=0):
    """stereoCallback replayed on ClassicFeatureFrontEnd(ORB, ORB, BF, ...) (node.cpp:353-360) over host image pairs at their native
    resolution: returns (poses [n, 7] = q xyzw + t of cam0_curr_T_cam0_prev, stats [n, 4] = keypoints L, R, stereo matches, PnP
    inliers, seconds spent on frames warm .. n-1)."""
    lib = load()
    lib.spvo_host_classic_sequence.restype = C.c_int
    lib.spvo_host_classic_sequence.argtypes = [C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int,
                                               C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double)]
    n = len(frames)
    ls = [np.ascontiguousarray(f[0], np.uint8) for f in frames]
    rs = [np.ascontiguousarray(f[1], np.uint8) for f in frames]
    rows, cols = ls[0].shape
    pl = (C.c_void_p * n)(*[a.ctypes.data for a in ls])
    pr = (C.c_void_p * n)(*[a.ctypes.data for a in rs])
    Pl = np.ascontiguousarray(P_l, np.float64).reshape(12)
    Pr = np.ascontiguousarray(P_r, np.float64).reshape(12)
    poses = np.zeros((n, 7), np.float64)
    stats = np.zeros((n, 4), np.int32)
    sec = C.c_double(0)
    rc = lib.spvo_host_classic_sequence(n, pl, pr, rows, cols, Pl.ctypes.data, Pr.ctypes.data, 1 if selector == "KNN" else 0, int(cross_check), stereo_threshold,
                                        refinement_degree, warm, poses.ctypes.data, stats.ctypes.data, C.byref(sec))
    if rc != n:
        raise RuntimeError("classic front end failed at frame %d" % (-rc - 1))
    return poses, stats, sec.value
