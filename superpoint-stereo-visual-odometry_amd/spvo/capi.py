"""ctypes binding of the C ABI in include/spvo.h (used by tests/ and bench.py).

There is no CPU path: `load()` raises if libspvo.so has not been built
(`python __graft_entry__.py` or `make -C superpoint-stereo-visual-odometry_amd`).
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(os.environ.get("SPVO_LIB_DIR") or os.path.dirname(_HERE), "libspvo.so")   # SPVO_LIB_DIR: a side build (make BUILD=... OUT=variants/x) for A/B measurements


class Config(C.Structure):
    _fields_ = [("device", C.c_int), ("net_height", C.c_int), ("net_width", C.c_int),
                ("max_batch", C.c_int), ("conf_thresh", C.c_float), ("dist_thresh", C.c_int),
                ("border_remove", C.c_int), ("max_keypoints", C.c_int), ("bug_compat_p", C.c_int)]


class Features(C.Structure):
    _fields_ = [("n", C.c_int), ("xy", C.POINTER(C.c_float)), ("desc", C.POINTER(C.c_float))]


class DetectMirrors(C.Structure):
    _fields_ = [("n", C.c_int * 2), ("xy", C.POINTER(C.c_float) * 2), ("desc", C.POINTER(C.c_float) * 2), ("resized", C.POINTER(C.c_uint8) * 2), ("token", C.c_int)]


class RansacOpts(C.Structure):
    _fields_ = [("iterations", C.c_int), ("reproj_error", C.c_double), ("confidence", C.c_double),
                ("seed", C.c_uint32)]


class RefineOpts(C.Structure):
    _fields_ = [("max_iterations", C.c_int), ("huber_delta", C.c_double)]


class RefineSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("usable", C.c_int),
                ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class SolveInput(C.Structure):
    _fields_ = [("n", C.c_int), ("xy_cl", C.c_void_p), ("xy_cr", C.c_void_p), ("xy_pl", C.c_void_p), ("xy_pr", C.c_void_p),
                ("prev_xyz", C.c_void_p), ("prev_valid", C.c_void_p), ("P_l", C.c_double * 12), ("P_r", C.c_double * 12),
                ("rvec_pred", C.c_double * 3), ("tvec_pred", C.c_double * 3), ("frame_count", C.c_int),
                ("refinement_degree", C.c_int), ("ransac", RansacOpts), ("refine", RefineOpts), ("prev_index", C.c_void_p), ("late_prior", C.c_int)]


class SolveOutput(C.Structure):
    _fields_ = [("q", C.c_double * 4), ("t", C.c_double * 3), ("rvec", C.c_double * 3), ("tvec", C.c_double * 3),
                ("pnp_ok", C.c_int), ("accepted", C.c_int), ("refined", C.c_int), ("n_inliers", C.c_int),
                ("summary", RefineSummary)]


OBS_DTYPE = np.dtype([("X", np.float32, 3), ("uv", np.float32, 2), ("cam", np.int32), ("inverse", np.int32)])

# every symbol include/spvo.h declares
SYMBOLS = [
    "spvo_default_config", "spvo_create", "spvo_destroy", "spvo_last_error", "spvo_load_weights", "spvo_engine_precision", "spvo_set_fp32_split",
    "spvo_preprocess", "spvo_forward", "spvo_debug_tensor", "spvo_heatmap", "spvo_nms",
    "spvo_sample_descriptors", "spvo_detect", "spvo_detect_dev", "spvo_detect_dev_submit", "spvo_detect_wait", "spvo_set_trunk_pairing", "spvo_detect_submit", "spvo_detect_collect", "spvo_detect_collect_mirrors", "spvo_detect_mirrors_wait", "spvo_match", "spvo_match_slots", "spvo_set_prematch", "spvo_set_match_fp8", "spvo_get_match_fp8",
    "spvo_match_hamming", "spvo_orb_detect", "spvo_orb_tables", "spvo_triangulate", "spvo_pnp_ransac", "spvo_pnp_refine", "spvo_solve_stereo_odometry", "spvo_solve_submit", "spvo_solve_wait", "spvo_solve_wait_prior", "spvo_solve_pending", "spvo_stream", "spvo_synchronize",
    "spvo_profile_enable", "spvo_profile_reset", "spvo_profile_only", "spvo_profile_count", "spvo_profile_get", "spvo_profile_stage_kernel",
    "spvo_set_tuning", "spvo_get_tuning", "spvo_clear_tuning",
    "spvo_comm_unique_id", "spvo_comm_available", "spvo_comm_create", "spvo_comm_create_host", "spvo_comm_rank", "spvo_comm_world", "spvo_comm_destroy",
    "spvo_pose_allgather", "spvo_pose_allgather_n",
]

_lib = None


def load() -> C.CDLL:
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(f"{LIB_PATH} is missing: build the HIP library first "
                           "(python __graft_entry__.py); there is no CPU path")
    lib = C.CDLL(LIB_PATH)
    vp, ip, fp, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_float), C.POINTER(C.c_double)
    lib.spvo_default_config.argtypes = [C.POINTER(Config)]
    lib.spvo_default_config.restype = None
    lib.spvo_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.spvo_destroy.argtypes = [vp]
    lib.spvo_destroy.restype = None
    lib.spvo_last_error.argtypes = [vp]
    lib.spvo_last_error.restype = C.c_char_p
    lib.spvo_load_weights.argtypes = [vp, C.c_char_p]
    lib.spvo_preprocess.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, dp, vp]
    lib.spvo_forward.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.spvo_debug_tensor.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t]
    lib.spvo_heatmap.argtypes = [vp, vp, vp]
    lib.spvo_nms.argtypes = [vp, vp, vp, ip]
    lib.spvo_sample_descriptors.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.spvo_detect.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, dp, dp, C.c_int, C.c_int,
                                C.POINTER(Features), C.POINTER(Features), vp, vp]
    lib.spvo_detect_dev.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, dp, dp, C.c_int, C.c_int,
                                    C.POINTER(Features), C.POINTER(Features)]
    lib.spvo_detect_dev_submit.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int]
    lib.spvo_detect_wait.argtypes = [vp, dp, dp, C.POINTER(Features), C.POINTER(Features)]
    lib.spvo_detect_submit.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, C.c_int, C.c_int]
    lib.spvo_detect_collect.argtypes = [vp, dp, dp, C.POINTER(Features), C.POINTER(Features), vp, vp]
    lib.spvo_detect_collect_mirrors.argtypes = [vp, dp, dp, C.POINTER(DetectMirrors)]
    lib.spvo_detect_mirrors_wait.argtypes = [vp, C.POINTER(DetectMirrors)]
    lib.spvo_set_tuning.argtypes = [C.c_char_p, C.c_int]
    lib.spvo_get_tuning.argtypes = [C.c_char_p, C.c_int]
    lib.spvo_get_tuning.restype = C.c_int
    lib.spvo_clear_tuning.argtypes = []
    lib.spvo_clear_tuning.restype = None
    lib.spvo_match.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
    lib.spvo_match_slots.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
    lib.spvo_match_hamming.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
    lib.spvo_orb_detect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, C.c_int, vp, vp, C.c_int, ip]
    lib.spvo_orb_tables.argtypes = [vp, vp]
    lib.spvo_set_prematch.argtypes = [vp, C.c_int, C.c_int, C.c_int, C.c_float]
    lib.spvo_triangulate.argtypes = [vp, dp, dp, vp, vp, C.c_int, vp]
    lib.spvo_pnp_ransac.argtypes = [vp, dp, vp, vp, C.c_int, C.POINTER(RansacOpts), dp, dp, vp, ip, ip]
    lib.spvo_pnp_refine.argtypes = [vp, dp, dp, vp, C.c_int, C.POINTER(RefineOpts), dp, dp,
                                    C.POINTER(RefineSummary)]
    lib.spvo_solve_stereo_odometry.argtypes = [vp, C.POINTER(SolveInput), C.POINTER(SolveOutput), vp, vp]
    lib.spvo_solve_submit.argtypes = [vp, C.POINTER(SolveInput)]
    lib.spvo_solve_wait.argtypes = [vp, C.POINTER(SolveOutput), vp, vp]
    lib.spvo_solve_wait_prior.argtypes = [vp, vp, vp, C.c_int, C.POINTER(SolveOutput), vp, vp]
    lib.spvo_solve_pending.argtypes = [vp]
    lib.spvo_stream.argtypes = [vp]
    lib.spvo_stream.restype = vp
    lib.spvo_synchronize.argtypes = [vp]
    lib.spvo_profile_enable.argtypes = [vp, C.c_int]
    lib.spvo_profile_reset.argtypes = [vp]
    lib.spvo_profile_only.argtypes = [vp, C.c_char_p]
    lib.spvo_profile_count.argtypes = [vp]
    lib.spvo_profile_get.argtypes = [vp, C.c_int, C.c_char_p, C.c_size_t, dp, C.POINTER(C.c_longlong), dp, dp]
    lib.spvo_profile_stage_kernel.argtypes = [vp, C.c_char_p, C.c_char_p, C.c_size_t, dp]
    lib.spvo_comm_unique_id.argtypes = [vp]
    lib.spvo_comm_create.argtypes = [C.c_int, C.c_int, C.c_int, vp, C.POINTER(vp)]
    lib.spvo_comm_create_host.argtypes = [C.c_char_p, C.c_int, C.c_int, C.POINTER(vp)]
    lib.spvo_comm_rank.argtypes = [vp]
    lib.spvo_comm_world.argtypes = [vp]
    lib.spvo_comm_destroy.argtypes = [vp]
    lib.spvo_comm_destroy.restype = None
    lib.spvo_pose_allgather.argtypes = [vp, dp, dp]
    lib.spvo_pose_allgather_n.argtypes = [vp, dp, C.c_int, dp]
    _lib = lib
    return lib


class SpvoError(RuntimeError):
    def __init__(self, code: int, msg: str):
        super().__init__(f"spvo error {code}: {msg}")
        self.code = code


def _ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _dptr(a: np.ndarray):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class Context:
    """Thin object wrapper: one method per C entry point, numpy in / numpy out."""

    def __init__(self, **kw):
        self.lib = load()
        self.cfg = Config()
        self.lib.spvo_default_config(C.byref(self.cfg))
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise TypeError(k)
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        rc = self.lib.spvo_create(C.byref(self.cfg), C.byref(self.h))
        if rc:
            raise SpvoError(rc, self.lib.spvo_last_error(None).decode())
        self.H, self.W = self.cfg.net_height, self.cfg.net_width
        self.Hc, self.Wc = self.H // 8, self.W // 8
        self.cap = self.cfg.max_keypoints

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.spvo_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise SpvoError(rc, self.lib.spvo_last_error(self.h).decode())

    def load_weights(self, path: str):
        self._check(self.lib.spvo_load_weights(self.h, path.encode()))

    def preprocess(self, img: np.ndarray, P: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        img = np.ascontiguousarray(img, np.uint8)
        P2 = np.ascontiguousarray(P, np.float64).reshape(12).copy()
        out = np.empty((self.H, self.W), np.uint8)
        self._check(self.lib.spvo_preprocess(self.h, _ptr(img), img.shape[0], img.shape[1], img.strides[0],
                                             _dptr(P2), _ptr(out)))
        return out, P2.reshape(3, 4)

    def forward(self, x: np.ndarray) -> Tuple[np.ndarray, np.ndarray]:
        x = np.ascontiguousarray(x, np.float32)
        b = x.shape[0]
        assert x.shape == (b, 1, self.H, self.W)
        det = np.empty((b, 65, self.Hc, self.Wc), np.float32)
        desc = np.empty((b, self.Hc, self.Wc, 256), np.float32)
        self._check(self.lib.spvo_forward(self.h, _ptr(x), b, _ptr(det), _ptr(desc)))
        return det, desc

    def debug_tensor(self, tid: int, batch: int, channels: int, level: int) -> np.ndarray:
        out = np.empty((batch, channels, self.H >> level, self.W >> level), np.float32)
        self._check(self.lib.spvo_debug_tensor(self.h, tid, batch, _ptr(out), out.size))
        return out

    def heatmap(self, det: np.ndarray) -> np.ndarray:
        det = np.ascontiguousarray(det, np.float32)
        assert det.shape == (65, self.Hc, self.Wc)
        heat = np.empty((self.H, self.W), np.float32)
        self._check(self.lib.spvo_heatmap(self.h, _ptr(det), _ptr(heat)))
        return heat

    def nms(self, heat: np.ndarray) -> np.ndarray:
        heat = np.ascontiguousarray(heat, np.float32)
        assert heat.shape == (self.H, self.W)
        xy = np.zeros((self.cap, 2), np.int32)
        n = C.c_int(0)
        self._check(self.lib.spvo_nms(self.h, _ptr(heat), _ptr(xy), C.byref(n)))
        return xy[:n.value].copy()

    def sample_descriptors(self, desc_nhwc: np.ndarray, xy: np.ndarray) -> np.ndarray:
        desc_nhwc = np.ascontiguousarray(desc_nhwc, np.float32)
        assert desc_nhwc.shape == (self.Hc, self.Wc, 256)
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        out = np.empty((len(xy), 256), np.float32)
        self._check(self.lib.spvo_sample_descriptors(self.h, _ptr(desc_nhwc), _ptr(xy), len(xy), _ptr(out)))
        return out

    def _features(self, want_desc=True):
        xy = np.zeros((self.cap, 2), np.float32)
        desc = np.zeros((self.cap, 256), np.float32) if want_desc else None
        f = Features(0, xy.ctypes.data_as(C.POINTER(C.c_float)),
                     desc.ctypes.data_as(C.POINTER(C.c_float)) if want_desc else None)
        return f, xy, desc

    def detect(self, img_l: np.ndarray, img_r: np.ndarray, P_l, P_r, slot_l=2, slot_r=3, want_resized=False):
        img_l = np.ascontiguousarray(img_l, np.uint8)
        img_r = np.ascontiguousarray(img_r, np.uint8)
        assert img_l.shape == img_r.shape and img_l.strides == img_r.strides
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12).copy()
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12).copy()
        fl, xyl, dl = self._features()
        fr, xyr, dr = self._features()
        rl = np.empty((self.H, self.W), np.uint8) if want_resized else None
        rr = np.empty((self.H, self.W), np.uint8) if want_resized else None
        self._check(self.lib.spvo_detect(self.h, _ptr(img_l), _ptr(img_r), img_l.shape[0], img_l.shape[1],
                                         img_l.strides[0], _dptr(Pl), _dptr(Pr), slot_l, slot_r,
                                         C.byref(fl), C.byref(fr), _ptr(rl), _ptr(rr)))
        out = dict(xy_l=xyl[:fl.n].copy(), desc_l=dl[:fl.n].copy(), xy_r=xyr[:fr.n].copy(),
                   desc_r=dr[:fr.n].copy(), P_l=Pl.reshape(3, 4), P_r=Pr.reshape(3, 4))
        if want_resized:
            out["resized_l"], out["resized_r"] = rl, rr
        return out

    def detect_dev(self, d_img_l: int, d_img_r: int, rows: int, cols: int, stride: int, P_l, P_r,
                   slot_l=2, slot_r=3, want_desc=False):
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12).copy()
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12).copy()
        fl, xyl, dl = self._features(want_desc)
        fr, xyr, dr = self._features(want_desc)
        self._check(self.lib.spvo_detect_dev(self.h, C.c_void_p(d_img_l), C.c_void_p(d_img_r), rows, cols, stride,
                                             _dptr(Pl), _dptr(Pr), slot_l, slot_r, C.byref(fl), C.byref(fr)))
        return dict(xy_l=xyl[:fl.n], xy_r=xyr[:fr.n], desc_l=None if dl is None else dl[:fl.n],
                    desc_r=None if dr is None else dr[:fr.n], P_l=Pl.reshape(3, 4), P_r=Pr.reshape(3, 4))

    def set_fp32_split(self, enable: bool):
        """FP32 engines loaded after this call evaluate their convolutions on the bf16x3 split kernels (include/spvo.h)."""
        self._check(self.lib.spvo_set_fp32_split(self.h, int(enable)))

    def set_match_fp8(self, enable: bool):
        self._check(self.lib.spvo_set_match_fp8(self.h, int(enable)))

    def match_fp8(self) -> bool:
        rc = self.lib.spvo_get_match_fp8(self.h)
        self._check(min(rc, 0))
        return rc == 1

    def engine_precision(self) -> str:
        rc = self.lib.spvo_engine_precision(self.h)
        if rc < 0:
            raise SpvoError(rc, "no weights loaded")
        return {0: "FP32", 1: "FP16", 2: "INT8"}[rc]

    def detect_dev_submit(self, d_img_l: int, d_img_r: int, rows: int, cols: int, stride: int, slot_l: int, slot_r: int):
        """Enqueue a detector pass (at most six may be in flight); complete them oldest-first with detect_wait."""
        self._check(self.lib.spvo_detect_dev_submit(self.h, C.c_void_p(d_img_l), C.c_void_p(d_img_r), rows, cols, stride, slot_l, slot_r))

    def set_trunk_pairing(self, on: bool):
        self._check(self.lib.spvo_set_trunk_pairing(self.h, int(on)))

    def detect_submit(self, img_l: np.ndarray, img_r: np.ndarray, slot_l: int, slot_r: int, extras: int = 3):
        """Asynchronous detector pass on HOST images (spvo_detect_submit); complete with detect_collect."""
        img_l = np.ascontiguousarray(img_l, np.uint8)
        img_r = np.ascontiguousarray(img_r, np.uint8)
        self._check(self.lib.spvo_detect_submit(self.h, _ptr(img_l), _ptr(img_r), img_l.shape[0], img_l.shape[1], img_l.strides[0], slot_l, slot_r, extras))

    def detect_collect(self, P_l, P_r, want_desc=True, want_resized=True):
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12).copy()
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12).copy()
        fl, xyl, dl = self._features(want_desc)
        fr, xyr, dr = self._features(want_desc)
        rl = np.empty((self.H, self.W), np.uint8) if want_resized else None
        rr = np.empty((self.H, self.W), np.uint8) if want_resized else None
        self._check(self.lib.spvo_detect_collect(self.h, _dptr(Pl), _dptr(Pr), C.byref(fl), C.byref(fr), _ptr(rl), _ptr(rr)))
        out = dict(xy_l=xyl[:fl.n].copy(), xy_r=xyr[:fr.n].copy(), P_l=Pl.reshape(3, 4), P_r=Pr.reshape(3, 4))
        if want_desc:
            out["desc_l"], out["desc_r"] = dl[:fl.n].copy(), dr[:fr.n].copy()
        if want_resized:
            out["resized_l"], out["resized_r"] = rl, rr
        return out

    def detect_collect_mirrors(self, P_l, P_r):
        """spvo_detect_collect_mirrors: numpy VIEWS of the submission's pinned mirrors (valid until seven more submissions)"""
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12).copy()
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12).copy()
        m = DetectMirrors()
        self._check(self.lib.spvo_detect_collect_mirrors(self.h, _dptr(Pl), _dptr(Pr), C.byref(m)))
        self._check(self.lib.spvo_detect_mirrors_wait(self.h, C.byref(m)))      # the descriptors arrive beside the matches
        out = dict(P_l=Pl.reshape(3, 4), P_r=Pr.reshape(3, 4))
        for i, side in enumerate("lr"):
            n = m.n[i]
            out["xy_" + side] = np.ctypeslib.as_array(m.xy[i], (max(n, 1), 2))[:n]
            out["desc_" + side] = np.ctypeslib.as_array(m.desc[i], (max(n, 1), 256))[:n] if m.desc[i] else None
            out["resized_" + side] = np.ctypeslib.as_array(m.resized[i], (self.H, self.W)) if m.resized[i] else None
        return out

    def detect_wait(self, P_l, P_r):
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12).copy()
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12).copy()
        fl, xyl, _ = self._features(False)
        fr, xyr, _ = self._features(False)
        self._check(self.lib.spvo_detect_wait(self.h, _dptr(Pl), _dptr(Pr), C.byref(fl), C.byref(fr)))
        return dict(xy_l=xyl[:fl.n], xy_r=xyr[:fr.n], P_l=Pl.reshape(3, 4), P_r=Pr.reshape(3, 4))

    def orb(self, img: np.ndarray, nfeatures=2000):
        """ORB keypoints + descriptors of one u8 image (spvo_orb_detect): dict of xy [n,2], angle, response, octave, desc [n,32]."""
        img = np.ascontiguousarray(img, np.uint8)
        kp = np.zeros((nfeatures, 5), np.float32)          # x, y, angle, response, octave (int32 bits)
        desc = np.zeros((nfeatures, 32), np.uint8)
        n = C.c_int(0)
        self._check(self.lib.spvo_orb_detect(self.h, _ptr(img), img.shape[0], img.shape[1], img.strides[0], nfeatures, _ptr(kp), _ptr(desc), nfeatures, C.byref(n)))
        k = min(n.value, nfeatures)
        return dict(xy=kp[:k, :2].copy(), angle=kp[:k, 2].copy(), response=kp[:k, 3].copy(), octave=kp[:k, 4].copy().view(np.int32), desc=desc[:k].copy())

    def orb_tables(self):
        pat = np.zeros(1024, np.float32)
        taps = np.zeros(7, np.float32)
        self._check(self.lib.spvo_orb_tables(_ptr(pat), _ptr(taps)))
        return pat, taps

    def match_hamming(self, a: np.ndarray, b: np.ndarray, selector="KNN", cross_check=False, ratio=0.8):
        """cv::BFMatcher(NORM_HAMMING) on u8 descriptor rows (spvo_match_hamming)."""
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        nbytes = a.shape[1] if a.ndim == 2 and a.shape[1] else (b.shape[1] if b.ndim == 2 and b.shape[1] else 32)
        a = a.reshape(len(a), nbytes)
        b = b.reshape(len(b), nbytes)
        idx = np.full(len(a), -1, np.int32)
        dist = np.zeros(len(a), np.float32)
        self._check(self.lib.spvo_match_hamming(self.h, _ptr(a), len(a), _ptr(b), len(b), nbytes, 1 if selector == "KNN" else 0,
                                                int(cross_check), ratio, _ptr(idx), _ptr(dist)))
        return idx, dist

    def match(self, a: np.ndarray, b: np.ndarray, selector="KNN", cross_check=False, ratio=0.8):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 256)
        b = np.ascontiguousarray(b, np.float32).reshape(-1, 256)
        idx = np.full(len(a), -1, np.int32)
        dist = np.zeros(len(a), np.float32)
        self._check(self.lib.spvo_match(self.h, _ptr(a), len(a), _ptr(b), len(b), 1 if selector == "KNN" else 0,
                                        int(cross_check), ratio, _ptr(idx), _ptr(dist)))
        return idx, dist

    def match_slots(self, slot_a: int, slot_b: int, n_a: int, selector="KNN", cross_check=False, ratio=0.8):
        idx = np.full(max(n_a, 1), -1, np.int32)
        dist = np.zeros(max(n_a, 1), np.float32)
        self._check(self.lib.spvo_match_slots(self.h, slot_a, slot_b, 1 if selector == "KNN" else 0,
                                              int(cross_check), ratio, _ptr(idx), _ptr(dist)))
        return idx[:n_a], dist[:n_a]

    def set_prematch(self, enable=True, selector="KNN", cross_check=False, ratio=0.8):
        self._check(self.lib.spvo_set_prematch(self.h, int(enable), 1 if selector == "KNN" else 0, int(cross_check), ratio))

    def triangulate(self, P_l, P_r, xy_l, xy_r) -> np.ndarray:
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12)
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12)
        a = np.ascontiguousarray(xy_l, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(xy_r, np.float32).reshape(-1, 2)
        out = np.zeros((len(a), 3), np.float32)
        self._check(self.lib.spvo_triangulate(self.h, _dptr(Pl), _dptr(Pr), _ptr(a), _ptr(b), len(a), _ptr(out)))
        return out

    def pnp_ransac(self, K, xyz, xy, rvec0, tvec0, iterations=500, reproj_error=2.0, seed=0):
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        r = np.ascontiguousarray(rvec0, np.float64).reshape(3).copy()
        t = np.ascontiguousarray(tvec0, np.float64).reshape(3).copy()
        inl = np.zeros(max(len(xyz), 1), np.int32)
        n_inl, ok = C.c_int(0), C.c_int(0)
        opts = RansacOpts(iterations, reproj_error, 0.999, seed)
        self._check(self.lib.spvo_pnp_ransac(self.h, _dptr(K), _ptr(xyz), _ptr(xy), len(xyz), C.byref(opts),
                                             _dptr(r), _dptr(t), _ptr(inl), C.byref(n_inl), C.byref(ok)))
        return bool(ok.value), r, t, inl[:n_inl.value].copy()

    def pnp_refine(self, P_l, P_r, obs: np.ndarray, q0, t0, max_iterations=40, huber_delta=1.0):
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12)
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12)
        obs = np.ascontiguousarray(obs, OBS_DTYPE)
        q = np.ascontiguousarray(q0, np.float64).reshape(4).copy()
        t = np.ascontiguousarray(t0, np.float64).reshape(3).copy()
        s = RefineSummary()
        opts = RefineOpts(max_iterations, huber_delta)
        self._check(self.lib.spvo_pnp_refine(self.h, _dptr(Pl), _dptr(Pr), _ptr(obs), len(obs), C.byref(opts),
                                             _dptr(q), _dptr(t), C.byref(s)))
        return q, t, s

    def solve(self, P_l, P_r, cl, cr, pl, pr, prev_xyz=None, prev_valid=None, rvec_pred=(0, 0, 0), tvec_pred=(0, 0, 0),
              frame_count=0, refinement_degree=4, seed=0, iterations=500, reproj_error=2.0, max_iterations=40, split=None, prev_index=None, late_prior=False):
        """spvo_solve_stereo_odometry, or (split="submit") spvo_solve_submit alone: complete with solve_wait(n) / solve_wait_prior(n, ...).
        prev_index: instead of prev_xyz / prev_valid, indices into the points of the previous submit on this context (-1: none);
        late_prior: the prior is handed to solve_wait_prior instead (the previous solve may still be in flight); 2: and the chain's last kernel
        is held back for the next submission's launch (include/spvo.h)."""
        arrs = [np.ascontiguousarray(a, np.float32).reshape(-1, 2) for a in (cl, cr, pl, pr)]
        n = len(arrs[0])
        si = SolveInput()
        si.n = n
        si.xy_cl, si.xy_cr, si.xy_pl, si.xy_pr = [a.ctypes.data for a in arrs]
        if prev_xyz is not None:
            px = np.ascontiguousarray(prev_xyz, np.float32).reshape(-1, 3)
            pv = np.ascontiguousarray(prev_valid, np.int32)
            si.prev_xyz, si.prev_valid = px.ctypes.data, pv.ctypes.data
        si.P_l[:] = np.asarray(P_l, np.float64).reshape(12).tolist()
        si.P_r[:] = np.asarray(P_r, np.float64).reshape(12).tolist()
        si.rvec_pred[:] = list(map(float, rvec_pred))
        si.tvec_pred[:] = list(map(float, tvec_pred))
        si.frame_count, si.refinement_degree = frame_count, refinement_degree
        si.ransac = RansacOpts(iterations, reproj_error, 0.999, seed)
        si.refine = RefineOpts(max_iterations, 1.0)
        if prev_index is not None:
            pi = np.ascontiguousarray(prev_index, np.int32)
            si.prev_index = pi.ctypes.data
        si.late_prior = int(late_prior)
        if split == "submit":      # spvo_solve_submit only: the inputs are staged, complete with solve_wait(n)
            self._check(self.lib.spvo_solve_submit(self.h, C.byref(si)))
            return n
        so = SolveOutput()
        xyz = np.zeros((max(n, 1), 3), np.float32)
        inl = np.zeros(max(n, 1), np.int32)
        self._check(self.lib.spvo_solve_stereo_odometry(self.h, C.byref(si), C.byref(so), _ptr(xyz), _ptr(inl)))
        return self._solve_result(so, xyz, inl, n)

    def solve_wait(self, n):
        """Second half of solve(..., split="submit")."""
        so = SolveOutput()
        xyz = np.zeros((max(n, 1), 3), np.float32)
        inl = np.zeros(max(n, 1), np.int32)
        self._check(self.lib.spvo_solve_wait(self.h, C.byref(so), _ptr(xyz), _ptr(inl)))
        return self._solve_result(so, xyz, inl, n)

    def solve_wait_prior(self, n, rvec_pred, tvec_pred, frame_count):
        """spvo_solve_wait_prior: the oldest pending solve, gated against the prior given NOW."""
        so = SolveOutput()
        xyz = np.zeros((max(n, 1), 3), np.float32)
        inl = np.zeros(max(n, 1), np.int32)
        r = np.ascontiguousarray(rvec_pred, np.float64).reshape(3)
        t = np.ascontiguousarray(tvec_pred, np.float64).reshape(3)
        self._check(self.lib.spvo_solve_wait_prior(self.h, _dptr(r), _dptr(t), int(frame_count), C.byref(so), _ptr(xyz), _ptr(inl)))
        return self._solve_result(so, xyz, inl, n)

    def solve_pending(self):
        return int(self.lib.spvo_solve_pending(self.h))

    @staticmethod
    def _solve_result(so, xyz, inl, n):
        return dict(q=np.array(so.q[:]), t=np.array(so.t[:]), rvec=np.array(so.rvec[:]), tvec=np.array(so.tvec[:]),
                    pnp_ok=bool(so.pnp_ok), accepted=bool(so.accepted), refined=bool(so.refined),
                    inliers=inl[:so.n_inliers].copy(), xyz=xyz[:n].copy(), iterations=so.summary.iterations,
                    converged=bool(so.summary.converged), final_cost=so.summary.final_cost)

    def stream(self) -> int:
        return self.lib.spvo_stream(self.h) or 0

    def synchronize(self):
        self._check(self.lib.spvo_synchronize(self.h))

    def profile_enable(self, on=True):
        self._check(self.lib.spvo_profile_enable(self.h, int(on)))

    def profile_reset(self):
        self._check(self.lib.spvo_profile_reset(self.h))

    def profile_only(self, stage=None):
        """Time one stage only (e.g. "conv:1"); None = every stage."""
        self._check(self.lib.spvo_profile_only(self.h, stage.encode() if stage else None))

    def profile(self):
        out = {}
        n = self.lib.spvo_profile_count(self.h)
        for i in range(n):
            name = C.create_string_buffer(64)
            ms, fl, by = C.c_double(), C.c_double(), C.c_double()
            calls = C.c_longlong()
            self._check(self.lib.spvo_profile_get(self.h, i, name, 64, C.byref(ms), C.byref(calls), C.byref(fl), C.byref(by)))
            out[name.value.decode()] = dict(total_ms=ms.value, calls=calls.value, flops=fl.value, bytes=by.value)
        return out

    def stage_kernel(self, stage):
        """(kernel family, executed / algorithmic multiply-adds) of a "conv:<op index>" stage of the loaded engine."""
        name = C.create_string_buffer(64)
        f = C.c_double()
        self._check(self.lib.spvo_profile_stage_kernel(self.h, stage.encode(), name, 64, C.byref(f)))
        return name.value.decode(), f.value


def set_tuning(name: str, value: int):
    """spvo_set_tuning: a diagnostic switch of the library (process-wide; contexts created / engines loaded afterwards).  The library
    reads no environment variable for these: tests and tools set them through this call."""
    rc = load().spvo_set_tuning(name.encode(), int(value))
    if rc:
        raise SpvoError(rc, f"unknown tuning name {name!r}")


def get_tuning(name: str, default: int) -> int:
    return load().spvo_get_tuning(name.encode(), int(default))


def clear_tuning():
    load().spvo_clear_tuning()


def tuning_from_env(prefix: str = "SPVO_TUNE_"):
    """tools/ only: SPVO_TUNE_<NAME>=<int> in the environment of a measurement script -> set_tuning (the script opts in by calling
    this; the library itself never looks)."""
    for k, v in os.environ.items():
        if k.startswith(prefix):
            set_tuning(k[len(prefix):].lower(), int(v))


COMM_ID_BYTES = 128


def comm_available() -> bool:
    """spvo_comm_available: librccl opens and has the entry points the library uses (no bootstrap state is created)"""
    return load().spvo_comm_available() == 0


def comm_unique_id() -> bytes:
    """rank 0: the RCCL id every rank passes to Comm.rccl (hand it over out of band)."""
    lib = load()
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = lib.spvo_comm_unique_id(buf)
    if rc:
        raise SpvoError(rc, lib.spvo_last_error(None).decode())
    return buf.raw


class Comm:
    """spvo_comm: the pose all-gather of include/spvo.h (RCCL over xGMI; `host` = the file transport of the CPU tests)."""

    def __init__(self, handle, lib):
        self.h, self.lib = handle, lib
        self.rank, self.world = lib.spvo_comm_rank(handle), lib.spvo_comm_world(handle)

    @classmethod
    def rccl(cls, device: int, rank: int, world: int, unique_id: bytes) -> "Comm":
        lib = load()
        h = C.c_void_p()
        rc = lib.spvo_comm_create(device, rank, world, C.create_string_buffer(unique_id, COMM_ID_BYTES), C.byref(h))
        if rc:
            raise SpvoError(rc, lib.spvo_last_error(None).decode())
        return cls(h, lib)

    @classmethod
    def host(cls, directory: str, rank: int, world: int) -> "Comm":
        lib = load()
        h = C.c_void_p()
        rc = lib.spvo_comm_create_host(directory.encode(), rank, world, C.byref(h))
        if rc:
            raise SpvoError(rc, lib.spvo_last_error(None).decode())
        return cls(h, lib)

    def allgather(self, poses: np.ndarray) -> np.ndarray:
        """poses [n, 7] (or [7]) float64 -> [world, n, 7]"""
        p = np.ascontiguousarray(poses, np.float64).reshape(-1, 7)
        out = np.empty((self.world, len(p), 7), np.float64)
        rc = self.lib.spvo_pose_allgather_n(self.h, _dptr(p), len(p), _dptr(out))
        if rc:
            raise SpvoError(rc, self.lib.spvo_last_error(None).decode())
        return out

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.spvo_comm_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def obs_array(X, uv, cam, inverse) -> np.ndarray:
    n = len(X)
    a = np.zeros(n, OBS_DTYPE)
    if n:
        a["X"] = np.asarray(X, np.float32).reshape(n, 3)
        a["uv"] = np.asarray(uv, np.float32).reshape(n, 2)
        a["cam"] = np.asarray(cam, np.int32)
        a["inverse"] = np.asarray(inverse, np.int32)
    return a
