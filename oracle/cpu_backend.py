"""ctypes binding of oracle/cpu/libspvo_cpu.so -- the CPU restatement of the hot path in C++17 + OpenMP.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): the compiled second oracle (tests/test_cpu_backend.py checks it against
the numpy / torch restatement in this directory) and the CPU timing baseline of bench.py.  Nothing in the product
package imports this module or loads that library.
"""
from __future__ import annotations

import ctypes as C
import os
import shutil
import subprocess
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
SRC_DIR = os.path.join(_HERE, "cpu")
LIB_PATH = os.path.join(SRC_DIR, "libspvo_cpu.so")


class Config(C.Structure):
    _fields_ = [("net_height", C.c_int), ("net_width", C.c_int), ("conf_thresh", C.c_float), ("dist_thresh", C.c_int),
                ("border_remove", C.c_int), ("max_keypoints", C.c_int), ("bug_compat_p", C.c_int), ("num_threads", C.c_int)]


class RefineSummary(C.Structure):
    _fields_ = [("iterations", C.c_int), ("converged", C.c_int), ("usable", C.c_int), ("initial_cost", C.c_double), ("final_cost", C.c_double)]


class StepResult(C.Structure):
    _fields_ = [("t_detect_ms", C.c_float), ("t_match_ms", C.c_float), ("t_solve_ms", C.c_float), ("t_total_ms", C.c_float),
                ("n_kp_l", C.c_int), ("n_kp_r", C.c_int), ("n_stereo", C.c_int), ("n_temporal", C.c_int), ("n_joined", C.c_int),
                ("n_inliers", C.c_int), ("pnp_ok", C.c_int), ("accepted", C.c_int), ("refined", C.c_int), ("lm_iterations", C.c_int),
                ("q", C.c_double * 4), ("t", C.c_double * 3)]


OBS_DTYPE = np.dtype([("X", np.float32, 3), ("uv", np.float32, 2), ("cam", np.int32), ("inverse", np.int32)])


def build(arch: Optional[str] = None, out: Optional[str] = None) -> str:
    """Compile the library (g++ -O3 -fopenmp).  arch = "native" targets the host this runs on (what bench.py does on the
    machine whose cores it times); default x86-64-v3 so the file built here also loads elsewhere."""
    if not shutil.which("g++") or not shutil.which("make"):
        raise RuntimeError("g++ / make not available")
    cmd = ["make", "-C", SRC_DIR, "-B" if (arch or out) else "-s"]
    if arch:
        cmd.append(f"CPU_ARCH={arch}")
    if out:
        cmd.append(f"OUT={out}")
    subprocess.run(cmd, check=True, capture_output=True)
    return out or LIB_PATH


def load(path: Optional[str] = None) -> C.CDLL:
    path = path or LIB_PATH
    if not os.path.exists(path):
        build()
    lib = C.CDLL(path)
    vp, ip, dp = C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)
    lib.spvo_cpu_default_config.argtypes = [C.POINTER(Config)]
    lib.spvo_cpu_default_config.restype = None
    lib.spvo_cpu_create.argtypes = [C.POINTER(Config), C.POINTER(vp)]
    lib.spvo_cpu_destroy.argtypes = [vp]
    lib.spvo_cpu_destroy.restype = None
    lib.spvo_cpu_last_error.restype = C.c_char_p
    lib.spvo_cpu_threads.argtypes = [vp]
    lib.spvo_cpu_load_weights.argtypes = [vp, C.c_char_p]
    lib.spvo_cpu_preprocess.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, dp, vp]
    lib.spvo_cpu_forward.argtypes = [vp, vp, C.c_int, vp, vp]
    lib.spvo_cpu_heatmap.argtypes = [vp, vp, vp]
    lib.spvo_cpu_nms.argtypes = [vp, vp, vp, ip]
    lib.spvo_cpu_sample_descriptors.argtypes = [vp, vp, vp, C.c_int, vp]
    lib.spvo_cpu_detect.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, dp, vp, vp, ip]
    lib.spvo_cpu_match.argtypes = [vp, vp, C.c_int, vp, C.c_int, C.c_int, C.c_int, C.c_float, vp, vp]
    lib.spvo_cpu_triangulate.argtypes = [dp, dp, vp, vp, C.c_int, vp]
    lib.spvo_cpu_pnp_ransac.argtypes = [dp, vp, vp, C.c_int, C.c_int, C.c_double, C.c_uint32, dp, dp, vp, ip, ip]
    lib.spvo_cpu_pnp_refine.argtypes = [dp, dp, vp, C.c_int, C.c_int, C.c_double, dp, dp, C.POINTER(RefineSummary)]
    lib.spvo_cpu_frontend_reset.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_float, C.c_int]
    lib.spvo_cpu_frontend_step.argtypes = [vp, vp, vp, C.c_int, C.c_int, C.c_size_t, dp, dp, C.POINTER(StepResult)]
    lib.spvo_cpu_frontend_map.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.spvo_cpu_frontend_keypoints.argtypes = [vp, C.c_int, vp, C.c_int]
    lib.spvo_cpu_frontend_reset_classic.argtypes = [vp, C.c_int, C.c_int, C.c_float, C.c_int]
    lib.spvo_cpu_orb.argtypes = [vp, vp, C.c_int, C.c_int, C.c_size_t, vp, vp, vp, C.c_int, ip]
    lib.spvo_cpu_orb_tables.argtypes = [vp, vp]
    return lib


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _d(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class CpuBackend:
    def __init__(self, lib_path: Optional[str] = None, **kw):
        self.lib = load(lib_path)
        self.cfg = Config()
        self.lib.spvo_cpu_default_config(C.byref(self.cfg))
        for k, v in kw.items():
            if not hasattr(self.cfg, k):
                raise TypeError(k)
            setattr(self.cfg, k, v)
        self.h = C.c_void_p()
        self._check(self.lib.spvo_cpu_create(C.byref(self.cfg), C.byref(self.h)))
        self.H, self.W = self.cfg.net_height, self.cfg.net_width
        self.Hc, self.Wc, self.cap = self.H // 8, self.W // 8, self.cfg.max_keypoints
        self.threads = self.lib.spvo_cpu_threads(self.h)

    def _check(self, rc):
        if rc:
            raise RuntimeError(f"spvo_cpu error {rc}: {self.lib.spvo_cpu_last_error().decode()}")

    def close(self):
        if getattr(self, "h", None) and self.h.value:
            self.lib.spvo_cpu_destroy(self.h)
            self.h = C.c_void_p()

    def load_weights(self, path: str):
        self._check(self.lib.spvo_cpu_load_weights(self.h, path.encode()))

    def preprocess(self, img, P):
        img = np.ascontiguousarray(img, np.uint8)
        P2 = np.ascontiguousarray(P, np.float64).reshape(12).copy()
        out = np.empty((self.H, self.W), np.uint8)
        self._check(self.lib.spvo_cpu_preprocess(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], _d(P2), _p(out)))
        return out, P2.reshape(3, 4)

    def forward(self, x):
        x = np.ascontiguousarray(x, np.float32)
        b = x.shape[0]
        det = np.empty((b, 65, self.Hc, self.Wc), np.float32)
        desc = np.empty((b, 256, self.Hc, self.Wc), np.float32)
        self._check(self.lib.spvo_cpu_forward(self.h, _p(x), b, _p(det), _p(desc)))
        return det, desc

    def heatmap(self, det):
        det = np.ascontiguousarray(det, np.float32)
        heat = np.empty((self.H, self.W), np.float32)
        self._check(self.lib.spvo_cpu_heatmap(self.h, _p(det), _p(heat)))
        return heat

    def nms(self, heat):
        heat = np.ascontiguousarray(heat, np.float32)
        xy = np.zeros((self.cap, 2), np.int32)
        n = C.c_int(0)
        self._check(self.lib.spvo_cpu_nms(self.h, _p(heat), _p(xy), C.byref(n)))
        return xy[:n.value].copy()

    def sample_descriptors(self, desc_nchw, xy):
        desc_nchw = np.ascontiguousarray(desc_nchw, np.float32)
        xy = np.ascontiguousarray(xy, np.int32).reshape(-1, 2)
        out = np.empty((len(xy), 256), np.float32)
        self._check(self.lib.spvo_cpu_sample_descriptors(self.h, _p(desc_nchw), _p(xy), len(xy), _p(out)))
        return out

    def detect(self, img, P):
        img = np.ascontiguousarray(img, np.uint8)
        P2 = np.ascontiguousarray(P, np.float64).reshape(12).copy()
        xy = np.zeros((self.cap, 2), np.float32)
        desc = np.zeros((self.cap, 256), np.float32)
        n = C.c_int(0)
        self._check(self.lib.spvo_cpu_detect(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], _d(P2), _p(xy), _p(desc), C.byref(n)))
        return dict(xy=xy[:n.value].copy(), descriptors=desc[:n.value].copy(), P=P2.reshape(3, 4))

    def match(self, a, b, selector="KNN", cross_check=False, ratio=0.8):
        a = np.ascontiguousarray(a, np.float32).reshape(-1, 256)
        b = np.ascontiguousarray(b, np.float32).reshape(-1, 256)
        idx = np.full(max(len(a), 1), -1, np.int32)
        dist = np.zeros(max(len(a), 1), np.float32)
        self._check(self.lib.spvo_cpu_match(self.h, _p(a), len(a), _p(b), len(b), 1 if selector == "KNN" else 0, int(cross_check), ratio, _p(idx), _p(dist)))
        return idx[:len(a)], dist[:len(a)]

    def triangulate(self, P_l, P_r, xy_l, xy_r):
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12)
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12)
        a = np.ascontiguousarray(xy_l, np.float32).reshape(-1, 2)
        b = np.ascontiguousarray(xy_r, np.float32).reshape(-1, 2)
        out = np.zeros((len(a), 3), np.float32)
        self._check(self.lib.spvo_cpu_triangulate(_d(Pl), _d(Pr), _p(a), _p(b), len(a), _p(out)))
        return out

    def pnp_ransac(self, K, xyz, xy, rvec0, tvec0, iterations=500, reproj_error=2.0, seed=0):
        K = np.ascontiguousarray(K, np.float64).reshape(9)
        xyz = np.ascontiguousarray(xyz, np.float32).reshape(-1, 3)
        xy = np.ascontiguousarray(xy, np.float32).reshape(-1, 2)
        r = np.ascontiguousarray(rvec0, np.float64).reshape(3).copy()
        t = np.ascontiguousarray(tvec0, np.float64).reshape(3).copy()
        inl = np.zeros(max(len(xyz), 1), np.int32)
        n, ok = C.c_int(0), C.c_int(0)
        self._check(self.lib.spvo_cpu_pnp_ransac(_d(K), _p(xyz), _p(xy), len(xyz), iterations, reproj_error, seed, _d(r), _d(t), _p(inl), C.byref(n), C.byref(ok)))
        return bool(ok.value), r, t, inl[:n.value].copy()

    def pnp_refine(self, P_l, P_r, obs, q0, t0, max_iterations=40, huber_delta=1.0):
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12)
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12)
        obs = np.ascontiguousarray(obs, OBS_DTYPE)
        q = np.ascontiguousarray(q0, np.float64).reshape(4).copy()
        t = np.ascontiguousarray(t0, np.float64).reshape(3).copy()
        s = RefineSummary()
        self._check(self.lib.spvo_cpu_pnp_refine(_d(Pl), _d(Pr), _p(obs), len(obs), max_iterations, huber_delta, _d(q), _d(t), C.byref(s)))
        return q, t, s

    def frontend_reset(self, selector="KNN", cross_check=True, stereo_threshold=2.0, min_disparity=0.25, refinement_degree=4):
        self._check(self.lib.spvo_cpu_frontend_reset(self.h, 1 if selector == "KNN" else 0, int(cross_check), stereo_threshold, min_disparity, refinement_degree))

    def frontend_step(self, img_l, img_r, P_l, P_r) -> StepResult:
        img_l = np.ascontiguousarray(img_l, np.uint8)
        img_r = np.ascontiguousarray(img_r, np.uint8)
        Pl = np.ascontiguousarray(P_l, np.float64).reshape(12)
        Pr = np.ascontiguousarray(P_r, np.float64).reshape(12)
        res = StepResult()
        self._check(self.lib.spvo_cpu_frontend_step(self.h, _p(img_l), _p(img_r), img_l.shape[0], img_l.shape[1], img_l.strides[0], _d(Pl), _d(Pr), C.byref(res)))
        return res

    def frontend_map(self, match_type: int) -> np.ndarray:
        cap = max(self.cap, 4096)
        out = np.zeros(cap, np.int32)
        n = self.lib.spvo_cpu_frontend_map(self.h, match_type, _p(out), cap)
        return out[:n].copy()

    def frontend_keypoints(self, position: int) -> np.ndarray:
        """keypoints of deque position -4..-1 (prevL, prevR, currL, currR) after the last frontend_step: [n, 2] float32"""
        cap = max(self.cap, 4096)
        xy = np.zeros((cap, 2), np.float32)
        n = self.lib.spvo_cpu_frontend_keypoints(self.h, position, _p(xy), cap)
        return xy[:max(n, 0)].copy()

    def frontend_reset_classic(self, selector="KNN", cross_check=True, stereo_threshold=2.0, refinement_degree=4):
        """ClassicFeatureFrontEnd(ORB, ORB, BF, ...) of launch/visual_odometry_classic.launch on the CPU (BASELINE config 1)."""
        self._check(self.lib.spvo_cpu_frontend_reset_classic(self.h, 1 if selector == "KNN" else 0, int(cross_check), stereo_threshold, refinement_degree))

    def orb_tables(self):
        pat = np.zeros(1024, np.float32)
        taps = np.zeros(7, np.float32)
        self._check(self.lib.spvo_cpu_orb_tables(_p(pat), _p(taps)))
        return pat, taps

    def orb(self, img, cap=4096):
        img = np.ascontiguousarray(img, np.uint8)
        xy = np.zeros((cap, 2), np.float32)
        aro = np.zeros((cap, 3), np.float32)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int(0)
        self._check(self.lib.spvo_cpu_orb(self.h, _p(img), img.shape[0], img.shape[1], img.strides[0], _p(xy), _p(aro), _p(desc), cap, C.byref(n)))
        k = min(n.value, cap)
        return dict(xy=xy[:k].copy(), angle=aro[:k, 0].copy(), response=aro[:k, 1].copy(), octave=aro[:k, 2].astype(np.int32), desc=desc[:k].copy())
