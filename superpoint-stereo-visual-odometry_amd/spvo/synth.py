"""Synthetic KITTI-like stereo sequences with exact ground-truth poses.

Neither KITTI bags nor right-camera images exist in the reference tree
(src/odml_visual_odometry/sample_images holds 22 LEFT frames only), so tests
and bench.py render their own stereo pairs: a piece-wise planar world (ground
plane + a fronto-parallel wall + two side walls) textured with a real sample
frame (or seeded noise), viewed by a rectified stereo rig that moves with a
known ego-motion.  Calibration constants are the public KITTI odometry ones
(fx = fy = 718.856, cx = 607.1928, cy = 185.2157, baseline 0.5372 m), shaped as
the 3x4 P matrices visual_odometry_node.cpp:84-98 builds from CameraInfo.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Optional, Tuple

import numpy as np

FX = 718.856
CX = 607.1928
CY = 185.2157
BASELINE = 0.5372
ROWS, COLS = 376, 1241


def projection_matrices() -> Tuple[np.ndarray, np.ndarray]:
    P_l = np.array([[FX, 0, CX, 0], [0, FX, CY, 0], [0, 0, 1, 0]], np.float64)
    P_r = P_l.copy()
    P_r[0, 3] = -FX * BASELINE           # -386.1448
    return P_l, P_r


def _rot_y(a: float) -> np.ndarray:
    c, s = np.cos(a), np.sin(a)
    return np.array([[c, 0, s], [0, 1, 0], [-s, 0, c]])


def ego_motion(n_frames: int, seed: int = 0, step: float = 0.8, yaw: float = 0.004):
    """world_T_cam for each frame: ~`step` m forward per frame with a gentle seeded yaw."""
    rng = np.random.RandomState(seed)
    R = np.eye(3)
    t = np.zeros(3)
    poses = []
    for _ in range(n_frames):
        poses.append((R.copy(), t.copy()))
        dyaw = yaw * (1.0 + 0.5 * rng.randn())
        dt = np.array([0.02 * rng.randn(), 0.0, step * (1.0 + 0.05 * rng.randn())])
        t = t + R @ dt
        R = R @ _rot_y(dyaw)
    return poses


@dataclass
class Scene:
    texture: np.ndarray              # float32 [th, tw] in [0, 255]
    cam_height: float = 1.65         # ground plane at y = +cam_height (y points down)
    wall_z: float = 60.0
    side_x: float = 9.0
    tex_scale: float = 28.0          # texture pixels per metre


def load_texture(path: Optional[str], seed: int = 1) -> np.ndarray:
    if path and os.path.exists(path):
        from PIL import Image
        return np.asarray(Image.open(path).convert("L"), np.float32)
    rng = np.random.RandomState(seed)
    # band-limited noise: corners everywhere, but smooth enough to survive resampling
    a = rng.rand(96, 312).astype(np.float32)
    a = np.kron(a, np.ones((4, 4), np.float32))
    k = np.array([1, 4, 6, 4, 1], np.float32) / 16
    a = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 1, a)
    a = np.apply_along_axis(lambda r: np.convolve(r, k, mode="same"), 0, a)
    a = (a - a.min()) / (a.max() - a.min())
    return (a * 255).astype(np.float32)


def _sample(tex: np.ndarray, u: np.ndarray, v: np.ndarray) -> np.ndarray:
    th, tw = tex.shape
    u = np.mod(u, tw - 1.0)
    v = np.mod(v, th - 1.0)
    u0 = np.floor(u).astype(np.int64)
    v0 = np.floor(v).astype(np.int64)
    fu, fv = (u - u0).astype(np.float32), (v - v0).astype(np.float32)
    a = tex[v0, u0] * (1 - fu) + tex[v0, u0 + 1] * fu
    b = tex[v0 + 1, u0] * (1 - fu) + tex[v0 + 1, u0 + 1] * fu
    return a * (1 - fv) + b * fv


def render(scene: Scene, R: np.ndarray, t: np.ndarray, rows: int = ROWS, cols: int = COLS,
           cam_offset_x: float = 0.0) -> Tuple[np.ndarray, np.ndarray]:
    """Image (u8) and depth (f32, metres along the optical axis) of a camera at world pose (R, t)
    shifted by `cam_offset_x` along its own x axis (right camera: +BASELINE)."""
    ys, xs = np.mgrid[0:rows, 0:cols]
    d = np.stack([(xs - CX) / FX, (ys - CY) / FX, np.ones_like(xs, np.float64)], -1)   # camera rays
    dw = d @ R.T                                                                      # world rays
    o = t + R @ np.array([cam_offset_x, 0, 0])
    big = 1e9
    with np.errstate(divide="ignore", invalid="ignore"):
        lam_g = np.where(dw[..., 1] > 1e-9, (scene.cam_height - o[1]) / dw[..., 1], big)
        lam_w = np.where(dw[..., 2] > 1e-9, (scene.wall_z + 0.0 - o[2]) / dw[..., 2], big)
        lam_l = np.where(dw[..., 0] < -1e-9, (-scene.side_x - o[0]) / dw[..., 0], big)
        lam_r = np.where(dw[..., 0] > 1e-9, (scene.side_x - o[0]) / dw[..., 0], big)
    lam = np.stack([lam_g, lam_w, lam_l, lam_r], -1)
    lam = np.where(lam > 1e-6, lam, big)
    which = np.argmin(lam, -1)
    lmin = np.take_along_axis(lam, which[..., None], -1)[..., 0]
    X = o + dw * lmin[..., None]
    s = scene.tex_scale
    u = np.select([which == 0, which == 1, which == 2, which == 3],
                  [X[..., 0] * s + 300.0, X[..., 0] * s * 0.5 + 600.0, X[..., 2] * s * 0.7 + 100.0, X[..., 2] * s * 0.7 + 900.0])
    v = np.select([which == 0, which == 1, which == 2, which == 3],
                  [X[..., 2] * s, X[..., 1] * s * 0.5 + 150.0, X[..., 1] * s * 0.7 + 200.0, X[..., 1] * s * 0.7 + 200.0])
    img = _sample(scene.texture, u, v)
    depth = (lmin * d[..., 2]).astype(np.float32)
    return np.clip(np.rint(img), 0, 255).astype(np.uint8), depth


def stereo_sequence(n_frames: int, texture_path: Optional[str] = None, seed: int = 0,
                    rows: int = ROWS, cols: int = COLS, drop=()):
    """Returns (frames [(left u8, right u8)], poses [(R, t) world_T_cam], P_l, P_r).

    `drop`: indices of the underlying ego-motion that are left out (never rendered): n_frames frames are still returned, and the step
    across a gap is a PLANTED JUMP of (1 + gap) x the usual ~0.8 m -- what the reference's acceleration gate (base.cpp:251-260:
    |t - t_pred| / 0.1 s > 8 m/s^2 once frame_count > IGNORE_FRAME_COUNT) exists to reject."""
    tex = load_texture(texture_path, seed + 1)
    drop = set(int(d) for d in drop)
    poses = [p for k, p in enumerate(ego_motion(n_frames + len(drop), seed)) if k not in drop]
    frames = []
    scene = Scene(tex, wall_z=poses[-1][1][2] + 45.0)     # static world: the wall stays put
    for R, t in poses:
        left, _ = render(scene, R, t, rows, cols, 0.0)
        right, _ = render(scene, R, t, rows, cols, BASELINE)
        frames.append((left, right))
    P_l, P_r = projection_matrices()
    return frames, poses, P_l, P_r


def relative_pose(pose_prev, pose_curr):
    """cam0_curr_T_cam0_prev (R, t): x_curr = R x_prev + t, the quantity
    solveStereoOdometry returns (base.cpp:377-385)."""
    Rp, tp = pose_prev
    Rc, tc = pose_curr
    R = Rc.T @ Rp
    t = Rc.T @ (tp - tc)
    return R, t
