"""Caller-side result files (pose integration, KITTI pose file, latency CSV): the C++ host code
against the oracle's restatement of visual_odometry_node.cpp / data_processing_node.cpp.  No GPU."""
import os

import numpy as np

import oracle  # noqa: F401
from oracle import odometry as od, vo_io
from spvo import host, synth


def _rel_poses(n, seed=0):
    poses = synth.ego_motion(n + 1, seed)
    out = []
    for k in range(1, n + 1):
        R, t = synth.relative_pose(poses[k - 1], poses[k])
        # rotation matrix -> quaternion via the axis-angle of a small yaw
        ang = np.arctan2(R[0, 2], R[0, 0])
        out.append((np.array([0, np.sin(ang / 2), 0, np.cos(ang / 2)]), t))
    return out


def test_kitti_pose_file_matches_oracle(tmp_path):
    rel = _rel_poses(12)
    rel[5] = (rel[5][0], np.array([0.0, 0.0, -25.0]))          # > 10 m step: rejected, last valid reused (node.cpp:118)
    base = (od.rvec_to_quat([0.3, -0.2, 0.1]), np.array([1.1, -0.3, 0.8]))   # base_link -> camera_gray_left
    n, (fq, ft) = host.write_kitti_poses(tmp_path, 4, rel, base, seq_start=2)
    assert n == 10
    name = vo_io.kitti_file_name(4)
    assert name == "04_pred.txt" and vo_io.kitti_file_name(13) == "13_pred.txt"     # dp.cpp:102-106
    got = open(os.path.join(tmp_path, name)).read().splitlines(keepends=True)
    W = vo_io.integrate(rel, base)
    ref = vo_io.kitti_lines(W, base, seq_start=2)
    assert len(got) == len(ref) == 10
    for g, r in zip(got, ref):
        gv, rv = np.array(g.split(), float), np.array(r.split(), float)
        assert g.endswith(" \n") and len(gv) == 12              # 12 numbers, each followed by a blank
        assert np.allclose(gv, rv, rtol=2e-5, atol=2e-6)        # 6 significant digits
    assert np.allclose(np.array(got[0].split(), float).reshape(3, 4), np.eye(4)[:3], atol=1e-12)   # first pose = identity
    assert np.allclose(ft, W[-1][:3, 3], atol=1e-9)
    # KITTI convention check: identity extrinsics -> the file holds the camera trajectory itself
    n, _ = host.write_kitti_poses(tmp_path, 0, rel[:3])
    lines = open(os.path.join(tmp_path, "00_pred.txt")).read().splitlines()
    P = np.array(lines[-1].split(), float).reshape(3, 4)
    # ... relative to the FIRST written pose (dp.cpp:159-162): frame 0 -> frame 2 is two steps
    T = np.eye(4)
    for q, t in rel[1:3]:
        T = T @ np.linalg.inv(vo_io._mat(q, t))
    assert np.allclose(P, T[:3], atol=2e-5)


def test_latency_csv(tmp_path):
    rows = [(7.391, 6.032, 1.631, 15.054), (1.5, 0.25, 0.125, 1.875)]
    n, name = host.write_latency_csv(tmp_path, "superpoint_pretrained", 2, 360, 1176, "FP32", 4, rows)
    assert n == 2 and name == vo_io.latency_file_name("superpoint_pretrained", 2, 360, 1176, "FP32", 4)
    assert name == "superpoint_pretrained_2_360_1176_FP32_seq_4.csv"                 # node.cpp:285-296
    got = open(os.path.join(tmp_path, name)).read().splitlines(keepends=True)
    assert got == [vo_io.latency_row(*r) for r in rows]
