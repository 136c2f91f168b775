"""Oracle for the caller-side result files ("next" row 2).  TEST INFRASTRUCTURE ONLY.

Restates
  * publishOdometry's pose integration            visual_odometry_node.cpp:100-148
  * visualOdomCallback's KITTI pose line          data_processing_node.cpp:144-188
  * the latency CSV row / file names              visual_odometry_node.cpp:246-258, 285-296,
                                                  data_processing_node.cpp:102-106
C++ `ostream << double` with default precision prints 6 significant digits ("%g").
"""
from __future__ import annotations

import numpy as np

from .odometry import quat_mul, quat_to_rot


def _mat(q, t):
    T = np.eye(4)
    q = np.asarray(q, float)
    T[:3, :3] = quat_to_rot(q / np.linalg.norm(q))
    T[:3, 3] = t
    return T


def integrate(rel_poses, base_T_cam0):
    """rel_poses: list of (q xyzw, t) = cam0_curr_T_cam0_prev.  Returns list of 4x4 world_T_base."""
    B = _mat(*base_T_cam0)
    Binv = np.linalg.inv(B)
    W = np.eye(4)
    last = np.eye(4)
    out = []
    for q, t in rel_poses:
        T = _mat(q, t)
        if np.linalg.norm(T[:3, 3]) > 10:                 # node.cpp:118-123
            T = last
        else:
            last = T
        W = W @ (B @ np.linalg.inv(T) @ Binv)             # node.cpp:125-127
        out.append(W.copy())
    return out


def kitti_lines(world_T_base, base_T_cam0, seq_start=0):
    B = _mat(*base_T_cam0)
    Binv = np.linalg.inv(B)
    lines = []
    start = None
    for k, W in enumerate(world_T_base):
        if k < seq_start:                                  # dp.cpp:145-148
            continue
        if start is None:
            start = W
        C = Binv @ (np.linalg.inv(start) @ W) @ B          # dp.cpp:159-178
        lines.append("".join("%g " % C[r, c] for r in range(3) for c in range(4)) + "\n")
    return lines


def kitti_file_name(kitti_eval_id: int) -> str:
    name = f"{kitti_eval_id}_pred.txt"
    return "0" + name if len(name) == 10 else name


def latency_file_name(prefix, batch, h, w, precision, kitti_eval_id):
    return f"{prefix}_{batch}_{h}_{w}_{precision}_seq_{kitti_eval_id}.csv"


def latency_row(t_detect, t_match, t_solve, t_total):
    return ",".join("%g" % np.float32(v) for v in (t_detect, t_match, t_solve, t_total)) + "\n"
