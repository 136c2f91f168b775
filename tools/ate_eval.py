"""Absolute trajectory error on a synthetic stereo sequence with known ego-motion (the reference only PLOTS
trajectories, VO/figures; KITTI itself is not available here).  Trained sp_squeeze weights, the reference's launch
parameters (KNN 0.8, refinement degree 4), 1241x376 frames rendered by spvo/synth.py.

Reports, for the FP32 and the FP16 engine: ATE RMSE / max of the integrated camera positions against ground truth,
and -- FP32 -- against the oracle's restatement of the state machine fed with the GPU's own keypoints, descriptors and
projection matrices (SURVEY.md section 8d parity gate: ATE(GPU trajectory, CPU trajectory) <= 1e-3 m).

usage: ate_eval.py [n_frames]
"""
import os, shutil, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
import oracle  # noqa: F401
from oracle import frontend as ofe, odometry as od
from spvo import host, synth, weights

n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
tex = os.path.join(ROOT, "tests", "golden", "images", "0000000000.png")
frames, poses, P_l, P_r = synth.stereo_sequence(n, tex, seed=0)
d = tempfile.mkdtemp()
os.makedirs(os.path.join(d, "laptop"))
plan = weights.load(os.path.join(ROOT, "tests", "golden", "sp_squeeze.spvw"))
for prec in ("FP32", "FP16"):
    weights.save(plan, os.path.join(d, "laptop", weights.engine_name("sp_squeeze", 2, 360, 1176, prec)), precision=prec)


def positions(rel):          # camera centre of every frame in the first camera's frame, from cam0_curr_T_cam0_prev steps
    T = np.eye(4)
    out = [np.zeros(3)]
    for q, t in rel:
        S = np.eye(4)
        S[:3, :3], S[:3, 3] = od.quat_to_rot(np.asarray(q)), t
        T = T @ np.linalg.inv(S)                               # world_T_curr = world_T_prev * prev_T_curr
        out.append(T[:3, 3].copy())
    return np.array(out)


gt_rel = [synth.relative_pose(poses[k - 1], poses[k]) for k in range(1, n)]
Tg = np.eye(4)
gt = [np.zeros(3)]
for R, t in gt_rel:
    S = np.eye(4)
    S[:3, :3], S[:3, 3] = R, t
    Tg = Tg @ np.linalg.inv(S)
    gt.append(Tg[:3, 3].copy())
gt = np.array(gt)
path_len = float(np.sum(np.linalg.norm(np.diff(gt, axis=0), axis=1)))


def ate(a, b):
    e = np.linalg.norm(a - b, axis=1)
    return float(np.sqrt(np.mean(e ** 2))), float(e.max())


print(f"{n} frames, path length {path_len:.1f} m, sp_squeeze (trained), net 360x1176")
traj = {}
for prec in ("FP32", "FP16"):
    fe = host.FrontEnd(d, prefix="sp_squeeze", selector="KNN", cross_check=True, precision=prec)
    assert fe.engine_loaded, fe.last_error
    st = od.FrontEndState()
    rel, rel_o = [], []
    for k, (L, R) in enumerate(frames):
        res = fe.step(L, R, P_l, P_r)
        if prec == "FP32":      # the oracle's state machine on identical upstream features
            _, Pl2 = ofe.preprocess(L, P_l, 360, 1176, True)
            _, Pr2 = ofe.preprocess(R, P_r, 360, 1176, True)
            od.add_features(st, fe.keypoints(host.CURR_LEFT), fe.descriptors(host.CURR_LEFT),
                            fe.keypoints(host.CURR_RIGHT), fe.descriptors(host.CURR_RIGHT), Pl2, Pr2)
            od.match_descriptors(st, 0, "KNN", False)
            if k:
                od.match_descriptors(st, 1, "KNN", False)
                oq, ot, _ = od.solve_stereo_odometry(st)
                rel_o.append((oq, ot))
        if res is not None:
            rel.append(res)
    fe.close()
    traj[prec] = positions(rel)
    r, m = ate(traj[prec], gt)
    print(f"GPU {prec}: ATE vs ground truth  rmse {r:.4f} m  max {m:.4f} m  ({100 * r / path_len:.3f} % of the path)")
    if prec == "FP32":
        r, m = ate(traj[prec], positions(rel_o))
        print(f"GPU FP32 vs oracle state machine on the same features: ATE rmse {r:.2e} m  max {m:.2e} m  (gate 1e-3 m)")
r, m = ate(traj["FP16"], traj["FP32"])
print(f"GPU FP16 vs GPU FP32: ATE rmse {r:.4f} m  max {m:.4f} m")
shutil.rmtree(d)
