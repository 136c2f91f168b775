"""Round 6: where do the GPU's and the oracle's PnP-RANSAC inlier sets part on a long sequence?  Three implementations on identical inputs
(frame by frame: the join of the oracle state machine fed with the GPU's features): libspvo (spvo_pnp_ransac), oracle/odometry.py, oracle/cpu."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
import torch
torch.cuda.init()
import oracle  # noqa
from oracle import cpu_backend, frontend as ofe, odometry as od
from spvo import capi, host, synth, weights
import shutil, tempfile

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10
drop = (5, 6, 15, 16)
frames, poses, P_l, P_r = synth.stereo_sequence(44, os.path.join(ROOT, "tests", "golden", "images", "0000000000.png"), seed=0, drop=drop)
d = tempfile.mkdtemp(); os.makedirs(os.path.join(d, "laptop"))
shutil.copyfile(os.path.join(ROOT, "tests", "golden", "sp_squeeze.spvw"), os.path.join(d, "laptop", weights.engine_name("sp_squeeze", 2, 360, 1176, "FP32")))
fe = host.FrontEnd(d, prefix="sp_squeeze", selector="KNN", cross_check=True)
ctx = fe.context()
cpu = cpu_backend.CpuBackend(net_height=360, net_width=1176)
st = od.FrontEndState()
_, Pl2 = ofe.preprocess(frames[0][0], P_l, 360, 1176, True)
_, Pr2 = ofe.preprocess(frames[0][1], P_r, 360, 1176, True)
for k in range(n):
    L, R = frames[k]
    res = fe.step(L, R, P_l, P_r)
    od.add_features(st, fe.keypoints(-2), fe.descriptors(-2), fe.keypoints(-1), fe.descriptors(-1), Pl2, Pr2)
    od.match_descriptors(st, 0, "KNN", False)
    if k == 0:
        continue
    od.match_descriptors(st, 1, "KNN", False)
    j = od.join(st, 2.0, 0.25, 4)
    pts3d = od.triangulate(st.P_l, st.P_r, j["cl"], j["cr"])
    K = st.P_l[:, :3].copy()
    r0, t0 = st.r_pred.copy(), st.t_pred.copy()
    ok_o, rv_o, tv_o, in_o = od.pnp_ransac(K, pts3d, j["pl"], r0, t0, 500, 2.0, 0)
    ok_g, rv_g, tv_g, in_g = ctx.pnp_ransac(K, pts3d, j["pl"], r0, t0, 500, 2.0, 0)
    ok_c, rv_c, tv_c, in_c = cpu.pnp_ransac(K, pts3d, j["pl"], r0, t0, 500, 2.0, 0)
    host_in = fe.inliers("pnp")
    pts_g = ctx.triangulate(st.P_l, st.P_r, j["cl"], j["cr"])
    dif = np.nonzero((pts_g.view(np.int32) != np.asarray(pts3d, np.float32).view(np.int32)).any(axis=1))[0]
    rel = np.abs(pts_g - pts3d) / np.maximum(np.abs(pts3d), 1e-3)
    ok_o2, rv_o2, tv_o2, in_o2 = od.pnp_ransac(K, pts_g, j["pl"], r0, t0, 500, 2.0, 0)
    ok_g2, rv_g2, tv_g2, in_g2 = ctx.pnp_ransac(K, pts_g, j["pl"], r0, t0, 500, 2.0, 0)
    print(f"   triangulation: {len(dif)} of {len(pts3d)} points differ in f32 bits, max rel {rel.max():.2e} (point {int(rel.max(axis=1).argmax())}, Z {pts3d[int(rel.max(axis=1).argmax())][2]:.1f}); "
          f"on the GPU's points: oracle {len(in_o2)} gpu {len(in_g2)} inliers, equal {np.array_equal(in_o2, in_g2)}, host==these {np.array_equal(host_in, in_g2)}; sym diff vs oracle points {sorted(set(in_o2.tolist()) ^ set(in_o.tolist()))}")
    if len(dif):
        i = int(rel.max(axis=1).argmax())
        print("      worst point", i, "oracle", pts3d[i], "gpu", pts_g[i], "cl", j["cl"][i], "cr", j["cr"][i])
    print(f"frame {k}: n {len(pts3d)} prior t {np.round(t0, 4)}  inliers oracle {len(in_o)} gpu {len(in_g)} cpu {len(in_c)} host-class {len(host_in)}  "
          f"gpu==oracle {np.array_equal(in_g, in_o)} cpu==oracle {np.array_equal(in_c, in_o)} host==gpu {np.array_equal(host_in, in_g)}  "
          f"|dt| gpu-oracle {np.abs(tv_g - tv_o).max():.2e} cpu-oracle {np.abs(tv_c - tv_o).max():.2e}")
    if not np.array_equal(in_g, in_o) or not np.array_equal(in_c, in_o):
        # per-hypothesis: which one wins where, and how close the disputed points sit to the threshold
        X = np.asarray(pts3d, np.float32).astype(np.float64); uv = np.asarray(j["pl"], np.float32).astype(np.float64)
        q0 = od.rvec_to_quat(r0)
        counts = []
        for it in range(500):
            s = od.sample_triplet(0, it, len(X))
            ok, q, t = od.minimal_solve(K, X[s], uv[s], q0, t0)
            counts.append(int(od.reproj_inliers(K, q, t, X, uv, 2.0).sum()) if ok else -1)
        counts = np.array(counts)
        best = int(np.argmax(counts))
        print("   oracle: best hypothesis", best, "count", counts[best], "; hypotheses with the same count:", np.nonzero(counts == counts[best])[0][:8], " runner-up", np.sort(counts)[-2])
        for name, (rv, tv, inl) in dict(oracle=(rv_o, tv_o, in_o), gpu=(rv_g, tv_g, in_g), cpu=(rv_c, tv_c, in_c)).items():
            q = od.rvec_to_quat(rv)
            Rm = od.quat_to_rot(q); p = (X @ Rm.T + tv) @ K.T
            e2 = (p[:, 0] / p[:, 2] - uv[:, 0]) ** 2 + (p[:, 1] / p[:, 2] - uv[:, 1]) ** 2
            print(f"   {name}: rvec {rv} tvec {tv}  refit-pose inlier count {(e2 <= 4).sum()}")
        sym = sorted(set(in_g.tolist()) ^ set(in_o.tolist()))
        print("   disputed (gpu vs oracle):", sym)
        s = od.sample_triplet(0, best, len(X)); ok, q, t = od.minimal_solve(K, X[s], uv[s], q0, t0)
        Rm = od.quat_to_rot(q); p = (X @ Rm.T + t) @ K.T
        e2 = (p[:, 0] / p[:, 2] - uv[:, 0]) ** 2 + (p[:, 1] / p[:, 2] - uv[:, 1]) ** 2
        for i in sym:
            print(f"     point {i}: e^2 under the oracle's best hypothesis {e2[i]:.12f} (threshold 4)")
    od.solve_stereo_odometry(st)
