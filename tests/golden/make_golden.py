"""Generates the committed fixtures under tests/golden/ (run in the build container only).

  * sp_squeeze.spvw, sp_mbv1.spvw, sp_mbv2.spvw
                        -- the reference's src/odml_visual_odometry/models/sp_{squeeze,mbv1,mbv2}_b1.onnx
                           (real trained weights: DATA, not source) re-packed by spvo/weights.py
  * images/*.png        -- three of the reference's sample frames (src/odml_visual_odometry/sample_images)
  * oracle_*.npz        -- outputs of the oracle on small inputs; they pin the oracle against
                           regressions (the reference itself holds no golden vectors: SURVEY.md 4, 8c)

Usage: python tests/golden/make_golden.py [--weights-only]   (needs /root/reference)
"""
import os
import shutil
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402,F401  (sets the package path)
from oracle import frontend as fe, matching, net, odometry as od  # noqa: E402
from spvo import synth, weights  # noqa: E402

REF = "/root/reference/src/odml_visual_odometry"


def main():
    os.makedirs(os.path.join(HERE, "images"), exist_ok=True)
    for i in range(3):
        src = f"{REF}/sample_images/{i:010d}.png"
        dst = os.path.join(HERE, "images", f"{i:010d}.png")
        if not os.path.exists(dst):
            shutil.copyfile(src, dst)
    for name in ("mbv1", "mbv2"):
        weights.save(weights.onnx_plan(f"{REF}/models/sp_{name}_b1.onnx"), os.path.join(HERE, f"sp_{name}.spvw"))
    plan = weights.onnx_plan(f"{REF}/models/sp_squeeze_b1.onnx")
    weights.save(plan, os.path.join(HERE, "sp_squeeze.spvw"))
    if "--weights-only" in sys.argv:
        return

    from PIL import Image
    img = np.asarray(Image.open(os.path.join(HERE, "images", "0000000000.png")))
    P_l, P_r = synth.projection_matrices()

    # --- front end at the reference's smallest net size (120 x 392), real squeeze weights
    r = fe.detect(plan, img, P_l, 120, 392)
    np.savez_compressed(os.path.join(HERE, "oracle_frontend_squeeze_120x392.npz"),
                        resized=r["resized"], P=r["P"], det=r["det"].astype(np.float16),
                        heat_sum=np.float64(r["heat"].astype(np.float64).sum()),
                        xy=r["xy"], desc_head=r["descriptors"][:16])
    # --- VGG (seeded synthetic weights) on a 64 x 64 crop: per-tensor checksums
    vgg = weights.vgg_plan(seed=0)
    x = (img[100:164, 300:364].astype(np.float32) / 255.0)[None, None]
    det, desc, vals = net.forward(vgg, x, return_all=True)
    np.savez_compressed(os.path.join(HERE, "oracle_vgg_64x64.npz"), x=x, det=det, desc=desc,
                        sums=np.array([float(np.abs(vals[k]).astype(np.float64).sum()) for k in sorted(vals)]))
    # --- matcher on seeded descriptors
    rng = np.random.RandomState(7)
    a = rng.randn(60, 256).astype(np.float32)
    b = np.concatenate([a[:40] + 0.05 * rng.randn(40, 256).astype(np.float32), rng.randn(30, 256).astype(np.float32)])
    a /= np.linalg.norm(a, axis=1, keepdims=True)
    b /= np.linalg.norm(b, axis=1, keepdims=True)
    knn_idx, knn_d = matching.bf_match(a, b, "KNN", False, 0.8)
    nn_idx, nn_d = matching.bf_match(a, b, "NN", True, 0.8)
    np.savez_compressed(os.path.join(HERE, "oracle_match.npz"), a=a, b=b, knn_idx=knn_idx, knn_d=knn_d,
                        nn_idx=nn_idx, nn_d=nn_d)
    # --- odometry on seeded synthetic correspondences
    rng = np.random.RandomState(3)
    n = 200
    Xc = np.stack([rng.uniform(-10, 10, n), rng.uniform(-2, 2, n), rng.uniform(5, 40, n)], 1)
    rv, tv = np.array([0.01, -0.02, 0.005]), np.array([0.05, -0.02, 1.0])
    R = od.quat_to_rot(od.rvec_to_quat(rv))
    Xp = Xc @ R.T + tv

    def proj(P, X):
        p = X @ P[:, :3].T + P[:, 3]
        return (p[:, :2] / p[:, 2:]).astype(np.float32)

    cl, cr, pl, pr = [proj(P, X) + (0.3 * rng.randn(n, 2)).astype(np.float32)
                      for P, X in ((P_l, Xc), (P_r, Xc), (P_l, Xp), (P_r, Xp))]
    bad = rng.rand(n) < 0.2
    pl[bad] += rng.uniform(-50, 50, (bad.sum(), 2)).astype(np.float32)
    pts = od.triangulate(P_l, P_r, cl, cr)
    ok, r_est, t_est, inl = od.pnp_ransac(P_l[:, :3], pts, pl, np.zeros(3), np.zeros(3), 500, 2.0, 0)
    X, uv, cam, inv = [], [], [], []
    for i in inl:
        X += [pts[i], pts[i]]
        uv += [pl[i], pr[i]]
        cam += [0, 1]
        inv += [0, 0]
    obs = (np.asarray(X, np.float64), np.asarray(uv, np.float64), np.asarray(cam), np.asarray(inv))
    q, t, s = od.pnp_refine(P_l, P_r, obs, od.rvec_to_quat(r_est), t_est)
    np.savez_compressed(os.path.join(HERE, "oracle_odometry.npz"), P_l=P_l, P_r=P_r, cl=cl, cr=cr, pl=pl, pr=pr,
                        pts=pts, ok=ok, rvec=r_est, tvec=t_est, inliers=inl, q=q, t=t,
                        iterations=s.iterations, converged=s.converged, true_rvec=rv, true_tvec=tv)
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
