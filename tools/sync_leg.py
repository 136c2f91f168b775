"""The bench's `host_interface.synchronous` leg on its own (addStereoImagePair(cv::Mat&, ...) on host images, one pair at a time,
matchDescriptors x 2, solveStereoOdometry), with the host wall time of each call.  python tools/sync_leg.py [steps = 300] [lookahead depth = 0] [depth of a first, untimed leg]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

os.environ.setdefault("SPVO_QUIET", "1")
torch.cuda.init()
from spvo import host, synth, weights  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
depth = int(sys.argv[2]) if len(sys.argv) > 2 else 0
H, W = 360, 1176
plan = weights.vgg_plan(seed=0)
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "laptop"))
weights.save(plan, os.path.join(tmp, "laptop", weights.engine_name("superpoint_pretrained", 2, H, W, "FP32")))
frames, poses, P_l, P_r = synth.stereo_sequence(8, os.path.join(ROOT, "tests", "golden", "images", "0000000000.png"), seed=0)
order = list(range(8)) + list(range(6, 0, -1))
fe = host.FrontEnd(tmp, prefix="superpoint_pretrained", selector="KNN", cross_check=True, batch=2, height=H, width=W, conf_thresh=0.015, dist_thresh=4,
                   border_remove=4, stereo_threshold=2.0, min_disparity=0.25, refinement_degree=4, precision="FP32")
assert fe.engine_loaded, fe.last_error
if len(sys.argv) > 4:   # 0 = copies inside addStereoImagePair, 1 = deferred behind the solve (default)
    fe.set_deferred_copies(int(sys.argv[4]))
mats = [(fe.make_image(L), fe.make_image(R)) for L, R in frames]
Pl, Pr = np.ascontiguousarray(P_l, np.float64), np.ascontiguousarray(P_r, np.float64)
import ctypes as C  # noqa: E402
import gc  # noqa: E402
gc.collect(); gc.freeze(); gc.disable()


def step(i, acc=None):
    m = mats[order[i % len(order)]]
    if depth == 0:
        t0 = time.perf_counter()
        fe.lib.spvo_host_add_stereo_pair_mat(fe.h, C.c_void_p(m[0]), C.c_void_p(m[1]), host._p(Pl), host._p(Pr))
        t1 = time.perf_counter()
        if fe.dq_size() < 4:
            fe.match_descriptors(host.CURR_LEFT_CURR_RIGHT)
            return
        fe.match_descriptors(host.CURR_LEFT_CURR_RIGHT)
        fe.match_descriptors(host.CURR_LEFT_PREV_LEFT)
        t2 = time.perf_counter()
        fe.solve_stereo_odometry()
        t3 = time.perf_counter()
        if acc is not None:
            acc += np.array([t1 - t0, t2 - t1, t3 - t2])
    else:
        a = [mats[order[(i + 1 + d) % len(order)]] if d < depth else None for d in range(3)]
        fe.step_host(m[0], m[1], P_l, P_r, a[0], a[1], deferred_solve=True, next3_pair=a[2])


if len(sys.argv) > 3:      # first run the OTHER mode in the same front end, as bench.py's legs follow each other
    d0, depth = depth, int(sys.argv[3])
    for i in range(200):
        step(i)
    fe.finish_solve()
    depth = d0
for i in range(30):
    step(i)
fe.finish_solve()
res = []
for rep in range(5):
    acc = np.zeros(3)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(30 + rep * steps, 30 + (rep + 1) * steps):
        step(i, acc)
    fe.finish_solve()
    torch.cuda.synchronize()
    res.append(((time.perf_counter() - t0) / steps, acc / steps))
res.sort(key=lambda r: r[0])
t, a = res[len(res) // 2]
print(f"depth {depth}: {1 / t:.1f} frames/s, {t * 1e3:.4f} ms per frame (median of 5 blocks of {steps}; min {res[0][0] * 1e3:.4f}, max {res[-1][0] * 1e3:.4f})"
      + (f"; addStereoImagePair {a[0] * 1e3:.4f}  matchDescriptors x2 {a[1] * 1e3:.4f}  solveStereoOdometry {a[2] * 1e3:.4f} ms" if depth == 0 else ""))
