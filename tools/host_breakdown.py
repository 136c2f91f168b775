"""Where the host thread spends a step (depth-2 pipeline): wait for the oldest detector submission, hand-over of the next
pairs, the two matches, the solve.  python tools/host_breakdown.py [FP32|FP16] [HxW] [split]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import torch  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "FP32"
H, W = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "360x1176").split("x"))
split = len(sys.argv) > 3 and sys.argv[3] == "split"
os.environ.setdefault("SPVO_QUIET", "1")
torch.cuda.init()
from spvo import host, synth, weights  # noqa: E402

plan = weights.vgg_plan(seed=0)
plan.precision = prec
tmp = tempfile.mkdtemp()
os.makedirs(os.path.join(tmp, "laptop"))
weights.save(plan, os.path.join(tmp, "laptop", weights.engine_name("superpoint_pretrained", 2, H, W, prec)))
cache = os.path.join(tempfile.gettempdir(), "spvo_hb_cache.npz")
if os.environ.get("HB_CACHE") == "1" and os.path.exists(cache):   # diagnostic: skip the seconds of rendering (the frame rate depends on it)
    z = np.load(cache)
    frames = [(z["L"][k], z["R"][k]) for k in range(8)]
    P_l, P_r = z["P_l"], z["P_r"]
else:
    frames, poses, P_l, P_r = synth.stereo_sequence(8, os.path.join(ROOT, "tests", "golden", "images", "0000000000.png"), seed=0)
    np.savez(cache, L=np.stack([f[0] for f in frames]), R=np.stack([f[1] for f in frames]), P_l=P_l, P_r=P_r)
d = [(torch.from_numpy(L).cuda(), torch.from_numpy(R).cuda()) for L, R in frames]
rows, cols = frames[0][0].shape
order = list(range(8)) + list(range(6, 0, -1))
fe = host.FrontEnd(tmp, prefix="superpoint_pretrained", height=H, width=W, precision=prec)
assert fe.engine_loaded, fe.last_error
if os.environ.get("HB_NOGC") == "1":
    import gc
    gc.disable()
import threading
print("python threads:", [t.name for t in threading.enumerate()])
acc = np.zeros(5)
N = 400
for i in range(N + 20):
    dl, dr = d[order[i % len(order)]]
    t0 = time.perf_counter()
    fe.add_stereo_image_pair_device(dl.data_ptr(), dr.data_ptr(), rows, cols, dl.stride(0), P_l, P_r)
    t1 = time.perf_counter()
    for k in (1, 2):
        nl, nr = d[order[(i + k) % len(order)]]
        fe.prefetch_device(nl.data_ptr(), nr.data_ptr(), rows, cols, dl.stride(0))
    t2 = time.perf_counter()
    fe.match_descriptors(host.CURR_LEFT_CURR_RIGHT)
    if fe.dq_size() >= 4:
        fe.match_descriptors(host.CURR_LEFT_PREV_LEFT)
    t3 = time.perf_counter()
    if fe.dq_size() >= 4:
        fe.solve_stereo_odometry()
    t4 = time.perf_counter()
    if i >= 20:
        acc += [t1 - t0, t2 - t1, t3 - t2, t4 - t3, t4 - t0]
print(f"{prec} {H}x{W} {'split' if split else ''}: per step [ms] add/wait {acc[0]/N*1e3:.3f}  hand-over {acc[1]/N*1e3:.3f}  "
      f"matches {acc[2]/N*1e3:.3f}  solve {acc[3]/N*1e3:.3f}  total {acc[4]/N*1e3:.3f}")
fe.close()
