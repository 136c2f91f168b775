"""round 6: how many launch segments of a config-3 loop go out as graph replays, and what the step costs the host with and without them"""
import os, sys, tempfile, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8"); os.environ.setdefault("SPVO_QUIET", "1")
import numpy as np, torch
torch.cuda.init()
from spvo import capi, host, synth, weights
capi.tuning_from_env()
capi.set_tuning("tail_streams", 2)
H, W = 192, 640
plan = weights.vgg_plan(seed=0); plan.precision = "FP16"
tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "laptop"))
weights.save(plan, os.path.join(tmp, "laptop", weights.engine_name("superpoint_pretrained", 2, H, W, "FP16")))
frames, _, P_l, P_r = synth.stereo_sequence(8, os.path.join(ROOT, "tests", "golden", "images", "0000000000.png"), seed=0)
order = list(range(8)) + list(range(6, 0, -1))
d = [(torch.from_numpy(frames[f][0]).cuda().clone(), torch.from_numpy(frames[f][1]).cuda().clone()) for f in order]
fe = host.FrontEnd(tmp, prefix="superpoint_pretrained", selector="KNN", cross_check=True, batch=2, height=H, width=W, precision="FP16")
assert fe.engine_loaded, fe.last_error
pl, pr = [a.data_ptr() for a, _ in d], [b.data_ptr() for _, b in d]
rows, cols, stride = frames[0][0].shape[0], frames[0][0].shape[1], d[0][0].stride(0)
fe.run_device_block(pl, pr, rows, cols, stride, P_l, P_r, 0, 64)
torch.cuda.synchronize()
t0 = time.perf_counter(); n = 0
for b in range(20):
    fe.run_device_block(pl, pr, rows, cols, stride, P_l, P_r, 64 + 64 * b, 64); n += 64
torch.cuda.synchronize()
dt = time.perf_counter() - t0
p = fe.context().profile()
print("graphs tuning", capi.get_tuning("graphs", 1), ": %.1f frames/s, %.1f us per frame; segments as graph replays %d, as plain launches %d" %
      (n / dt, 1e6 * dt / n, p.get("segment_graph_launch", {}).get("calls", 0), p.get("segment_plain_launch", {}).get("calls", 0)))
fe.close()
