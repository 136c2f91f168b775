"""The unchanged node's call sequence, one pair at a time on HOST images (addStereoImagePair(cv::Mat&, ...), matchDescriptors x 2,
solveStereoOdometry): where a frame's 1.4 ms go.  Host wall time per call, and the device stages of the same frames (HIP events).
python tools/sync_breakdown.py [sp_squeeze]"""
import os, sys, tempfile, time, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd")):
    sys.path.insert(0, p)
import numpy as np
os.environ.setdefault("SPVO_QUIET", "1")
from spvo import host, synth, weights, capi
capi.tuning_from_env()   # SPVO_TUNE_<NAME>=<int>: diagnostic switches for A/B runs

trained = len(sys.argv) > 1 and sys.argv[1] == "sp_squeeze"
tmp = tempfile.mkdtemp(); os.makedirs(os.path.join(tmp, "laptop"))
prefix = "sp_squeeze" if trained else "superpoint_pretrained"
if trained:
    shutil.copyfile(os.path.join(ROOT, "tests", "golden", "sp_squeeze.spvw"), os.path.join(tmp, "laptop", weights.engine_name(prefix, 2, 360, 1176, "FP32")))
else:
    weights.save(weights.vgg_plan(seed=0), os.path.join(tmp, "laptop", weights.engine_name(prefix, 2, 360, 1176, "FP32")))
frames, _, P_l, P_r = synth.stereo_sequence(8, os.path.join(ROOT, "tests", "golden", "images", "0000000000.png"), seed=0)
fe = host.FrontEnd(tmp, prefix=prefix)
assert fe.engine_loaded, fe.last_error
mats = [(fe.make_image(L), fe.make_image(R)) for L, R in frames]
order = list(range(8)) + list(range(6, 0, -1))
Pl = np.ascontiguousarray(P_l, np.float64); Pr = np.ascontiguousarray(P_r, np.float64)
import ctypes as C
ctx = fe.context()
acc = np.zeros(4); N = 300
for i in range(N + 30):
    if i == 30:
        ctx.profile_enable(True); ctx.profile_reset()
    m = mats[order[i % len(order)]]
    t0 = time.perf_counter()
    fe.lib.spvo_host_add_stereo_pair_mat(fe.h, C.c_void_p(m[0]), C.c_void_p(m[1]), Pl.ctypes.data_as(C.c_void_p), Pr.ctypes.data_as(C.c_void_p))
    t1 = time.perf_counter()
    fe.match_descriptors(host.CURR_LEFT_CURR_RIGHT)
    if fe.dq_size() >= 4:
        fe.match_descriptors(host.CURR_LEFT_PREV_LEFT)
    t2 = time.perf_counter()
    if fe.dq_size() >= 4:
        fe.solve_stereo_odometry()
    t3 = time.perf_counter()
    if i >= 30:
        acc += [t1 - t0, t2 - t1, t3 - t2, t3 - t0]
prof = ctx.profile(); ctx.profile_enable(False)
print(f"{'sp_squeeze' if trained else 'vgg (seeded)'}: host wall per frame [ms]: addStereoImagePair {acc[0]/N*1e3:.3f}  matchDescriptors x2 {acc[1]/N*1e3:.3f}  solveStereoOdometry {acc[2]/N*1e3:.3f}  total {acc[3]/N*1e3:.3f}  (with every stage timed: slower than the bench's synchronous leg)")
print("device stages [us]:", {k: round(v["total_ms"] / max(v["calls"], 1) * 1e3, 1) for k, v in prof.items() if v["calls"] and not k.startswith(("conv:", "pool:", "l2norm"))})
print("conv layers [us]:", {k: round(v["total_ms"] / max(v["calls"], 1) * 1e3, 1) for k, v in prof.items() if v["calls"] and k.startswith("conv:")})
fe.close()
