"""Per-op timing of an INT8 engine (device-calibrated on the timing input)."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import capi, quant, weights
capi.tuning_from_env()
if os.environ.get('PEROP_LIB'): capi.LIB_PATH = os.environ['PEROP_LIB']   # a variant build (make BUILD=... OUT=variants/x EXTRA=-D...)
graph = sys.argv[1] if len(sys.argv) > 1 else "mbv1"
H, Wd = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "360x1176").split("x"))
plan = weights.vgg_plan() if graph == "vgg" else weights.load(os.path.join(ROOT, "tests", "golden", f"sp_{graph}.spvw"))
x = np.random.RandomState(0).rand(2, 1, H, Wd).astype(np.float32)
plan.act_scales = quant.calibrate(plan, [x], H, Wd)
p = os.path.join(tempfile.mkdtemp(), "w.spvw"); weights.save(plan, p, precision="INT8")
ctx = capi.Context(net_height=H, net_width=Wd); ctx.load_weights(p)
for _ in range(20): ctx.forward(x)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(50): ctx.forward(x)
for k, v in ctx.profile().items():
    if v["calls"]:
        ms = v["total_ms"] / v["calls"]
        print(f"{k:12s} {ms*1e3:8.1f} us  {(v['flops'] / ms / 1e9 if v['flops'] else 0):7.1f} TOP/s")
