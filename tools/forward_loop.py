"""Forward-only loop of one engine (both images of a pair per pass, layers back to back, nothing else on the chip): the program the
per-layer PMC passes run (tools/collect_profiles_r04.sh -> tools/pmc_layers.py).
usage: forward_loop.py [vgg|squeeze|mbv1|mbv2] [FP32|FP16|INT8] [HxW] [passes = 30]"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import capi, weights
capi.set_tuning("heads_keep_raw", 0)   # the fused heads as a detector submission runs them
graph = sys.argv[1] if len(sys.argv) > 1 else "vgg"
prec = sys.argv[2] if len(sys.argv) > 2 else "FP32"
H, Wd = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "360x1176").split("x"))
n = int(sys.argv[4]) if len(sys.argv) > 4 else 30
plan = weights.vgg_plan() if graph == "vgg" else weights.load(os.path.join(ROOT, "tests", "golden", f"sp_{graph}.spvw"))
x = np.random.RandomState(0).rand(2, 1, H, Wd).astype(np.float32)
p = os.path.join(tempfile.mkdtemp(), "w.spvw")
if prec == "INT8":      # calibrated on the timing input, as bench.py's config 5 does
    from spvo import quant
    plan.act_scales = quant.calibrate(plan, [x], H, Wd)
    weights.save(plan, p, precision="INT8")
else:
    plan.precision = prec
    weights.save(plan, p)
ctx = capi.Context(net_height=H, net_width=Wd); ctx.load_weights(p)
for _ in range(n): ctx.forward(x)
ctx.close()
print("passes", n)
