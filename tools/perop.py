"""Per-op timing of the network (HIP events on the context stream) for tile tuning."""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import capi, weights
if len(sys.argv) > 1:   # a committed fixture: squeeze | mbv1 | mbv2
    p = os.path.join(ROOT, "tests", "golden", f"sp_{sys.argv[1]}.spvw")
else:
    plan = weights.vgg_plan()
    p = os.path.join(tempfile.mkdtemp(), "w.spvw"); weights.save(plan, p)
ctx = capi.Context(); ctx.load_weights(p)
x = np.random.RandomState(0).rand(2, 1, 360, 1176).astype(np.float32)
for _ in range(20): ctx.forward(x)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(50): ctx.forward(x)
prof = ctx.profile()
tot = 0
for k, v in prof.items():
    if v["calls"]:
        ms = v["total_ms"] / v["calls"]
        tf = v["flops"] / ms / 1e9 if v["flops"] else 0
        print(f"{k:12s} {ms*1e3:8.1f} us  {tf:7.1f} TF")
