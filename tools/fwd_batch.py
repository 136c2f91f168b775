"""Forward pass of the FP32 VGG engine on two and on four images per launch (spvo_set_trunk_pairing runs two stereo pairs through every
layer in one launch): per-layer HIP-event times.  python tools/fwd_batch.py"""
import os, sys, tempfile
ROOT = "/root/repo"
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import capi, weights
plan = weights.vgg_plan()
p = os.path.join(tempfile.mkdtemp(), "w.spvw"); weights.save(plan, p)
for B in (2, 4):
    capi.clear_tuning()
    # the same engine (kernels selected for two images at load) run on two and on four images per launch: what trunk pairing saves
    ctx = capi.Context(net_height=360, net_width=1176, max_batch=2); ctx.load_weights(p)
    fams = [ctx.stage_kernel(f"conv:{i}")[0] for i in range(1, 9)]
    x = np.random.RandomState(0).rand(B, 1, 360, 1176).astype(np.float32)
    for _ in range(20): ctx.forward(x)
    ctx.profile_enable(True); ctx.profile_reset()
    for _ in range(100): ctx.forward(x)
    prof = ctx.profile()
    net = prof["net"]["total_ms"] / prof["net"]["calls"] * 1e3
    layers = {k: round(v["total_ms"] / v["calls"] * 1e3, 1) for k, v in prof.items() if v["calls"] and (k.startswith("conv:") or k == "heads")}
    print(f"batch {B}: forward {net:.1f} us = {net / (B // 2):.1f} us per pair; sum of layers {sum(layers.values()):.1f}; {fams}\n   {layers}")
    ctx.close()
