"""profiles/rNN_layer_roofline.json: every layer of the FP32 VGG engine at the headline size -- kernel, duration (HIP events on the
context's stream, both images of a stereo pair per launch, layers run back to back alone on the chip), algorithmic GFLOP (direct
convolution), GFLOP EXECUTED on the matrix pipe (Winograd F(2x2,3x3) layers: 4/9 of the algorithmic count)
and the fraction of the 157.3 TFLOP/s fp32-MFMA peak the executed flops amount to; HBM-bound layers carry their GB/s instead.

usage: python tools/layer_roofline_json.py out.json [HxW] [FP32|FP16] [images per launch = 2; 4: two stereo pairs per launch, as under trunk pairing]

FP16 (BASELINE config 3): the same table for the FP16 engine against the 2.5 PFLOP/s dense fp16 peak; every row also carries the roofline that
bounds it (`bound`: the larger of executed flops / matrix peak and algorithmic bytes / 8 TB/s) -- the HBM <-> MFMA crossover per layer."""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import capi, weights
capi.tuning_from_env()   # SPVO_TUNE_WINOGRAD=0 etc.: this measurement script opts in (the library itself reads no environment variable)
capi.set_tuning("heads_keep_raw", 0)   # time the fused heads as a detector submission runs them (spvo_forward would also write the un-normalised planes it exposes)

out_path = sys.argv[1]
H, Wd = (int(v) for v in (sys.argv[2] if len(sys.argv) > 2 else "360x1176").split("x"))
prec = sys.argv[3] if len(sys.argv) > 3 else "FP32"
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
plan = weights.vgg_plan()
plan.precision = prec
p = os.path.join(tempfile.mkdtemp(), "w.spvw"); weights.save(plan, p)
ctx = capi.Context(net_height=H, net_width=Wd); ctx.load_weights(p)
x = np.random.RandomState(0).rand(B, 1, H, Wd).astype(np.float32)
for _ in range(30): ctx.forward(x)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(200): ctx.forward(x)
prof = ctx.profile()
PEAK, HBM = (157.3 if prec == "FP32" else 2500.0), 8000.0
names = ["conv1a", "conv1b", "conv2a", "conv2b", "conv3a", "conv3b", "conv4a", "conv4b", "convPa", "convDa", "convPb", "convDb", "l2norm"]
rows, tot_us, tot_exec, tot_alg = [], 0.0, 0.0, 0.0
i = 0
while i < len(plan.ops):
    op = plan.ops[i]
    key = {weights.OP_CONV: "conv", weights.OP_L2NORM: "l2norm"}.get(op.type, "op") + f":{i}"
    st = prof.get(key)
    label, ops = names[i], [op]
    if key == "conv:10" and (not st or not st["calls"]) and prof.get("heads", {}).get("calls"):      # convPb + convDb + l2norm: one launch
        st, label, ops, step = prof["heads"], "convPb+convDb+l2norm (heads.hip.h)", plan.ops[10:13], 3
    elif i + 1 < len(plan.ops) and plan.ops[i + 1].type == weights.OP_CONV and op.type == weights.OP_CONV and plan.ops[i + 1].inp == op.inp \
            and plan.ops[i + 1].out == op.out and not (prof.get(f"conv:{i + 1}") or {}).get("calls"):                # sibling layers in one launch
        label, ops, step = names[i] + "+" + names[i + 1], plan.ops[i:i + 2], 2
    else:
        step = 1
    if not st or not st["calls"]:
        i += step
        continue
    us = st["total_ms"] / st["calls"] * 1e3
    alg = byts = 0.0
    for o in ops:
        if o.type != weights.OP_CONV:
            continue
        lvl_in, lvl_out = plan.tensors[o.inp][1], plan.tensors[o.out][1]
        px_in, px_out = (H >> lvl_in) * (Wd >> lvl_in), (H >> lvl_out) * (Wd >> lvl_out)
        alg += 2.0 * B * px_in * o.cout * o.cin * o.ksize * o.ksize
        byts += B * 4 * (px_in * o.cin + px_out * o.cout) + o.weight.size * 4
    if st.get("bytes", 0) > 0:      # the library's own statement of the launch's algorithmic bytes (element sizes of the engine's tensors)
        byts = st["bytes"]
    o0 = ops[0]
    kfam, kfactor = ctx.stage_kernel(key) if o0.type == weights.OP_CONV and "heads" not in label else ("", 1.0)   # what the library runs the layer on
    wino = kfam.startswith("conv_wino")
    executed = alg * kfactor
    hbm_bound = o0.type == weights.OP_CONV and o0.cin == 1
    row = {"layer": label, "shape": f"{o0.cin}->{sum(o.cout for o in ops if o.type == weights.OP_CONV)} k{o0.ksize} @{H >> plan.tensors[o0.inp][1]}x{Wd >> plan.tensors[o0.inp][1]}",
           "kernel": "conv_first4_kernel (VALU, HBM-write bound)" if hbm_bound else (kfam + (" (Winograd F(4x4,3x3), fp32 MFMA: executes 1/4 of the direct method's multiplies)" if kfactor == 0.25 else " (Winograd F(2x2,3x3), fp32 MFMA: executes 4/9)")) if wino else
                     "heads_fused_kernel (fp32 MFMA)" if "heads" in label else "l2norm_nhwc_kernel (VALU, HBM)" if o0.type == weights.OP_L2NORM else (kfam + " (direct, fp16 MFMA, fp32 accumulate)" if prec == "FP16" else "conv_mfma_kernel (direct, fp32 MFMA)"),
           "duration_us": round(us, 2), "algorithmic_gflop": round(alg / 1e9, 3), "executed_gflop": round(executed / 1e9, 3),
           "executed_tflops": round(executed / us / 1e6, 2), "frac_of_mfma_peak": round(executed / us / 1e6 / PEAK, 4),
           "algorithmic_tflops": round(alg / us / 1e6, 2), "algorithmic_frac_of_mfma_peak": round(alg / us / 1e6 / PEAK, 4),
           "algorithmic_MB": round(byts / 1e6, 1), "algorithmic_GBps": round(byts / us / 1e3, 0)}
    t_mfma, t_hbm = executed / (PEAK * 1e12), byts / (HBM * 1e9)
    row["bound"] = "hbm" if (hbm_bound or t_hbm > t_mfma) else "mfma"
    row["frac_of_hbm_peak"] = round(byts / us / 1e3 / HBM, 4)
    row["frac_of_bound"] = round(max(t_mfma, t_hbm) / (us * 1e-6), 4) if not hbm_bound else row["frac_of_hbm_peak"]
    row["arithmetic_intensity_flop_per_byte"] = round(executed / max(byts, 1.0), 1)
    rows.append(row)
    tot_us += us; tot_exec += executed; tot_alg += alg
    i += step
res = {"_how": f"tools/layer_roofline_json.py on one MI355X: VGG SuperPoint {prec.lower()}, net {H}x{Wd}, {B} images per launch, 200 forward passes with every layer "
               "bracketed by HIP events (spvo_profile_*); layers run back to back, nothing else on the chip",
       "precision": prec, "peak_mfma_tflops": PEAK, "ridge_flop_per_byte": round(PEAK * 1e12 / (HBM * 1e9), 1), "layers": rows,
       "conv_stack": {"sum_of_layers_us": round(tot_us, 1), "algorithmic_gflop": round(tot_alg / 1e9, 2), "executed_gflop": round(tot_exec / 1e9, 2),
                      "executed_tflops": round(tot_exec / tot_us / 1e6, 2), "frac_of_mfma_peak": round(tot_exec / tot_us / 1e6 / PEAK, 4),
                      "algorithmic_tflops": round(tot_alg / tot_us / 1e6, 2), "algorithmic_frac_of_mfma_peak": round(tot_alg / tot_us / 1e6 / PEAK, 4)},
       "forward_pass_us": round(prof["net"]["total_ms"] / prof["net"]["calls"] * 1e3, 1)}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res, indent=1))
