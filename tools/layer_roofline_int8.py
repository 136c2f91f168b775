"""profiles/rNN_layer_roofline_int8.json: every launch of an INT8 engine (BASELINE config 5: sp_mbv1) -- duration (HIP events on the context's
stream, both images of a stereo pair per launch, launches back to back alone on the chip), the integer operations and the algorithmic bytes
the library states for the launch (spvo_profile_*: element sizes of the engine's tensors; a fused depthwise + pointwise block counts both
layers' operations and only the block's input and output tensors), and the roofline that bounds it: int8 MFMA dense peak 5000 TOP/s
against HBM 8 TB/s (ridge 625 op/byte: every layer of this graph is HBM-bound).

usage: python tools/layer_roofline_int8.py out.json [mbv1|mbv2|vgg] [HxW = 360x1176] [images per launch = 2]"""
import json, os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import capi, quant, weights
capi.tuning_from_env()
capi.set_tuning("heads_keep_raw", 0)   # the fused blocks and heads as a detector submission runs them (spvo_forward would also store the tensors they skip)

out_path = sys.argv[1]
graph = sys.argv[2] if len(sys.argv) > 2 else "mbv1"
H, Wd = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "360x1176").split("x"))
B = int(sys.argv[4]) if len(sys.argv) > 4 else 2
plan = weights.vgg_plan() if graph == "vgg" else weights.load(os.path.join(ROOT, "tests", "golden", f"sp_{graph}.spvw"))
x = np.random.RandomState(0).rand(B, 1, H, Wd).astype(np.float32)
plan.act_scales = quant.calibrate(plan, [x[:2]], H, Wd)
p = os.path.join(tempfile.mkdtemp(), "w.spvw"); weights.save(plan, p, precision="INT8")
ctx = capi.Context(net_height=H, net_width=Wd); ctx.load_weights(p)
for _ in range(30): ctx.forward(x)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(200): ctx.forward(x)
prof = ctx.profile()
PEAK, HBM = 5000.0, 8000.0
rows, tot_us, tot_ops, tot_b = [], 0.0, 0.0, 0.0
order = list(range(len(plan.ops)))
if len(order) >= 5 and plan.ops[-5].ksize == 3 and plan.ops[-4].ksize == 1 and plan.ops[-3].ksize == 3 and plan.ops[-2].ksize == 1:   # the library moves convPb behind convDa (spvo_core.hip)
    order[-4], order[-3] = order[-3], order[-4]
for key, st in prof.items():
    if not st["calls"] or (":" not in key and key != "heads"):
        continue
    if key == "heads":   # convPb + convDb + L2 norm in one launch (heads_i8.hip.h)
        i = order[-3]
    else:
        i = order[int(key.split(":")[1])]
    op = plan.ops[i]
    us = st["total_ms"] / st["calls"] * 1e3
    ops_n, byts = st["flops"], st.get("bytes", 0.0)
    kfam = ctx.stage_kernel(key)[0] if key != "heads" else ""
    lvl = plan.tensors[op.inp][1]
    row = {"layer": key, "first_op": f"{getattr(op, 'cin', 0)}->{getattr(op, 'cout', 0)} k{getattr(op, 'ksize', 0)} @{H >> lvl}x{Wd >> lvl}", "shape": f"@{H >> lvl}x{Wd >> lvl}",
           "kernel": "heads_i8_kernel (convPb + convDb + L2 norm)" if key == "heads" else (kfam or key.split(":")[0]), "duration_us": round(us, 2), "algorithmic_gop": round(ops_n / 1e9, 3),
           "algorithmic_tops": round(ops_n / us / 1e6, 2), "frac_of_mfma_peak": round(ops_n / us / 1e6 / PEAK, 4),
           "algorithmic_MB": round(byts / 1e6, 2), "algorithmic_GBps": round(byts / us / 1e3, 0), "frac_of_hbm_peak": round(byts / us / 1e3 / HBM, 4)}
    t_mfma, t_hbm = ops_n / (PEAK * 1e12), byts / (HBM * 1e9)
    row["bound"] = "hbm" if t_hbm >= t_mfma else "mfma"
    row["frac_of_bound"] = round(max(t_mfma, t_hbm) / (us * 1e-6), 4)
    rows.append(row)
    tot_us += us; tot_ops += ops_n; tot_b += byts
res = {"_how": f"tools/layer_roofline_int8.py on one MI355X: sp_{graph} INT8 (calibrated on the timing input), net {H}x{Wd}, {B} images per launch, 200 forward passes with "
               "every launch bracketed by HIP events (spvo_profile_*); launches run back to back, nothing else on the chip",
       "precision": "INT8", "peak_mfma_tops": PEAK, "ridge_op_per_byte": PEAK * 1e12 / (HBM * 1e9), "layers": rows,
       "conv_stack": {"sum_of_layers_us": round(tot_us, 1), "launches": len(rows), "algorithmic_gop": round(tot_ops / 1e9, 2), "algorithmic_tops": round(tot_ops / tot_us / 1e6, 2),
                      "algorithmic_MB": round(tot_b / 1e6, 1), "algorithmic_GBps": round(tot_b / tot_us / 1e3, 0), "frac_of_hbm_peak": round(tot_b / tot_us / 1e3 / HBM, 4)},
       "forward_pass_us": round(prof["net"]["total_ms"] / prof["net"]["calls"] * 1e3, 1)}
json.dump(res, open(out_path, "w"), indent=1)
print(json.dumps(res["conv_stack"]), res["forward_pass_us"])
for r in rows:
    print(f'{r["layer"]:12s} {r["kernel"][:26]:28s} {r["duration_us"]:7.1f} us {r["algorithmic_MB"]:8.1f} MB {r["algorithmic_GBps"]:6.0f} GB/s  {r["bound"]} {r["frac_of_bound"]:.3f}')
