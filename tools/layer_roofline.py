"""Per-layer roofline table of the network (BASELINE config 3: where the layers cross from HBM-bound to MFMA-bound).

For every op: algorithmic FLOPs and bytes (input tensor slice + weights + output tensor slice, each once, in the
engine's storage precision), arithmetic intensity, the measured time (HIP events on the context's stream, both images of a
stereo pair in one launch), achieved TFLOP/s and GB/s and the bound the roofline assigns (peaks from
/opt/skills/guides/MI355X_MICROARCH.md: HBM 8 TB/s, fp32 MFMA 157.3 TFLOP/s, fp16 MFMA 2500 TFLOP/s dense).

usage: layer_roofline.py [vgg|squeeze] [FP32|FP16] [HxW]
"""
import os, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "superpoint-stereo-visual-odometry_amd"))
import numpy as np
from spvo import capi, weights

graph = sys.argv[1] if len(sys.argv) > 1 else "vgg"
prec = sys.argv[2] if len(sys.argv) > 2 else "FP32"
H, Wd = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else "360x1176").split("x"))
plan = weights.vgg_plan() if graph == "vgg" else weights.load(os.path.join(ROOT, "tests", "golden", f"sp_{graph}.spvw"))
plan.precision = prec
p = os.path.join(tempfile.mkdtemp(), "w.spvw"); weights.save(plan, p)
ctx = capi.Context(net_height=H, net_width=Wd); ctx.load_weights(p)
x = np.random.RandomState(0).rand(2, 1, H, Wd).astype(np.float32)
for _ in range(30): ctx.forward(x)
ctx.profile_enable(True); ctx.profile_reset()
for _ in range(100): ctx.forward(x)
prof = ctx.profile()

HBM, PEAK = 8000.0, (157.3 if prec == "FP32" else 2500.0)          # GB/s, TFLOP/s
f32_tensors = {plan.input_tensor, plan.det_tensor, plan.desc_tensor} | {op.inp for op in plan.ops if op.type == weights.OP_L2NORM}
bpe = lambda t: 4 if (prec == "FP32" or t in f32_tensors) else 2
print(f"{graph} {prec} {H}x{Wd}, 2 images per launch; ridge = {PEAK * 1e3 / HBM:.0f} flop/byte")
print(f"{'op':10s} {'shape':22s} {'GFLOP':>8s} {'MB':>8s} {'flop/B':>7s} {'us':>8s} {'TFLOP/s':>8s} {'GB/s':>7s}  bound by roofline -> achieved fraction of it")
tot_us = 0.0
for i, op in enumerate(plan.ops):
    key = {weights.OP_CONV: "conv", weights.OP_DWCONV: "dwconv", weights.OP_MAXPOOL: "pool", weights.OP_L2NORM: "l2norm"}[op.type] + f":{i}"
    st = prof.get(key)
    if not st or not st["calls"]:
        continue
    # the library runs sibling layers (same input, adjacent output channel ranges: convPa + convDa) as one launch under
    # the first one's name: fold the silent sibling into this row
    nxt = plan.ops[i + 1] if i + 1 < len(plan.ops) else None
    nkey = f"conv:{i + 1}"
    if (nxt is not None and nxt.type == weights.OP_CONV and op.type == weights.OP_CONV and nxt.inp == op.inp and nxt.out == op.out
            and not (prof.get(nkey) or {}).get("calls")):
        import copy
        op = copy.copy(op)
        op.cout = op.cout + nxt.cout
        op.weight = np.concatenate([op.weight, nxt.weight])
        key = f"conv:{i}+{i + 1}"
    us = st["total_ms"] / st["calls"] * 1e3
    tot_us += us
    lvl_in, lvl_out = plan.tensors[op.inp][1], plan.tensors[op.out][1]
    px_in, px_out = (H >> lvl_in) * (Wd >> lvl_in), (H >> lvl_out) * (Wd >> lvl_out)
    k = op.ksize * op.ksize if op.type in (weights.OP_CONV, weights.OP_DWCONV) else 0
    flops = 2.0 * 2 * px_in * op.cout * (op.cin if op.type == weights.OP_CONV else 1) * k
    wbytes = (op.weight.size * (2 if prec == "FP16" else 4)) if op.weight is not None else 0
    byts = 2 * (px_in * op.cin * bpe(op.inp) + px_out * op.cout * bpe(op.out)) + wbytes
    ai = flops / byts
    t_mfma, t_hbm = flops / (PEAK * 1e12), byts / (HBM * 1e9)
    bound = "mfma" if t_mfma > t_hbm else "hbm"
    frac = max(t_mfma, t_hbm) / (us * 1e-6)
    shape = f"{op.cin}->{op.cout} k{op.ksize} @{H >> lvl_in}x{Wd >> lvl_in}"
    print(f"{key:10s} {shape:22s} {flops / 1e9:8.2f} {byts / 1e6:8.1f} {ai:7.0f} {us:8.1f} {flops / us / 1e6:8.1f} {byts / us / 1e3:7.0f}  {bound:4s} -> {frac:.2f}")
print(f"sum of layers {tot_us:.0f} us; forward pass {prof['net']['total_ms'] / prof['net']['calls'] * 1e3:.0f} us")
