"""GPU parity of the CNN (K1-K6) against the torch-CPU fp32 oracle, through the C ABI.

Tolerance (stated, per BASELINE.json north_star "within 1e-4 float"): the fp32 MFMA
is an exact k-ordered fmaf chain, torch-CPU (oneDNN) sums in another order, so
values agree to accumulated fp32 rounding: |gpu - oracle| <= 1e-4 * max(1, max|oracle|).
"""
import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import net
from tests.conftest import make_ctx

pytestmark = pytest.mark.gpu


def _tol(ref):
    return 1e-4 * max(1.0, float(np.abs(ref).max()))


def _input(sample_images, H, W, batch):
    xs = []
    for b in range(batch):
        img = sample_images[b % len(sample_images)]
        if img.shape[0] < H + 5 or img.shape[1] < W + 20 + 3 * b:            # 376 x 1240 does not fit a 375 x 1242 frame: wrap
            img = np.pad(img, ((0, max(0, H + 5 - img.shape[0])), (0, max(0, W + 20 + 3 * b - img.shape[1]))), mode="wrap")
        xs.append(img[5:5 + H, 20 + 3 * b:20 + 3 * b + W].astype(np.float32) / 255.0)
    return np.stack(xs)[:, None]


@pytest.mark.parametrize("H,W,batch", [(120, 392, 2), (360, 1176, 2), (192, 640, 1), (376, 1240, 2), (240, 784, 2)])   # 240x784: the reference's second engine size (engine_generation.py:20)
def test_vgg_forward_matches_oracle(vgg_weights_path, vgg_plan, sample_images, H, W, batch):
    ctx = make_ctx(vgg_weights_path, net_height=H, net_width=W)
    x = _input(sample_images, H, W, batch)
    det, desc = ctx.forward(x)
    rdet, rdesc, vals = net.forward(vgg_plan, x, return_all=True)
    # every intermediate activation (conv1a .. heads)
    for tid, (ch, lvl) in enumerate(vgg_plan.tensors):
        if tid in (vgg_plan.input_tensor, vgg_plan.desc_tensor):
            continue
        got = ctx.debug_tensor(tid, batch, ch, lvl)
        assert np.abs(got - vals[tid]).max() <= _tol(vals[tid]), f"tensor {tid}"
    assert np.abs(det - rdet).max() <= _tol(rdet)
    assert np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= 1e-4
    assert np.allclose(np.linalg.norm(desc, axis=-1), 1.0, atol=1e-5)
    ctx.close()


def _vgg_forward_f64(plan, x):
    """the same graph in float64 (torch CPU): the yardstick for "fp32 rounding level" below"""
    import torch
    import torch.nn.functional as F
    from spvo import weights as Wm
    vals = {plan.input_tensor: torch.from_numpy(x.astype(np.float64))}
    for op in plan.ops:
        src = vals[op.inp]
        if op.type == Wm.OP_CONV:
            in_off = getattr(op, "in_c_off", 0)
            y = F.conv2d(src[:, in_off:in_off + op.cin], torch.from_numpy(op.weight.astype(np.float64)), torch.from_numpy(op.bias.astype(np.float64)),
                         padding=op.ksize // 2)
            if op.flags & Wm.FLAG_RELU:
                y = F.relu(y)
            if op.flags & Wm.FLAG_POOL:
                y = F.max_pool2d(y, 2, 2)
            if op.out not in vals:
                vals[op.out] = torch.zeros((y.shape[0], plan.tensors[op.out][0], y.shape[2], y.shape[3]), dtype=torch.float64)
            vals[op.out][:, op.out_c_off:op.out_c_off + op.cout] = y
        else:
            assert op.type == Wm.OP_L2NORM
            vals[op.out] = src / torch.sqrt((src * src).sum(dim=1, keepdim=True))
    return {k: v.numpy() for k, v in vals.items()}


@pytest.mark.parametrize("H,W,batch", [(120, 392, 2), (360, 1176, 2), (192, 640, 1), (376, 1240, 2)])
def test_fp32_split_mode_matches_oracle_at_fp32_rounding_level(vgg_weights_path, vgg_plan, sample_images, H, W, batch):
    """spvo_set_fp32_split: the FP32 engine evaluated on the bf16 matrix pipe (three bf16 pieces per fp32 operand, six
    partial products per product, fp32 accumulation; csrc/conv_bf16x3.hip.h).  Same bar against the oracle as the native
    engine (1e-4, every intermediate tensor), AND its distance to a float64 evaluation of the graph stays within a small
    factor of the native fp32 engine's own distance: it is a re-ordered fp32 evaluation, not a lower precision."""
    from spvo import capi
    x = _input(sample_images, H, W, batch)
    rdet, rdesc, vals = net.forward(vgg_plan, x, return_all=True)
    ref64 = _vgg_forward_f64(vgg_plan, x)
    errs = {}
    for mode in ("native", "split"):
        ctx = capi.Context(net_height=H, net_width=W)
        ctx.set_fp32_split(mode == "split")
        ctx.load_weights(vgg_weights_path)
        assert ctx.engine_precision() == "FP32"
        det, desc = ctx.forward(x)
        e = {}
        for tid, (ch, lvl) in enumerate(vgg_plan.tensors):
            if tid in (vgg_plan.input_tensor, vgg_plan.desc_tensor):
                continue
            got = ctx.debug_tensor(tid, batch, ch, lvl)
            assert np.abs(got - vals[tid]).max() <= _tol(vals[tid]), f"{mode}: tensor {tid}"
            e[tid] = float(np.abs(got - ref64[tid]).max() / max(1.0, np.abs(ref64[tid]).max()))
        assert np.abs(det - rdet).max() <= _tol(rdet)
        assert np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= 1e-4
        assert np.allclose(np.linalg.norm(desc, axis=-1), 1.0, atol=1e-5)
        e["desc"] = float(np.abs(desc - ref64[vgg_plan.desc_tensor].transpose(0, 2, 3, 1)).max())
        errs[mode] = e
        ctx.close()
    print("max relative error against float64, native vs split:", {k: (errs["native"][k], errs["split"][k]) for k in errs["native"]})
    for k in errs["native"]:
        assert errs["split"][k] <= 4 * errs["native"][k] + 2e-7, (k, errs["native"][k], errs["split"][k])


# kernel family of conv1b, conv2a, conv2b, conv3a, conv3b, conv4a, conv4b, convPa(+Da) in the default engine, per network size: the plan
# loader's tile-count thresholds (csrc/spvo_core.hip, "Winograd F(2x2,3x3) for ..." / "F(4x4,3x3) ...") are pinned here, so that a change of
# the selection at a size shows up as a test failure, not as a silent change of arithmetic
_W4, _W2, _DIR = "conv_wino4_kernel", "conv_wino2_kernel", "conv_mfma_kernel"
DEFAULT_KERNELS = {
    (360, 1176, 2): [_W4, _W4, _W4, _W4, _W4, _W2, _W2, _W4],
    (376, 1240, 2): [_W4, _W4, _W4, _W4, _W4, _W2, _W2, _W4],
    (240, 784, 2): [_W4, _W4, _W4, _W2, _W2, _W2, _W2, _W2],        # 30 x 98 cells: conv4a / conv4b run the narrow F(2x2) form on 128 workgroups
    (192, 640, 1): None,
}


@pytest.mark.parametrize("H,W,batch", [(360, 1176, 2), (376, 1240, 2), (192, 640, 1), (240, 784, 2)])
def test_winograd_layers_stay_at_fp32_rounding_level(vgg_weights_path, vgg_plan, sample_images, H, W, batch, tuning):
    """The default FP32 engine runs its 3x3 layers through the Winograd kernels: F(4x4,3x3) (csrc/conv_wino4.hip.h) where the
    layer has enough 16 x 32 tiles, F(2x2,3x3) (csrc/conv_wino2.hip.h) for the rest; the diagnostic switch "wino4" = 0 keeps F(2x2)
    everywhere and "winograd" = 0 the direct kernel (spvo_set_tuning; read when an engine is loaded).  All three meet the 1e-4 bar against the oracle on
    every tensor, and against a float64 evaluation of the graph both Winograd engines stay within a small factor of the direct
    one: fp32 throughout; F(2x2)'s transforms only use the coefficients 0, +-1 and +-1/2, F(4x4)'s go up to 8 (points
    0, +-1, +-2, inf) and cost a factor of ~1-3 on this network's tensors."""
    from spvo import capi
    x = _input(sample_images, H, W, batch)
    rdet, rdesc, vals = net.forward(vgg_plan, x, return_all=True)
    ref64 = _vgg_forward_f64(vgg_plan, x)
    errs, kernels = {}, {}
    for mode in ("direct", "f2x2", "f4x4"):
        tuning(winograd=0 if mode == "direct" else 1, wino4=1 if mode == "f4x4" else 0)
        ctx = capi.Context(net_height=H, net_width=W)
        ctx.load_weights(vgg_weights_path)
        det, desc = ctx.forward(x)
        kernels[mode] = [ctx.stage_kernel(f"conv:{i}")[0] for i in range(1, 9)]
        e = {}
        for tid, (ch, lvl) in enumerate(vgg_plan.tensors):
            if tid in (vgg_plan.input_tensor, vgg_plan.desc_tensor):
                continue
            got = ctx.debug_tensor(tid, batch, ch, lvl)
            assert np.abs(got - vals[tid]).max() <= _tol(vals[tid]), f"{mode}: tensor {tid}"
            e[tid] = float(np.abs(got - ref64[tid]).max() / max(1.0, np.abs(ref64[tid]).max()))
        e["desc"] = float(np.abs(desc - ref64[vgg_plan.desc_tensor].transpose(0, 2, 3, 1)).max())
        assert np.abs(det - rdet).max() <= _tol(rdet) and np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= 1e-4
        errs[mode] = e
        ctx.close()
    print("kernels of conv1b .. convPa:", kernels)
    if DEFAULT_KERNELS[(H, W, batch)] is not None:
        assert kernels["f4x4"] == DEFAULT_KERNELS[(H, W, batch)], kernels["f4x4"]
    print("max relative error against float64, direct / F(2x2) / F(4x4):", {k: tuple(float(f"{errs[m][k]:.3g}") for m in errs) for k in errs["direct"]})
    # the switches did switch kernels
    assert all(k == "conv_mfma_kernel" for k in kernels["direct"]) and "conv_wino4_kernel" not in kernels["f2x2"]
    assert kernels["f4x4"][0] == "conv_wino4_kernel" and any(k.startswith("conv_wino") for k in kernels["f2x2"])
    for k in errs["direct"]:
        for mode in ("f2x2", "f4x4"):
            assert errs[mode][k] <= 4 * errs["direct"][k] + 2e-7, (mode, k, errs["direct"][k], errs[mode][k])


def _awkward_plan(seed):
    """A small graph whose 3x3 layers have the shapes VGG does not: channel counts that are not multiples of 64 (partial output
    tiles), 16 .. 72 input channels (4 .. 18 items per tile; the loader wants multiples of 8), a linear layer, pooled and unpooled
    layers at every level."""
    from spvo import weights as Wm
    rng = np.random.RandomState(seed)
    p = Wm.Plan()
    cur = p.add_tensor(1, 0)
    p.input_tensor = cur
    layers = [(1, 16, True, False), (16, 72, True, True), (72, 24, False, False), (24, 40, True, True), (40, 48, True, False), (48, 16, True, True), (16, 32, True, False)]
    level = 0
    for ci, co, relu, pool in layers:
        if pool:
            level += 1
        nxt = p.add_tensor(co, level)
        w = (rng.randn(co, ci, 3, 3) * np.sqrt(2.0 / (ci * 9))).astype(np.float32)
        b = (rng.randn(co) * 0.05).astype(np.float32)
        p.ops.append(Wm.Op(Wm.OP_CONV, cur, nxt, 0, ci, co, 3, (Wm.FLAG_RELU if relu else 0) | (Wm.FLAG_POOL if pool else 0), w, b))
        cur = nxt
    det, draw, desc = p.add_tensor(Wm.DET_CHANNELS, 3), p.add_tensor(Wm.DESC_CHANNELS, 3), p.add_tensor(Wm.DESC_CHANNELS, 3)
    for out, co in ((det, Wm.DET_CHANNELS), (draw, Wm.DESC_CHANNELS)):
        w = (rng.randn(co, 32, 1, 1) * np.sqrt(1.0 / 32)).astype(np.float32)
        p.ops.append(Wm.Op(Wm.OP_CONV, cur, out, 0, 32, co, 1, 0, w, (rng.randn(co) * 0.05).astype(np.float32)))
    p.ops.append(Wm.Op(Wm.OP_L2NORM, draw, desc, 0, Wm.DESC_CHANNELS, Wm.DESC_CHANNELS))
    p.det_tensor, p.desc_tensor = det, desc
    return p


@pytest.mark.parametrize("H,W", [(96, 168), (104, 200), (120, 392)])
@pytest.mark.parametrize("mode", ["f4x4", "f2x2"])
def test_winograd_kernels_on_awkward_layer_shapes(H, W, mode, sample_images, tuning, tmp_path):
    """Both Winograd kernels, forced onto every 3x3 layer of a graph with partial output tiles, short item chains and odd-sized
    maps (104 x 200 -> 52 x 100 -> 26 x 50 -> 13 x 25), against the oracle on every tensor."""
    from spvo import capi, weights as Wm
    plan = _awkward_plan(3)
    path = str(tmp_path / "awkward.spvw")
    Wm.save(plan, path)
    tuning(winograd_min_tiles=1, wino4_min_tiles=1, wino4=1 if mode == "f4x4" else 0)
    x = _input(sample_images, H, W, 2)
    rdet, rdesc, vals = net.forward(plan, x, return_all=True)
    ctx = capi.Context(net_height=H, net_width=W)
    ctx.load_weights(path)
    fams = [ctx.stage_kernel(f"conv:{i}")[0] for i in range(1, 7)]
    assert all(f.startswith("conv_wino") for f in fams), fams
    assert ("conv_wino4_kernel" in fams) == (mode == "f4x4"), fams
    det, desc = ctx.forward(x)
    for tid, (ch, lvl) in enumerate(plan.tensors):
        if tid in (plan.input_tensor, plan.desc_tensor):
            continue
        got = ctx.debug_tensor(tid, 2, ch, lvl)
        assert np.abs(got - vals[tid]).max() <= _tol(vals[tid]), (f"tensor {tid}", fams)
    assert np.abs(det - rdet).max() <= _tol(rdet) and np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= 1e-4
    ctx.close()


# Distance of an FP32 engine to a float64 evaluation of its graph (oracle/net.py with f64 = True: same fp32 weights and input),
# per tensor, relative to max(1, max |tensor|).  The yardstick is the fp32 ORACLE's own distance to float64 on the same tensor
# (torch-CPU fp32: what "an fp32 evaluation in some summation order" costs on this graph -- 1.5e-7 .. 7e-6 depending on the
# graph: BatchNorm scales of sp_mbv1 amplify it).  Two engines side by side:
#   * DIRECT kernels only (1x1 / depthwise / BatchNorm / residual layers, every 3x3 layer and the heads with the diagnostic
#     switches "winograd" = 0, "heads_fused" = 0): every tensor within 4 x the oracle's own error + 4e-7;
#   * the default engine (Winograd 3x3 layers, fused heads): within 4 x the direct engine's error + 2e-7 per tensor, or within
#     WINOGRAD_F64_LEVEL where the direct kernel happens to be much more accurate than fp32 needs to be.  F(4x4,3x3) on the
#     TRAINED sp_squeeze weights reaches 6.5e-6 of the tensor maximum (tensor 2 = its conv1b: 6.0e-6 against 4e-7 direct, a ratio
#     of 14; the seeded VGG weights: 2.5e-6, ratios 0.4 .. 2.7): recorded here, bar 8e-6 = 12 x inside north_star's 1e-4.
# The 1e-4 bar against the fp32 oracle (above) stays what north_star asks for; this test keeps an accuracy regression from
# hiding inside it (a kernel 40 x less accurate would still pass 1e-4).
WINOGRAD_F64_LEVEL = 8e-6


@pytest.mark.parametrize("graph,H,W,batch", [("vgg", 360, 1176, 2), ("vgg", 240, 784, 2), ("sp_squeeze", 360, 1176, 2), ("sp_squeeze", 240, 784, 2),
                                             ("sp_mbv1", 360, 1176, 2), ("sp_mbv2", 360, 1176, 2), ("sp_mbv1", 120, 392, 2), ("sp_mbv2", 120, 392, 2)])
def test_fp32_engines_stay_at_fp32_rounding_level_against_float64(graph, H, W, batch, vgg_weights_path, vgg_plan, sample_images, tuning):
    import os
    from spvo import weights
    from tests.conftest import GOLDEN
    path = vgg_weights_path if graph == "vgg" else os.path.join(GOLDEN, graph + ".spvw")
    plan = vgg_plan if graph == "vgg" else weights.load(path)
    x = _input(sample_images, H, W, batch)
    _, _, ref64 = net.forward(plan, x, return_all=True, f64=True)
    _, _, ref32 = net.forward(plan, x, return_all=True)

    def dist(get):
        e = {}
        for tid in range(len(plan.tensors)):
            if tid in (plan.input_tensor, plan.desc_tensor):
                continue
            e[tid] = float(np.abs(get(tid) - ref64[tid]).max() / max(1.0, np.abs(ref64[tid]).max()))
        return e
    errs, fams = {"oracle": dist(lambda tid: ref32[tid])}, {}
    errs["oracle"]["desc"] = float(np.abs(ref32[plan.desc_tensor] - ref64[plan.desc_tensor]).max())
    for mode in ("direct", "default"):
        if mode == "direct":
            tuning(winograd=0, heads_fused=0)
        else:
            tuning()
        ctx = make_ctx(path, net_height=H, net_width=W)
        det, desc = ctx.forward(x)
        e = dist(lambda tid: ctx.debug_tensor(tid, batch, *plan.tensors[tid]))
        e["desc"] = float(np.abs(desc - ref64[plan.desc_tensor].transpose(0, 2, 3, 1)).max())
        fams[mode] = sorted({ctx.stage_kernel(f"conv:{i}")[0] for i, op in enumerate(plan.ops) if op.type == weights.OP_CONV and op.cin > 1})
        errs[mode] = e
        ctx.close()
    worst = {m: max(errs[m], key=errs[m].get) for m in errs}
    print(f"{graph} {H}x{W}: fp32 oracle worst {errs['oracle'][worst['oracle']]:.3g} (tensor {worst['oracle']}); direct {fams['direct']} worst "
          f"{errs['direct'][worst['direct']]:.3g} (tensor {worst['direct']}); default {fams['default']} worst {errs['default'][worst['default']]:.3g} (tensor {worst['default']})")
    assert fams["direct"] == ["conv_mfma_kernel"], fams["direct"]
    wino = any(f.startswith("conv_wino") for f in fams["default"])
    for k, v in errs["direct"].items():
        assert v <= 4 * errs["oracle"][k] + 4e-7, (graph, "direct", k, errs["oracle"][k], v)
    for k, v in errs["default"].items():
        assert v <= max(4 * errs["direct"][k] + 2e-7, WINOGRAD_F64_LEVEL if wino else 0.0), (graph, "default", k, errs["direct"][k], v)


def test_fp32_split_mode_rejects_other_graphs(squeeze_weights_path):
    from spvo import capi
    ctx = capi.Context()
    ctx.set_fp32_split(True)
    with pytest.raises(capi.SpvoError):
        ctx.load_weights(squeeze_weights_path)
    ctx.set_fp32_split(False)
    ctx.load_weights(squeeze_weights_path)
    ctx.close()


def test_squeeze_forward_matches_oracle(ctx_squeeze, squeeze_plan, sample_images):
    x = _input(sample_images, 360, 1176, 2)
    det, desc = ctx_squeeze.forward(x)
    rdet, rdesc = net.forward(squeeze_plan, x)
    assert np.abs(det - rdet).max() <= _tol(rdet)
    assert np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= 1e-4


@pytest.mark.parametrize("name", ["mbv1", "mbv2"])
@pytest.mark.parametrize("H,W,batch", [(120, 392, 2), (360, 1176, 2)])
def test_mobilenet_forward_matches_oracle(name, H, W, batch, sample_images):
    """Config 3's graphs in fp32: depthwise 3x3, BatchNorm after ReLU (mbv1), residual Add + ReLU (mbv2),
    every one of them fused into the producing convolution; each intermediate tensor is compared."""
    import os
    from spvo import weights
    from tests.conftest import GOLDEN
    path = os.path.join(GOLDEN, f"sp_{name}.spvw")
    plan = weights.load(path)
    ctx = make_ctx(path, net_height=H, net_width=W)
    x = _input(sample_images, H, W, batch)
    det, desc = ctx.forward(x)
    rdet, rdesc, vals = net.forward(plan, x, return_all=True)
    for tid, (ch, lvl) in enumerate(plan.tensors):
        if tid in (plan.input_tensor, plan.desc_tensor):
            continue
        got = ctx.debug_tensor(tid, batch, ch, lvl)
        assert np.abs(got - vals[tid]).max() <= _tol(vals[tid]), f"{name} tensor {tid}"
    assert np.abs(det - rdet).max() <= _tol(rdet)
    assert np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= 1e-4
    ctx.close()


@pytest.mark.parametrize("name", ["sp_squeeze", "sp_mbv1", "sp_mbv2"])
def test_engines_match_the_direct_onnx_evaluation(name):
    """The HIP engine against what the reference's ONNX graph gives when it is evaluated node by node by an interpreter that
    shares nothing with the packer (oracle/onnx_direct.py; fixtures frozen by tests/golden/make_onnx_direct_golden.py from
    models/<name>_b1.onnx): pins the network stage to the reference's own artefact, not to the plan-based oracle."""
    import os
    from tests.conftest import GOLDEN
    g = np.load(os.path.join(GOLDEN, f"onnx_direct_{name}_64x96.npz"))
    ctx = make_ctx(os.path.join(GOLDEN, name + ".spvw"), net_height=64, net_width=96)
    det, desc = ctx.forward(g["x"])
    ctx.close()
    assert np.abs(det - g["det"]).max() <= _tol(g["det"])
    assert np.abs(desc - g["desc"].transpose(0, 2, 3, 1)).max() <= 1e-4


@pytest.mark.parametrize("graph,H,W,batch", [("vgg", 192, 640, 2), ("vgg", 360, 1176, 2), ("squeeze", 192, 640, 2), ("vgg", 120, 392, 1),
                                             ("mbv1", 192, 640, 2), ("mbv2", 192, 640, 2), ("mbv2", 360, 1176, 2)])
def test_fp16_engine_matches_fp16_oracle(graph, H, W, batch, vgg_plan, squeeze_plan, sample_images, tmp_path):
    """BASELINE config 3: FP16 engines (fp16 storage, fp32 accumulation, fp32 bindings) against the oracle's restatement
    of the same engine.  Tolerance: both sides round every stored activation to fp16 (relative step 2^-11 = 4.9e-4);
    their fp32 sums differ in order, so a value next to a rounding boundary may land one fp16 step apart and the
    difference propagates through the following layers: |gpu - oracle| <= 4e-3 * max(1, max|oracle|) per tensor
    (a handful of steps), and <= 4e-3 on the unit-norm descriptors."""
    import copy
    from spvo import weights
    if graph in ("vgg", "squeeze"):
        plan = copy.copy(vgg_plan if graph == "vgg" else squeeze_plan)
    else:                                                            # depthwise, BatchNorm-after-ReLU, residual layers
        import os
        from tests.conftest import GOLDEN
        plan = weights.load(os.path.join(GOLDEN, f"sp_{graph}.spvw"))
    plan.precision = "FP16"
    path = str(tmp_path / weights.engine_name(graph, 2, H, W, "FP16"))
    weights.save(plan, path)
    ctx = make_ctx(path, net_height=H, net_width=W)
    assert ctx.engine_precision() == "FP16"
    x = _input(sample_images, H, W, batch)
    det, desc = ctx.forward(x)
    rdet, rdesc, vals = net.forward(plan, x, return_all=True)
    # the MobileNet graphs are 20-27 layers deep (VGG: 12) and their BatchNorm scales amplify a one-step difference:
    # the same kind of divergence reaches 5e-3 at their outputs, so their bar is 8e-3
    rel = 4e-3 if graph in ("vgg", "squeeze") else 8e-3
    for tid, (ch, lvl) in enumerate(plan.tensors):
        if tid in (plan.input_tensor, plan.desc_tensor):
            continue
        got = ctx.debug_tensor(tid, batch, ch, lvl)
        tol = rel * max(1.0, float(np.abs(vals[tid]).max()))
        assert np.abs(got - vals[tid]).max() <= tol, f"tensor {tid}: {np.abs(got - vals[tid]).max()} > {tol}"
    assert np.abs(det - rdet).max() <= rel * max(1.0, float(np.abs(rdet).max()))
    assert np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= rel
    assert np.allclose(np.linalg.norm(desc, axis=-1), 1.0, atol=1e-5)
    # and the half-precision engine stays close to the fp32 one (sanity of the whole path, not a parity bar)
    plan32 = copy.copy(plan)
    plan32.precision = "FP32"
    d32, _ = net.forward(plan32, x)
    assert np.abs(det - d32).max() <= 2e-2 * max(1.0, float(np.abs(d32).max()))
    ctx.close()


@pytest.mark.parametrize("graph,H,W,batch", [("mbv1", 192, 640, 2), ("mbv2", 192, 640, 2), ("mbv1", 360, 1176, 2), ("vgg", 120, 392, 2)])
def test_int8_engine_is_bit_exact(graph, H, W, batch, vgg_plan, sample_images, tmp_path):
    """BASELINE config 5: INT8 engines.  The arithmetic is defined by oracle/net_int8.py (exact int32 accumulation,
    separately rounded fp32 multiply / add in the requantisation), so every int8 tensor and the fp32 detector
    output must agree with the oracle BIT FOR BIT; only the L2-normalised descriptors (a float reduction whose
    order differs) carry a tolerance, 1e-5."""
    import copy, os
    from oracle import net_int8
    from spvo import weights
    from tests.conftest import GOLDEN
    plan = copy.copy(vgg_plan) if graph == "vgg" else weights.load(os.path.join(GOLDEN, f"sp_{graph}.spvw"))
    x = _input(sample_images, H, W, batch)
    plan.act_scales = net_int8.calibrate(plan, [x[:1]])                    # calibrated on the first image only
    plan.precision = "INT8"
    path = str(tmp_path / weights.engine_name(graph, 2, H, W, "INT8"))
    weights.save(plan, path)
    ctx = make_ctx(path, net_height=H, net_width=W)
    assert ctx.engine_precision() == "INT8"
    det, desc = ctx.forward(x)
    rdet, rdesc, vals = net_int8.forward(plan, x, return_all=True)
    for tid, (ch, lvl) in enumerate(plan.tensors):
        if tid in (plan.input_tensor, plan.desc_tensor):
            continue
        got = ctx.debug_tensor(tid, batch, ch, lvl)
        if vals[tid].dtype == np.int8:
            assert np.array_equal(got, vals[tid].astype(np.float32)), f"int8 tensor {tid}: {(got != vals[tid]).sum()} of {got.size} values differ"
        else:
            assert np.array_equal(got, vals[tid]), f"fp32 tensor {tid}: max diff {np.abs(got - vals[tid]).max()}"
    assert np.array_equal(det, rdet)
    assert np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= 1e-5
    ctx.close()


def test_forward_is_deterministic_and_batch_independent(ctx_vgg, sample_images):
    x = _input(sample_images, 360, 1176, 2)
    d1, s1 = ctx_vgg.forward(x)
    d2, s2 = ctx_vgg.forward(x)
    assert np.array_equal(d1, d2) and np.array_equal(s1, s2)          # bit-identical reruns
    d3, s3 = ctx_vgg.forward(x[1:2])
    assert np.array_equal(d3[0], d1[1]) and np.array_equal(s3[0], s1[1])  # image 1 alone == image 1 in the pair
    # four images per launch (two stereo pairs: spvo_set_trunk_pairing) -- each image as it comes out of a two-image pass
    x4 = np.concatenate([x, _input(sample_images, 360, 1176, 3)[1:3]])
    d4, s4 = ctx_vgg.forward(x4)
    d5, s5 = ctx_vgg.forward(x4[2:4])
    assert np.array_equal(d4[:2], d1) and np.array_equal(s4[:2], s1) and np.array_equal(d4[2:], d5) and np.array_equal(s4[2:], s5)


def test_forward_linearity_of_first_layers(ctx_vgg, vgg_plan):
    # size-independent property at full size: a zero image gives relu(bias) = 0 everywhere in conv1a (bias 0)
    x = np.zeros((1, 1, 360, 1176), np.float32)
    ctx_vgg.forward(x)
    a = ctx_vgg.debug_tensor(1, 1, 64, 0)
    assert np.all(a == 0)


def test_errors(vgg_weights_path, tmp_path):
    from spvo import capi
    ctx = make_ctx(None)
    with pytest.raises(capi.SpvoError) as e:                          # nn.cpp:53-55 "no such engine file"
        ctx.load_weights(str(tmp_path / "missing.spvw"))
    assert e.value.code == -3 and "no such engine file" in str(e.value)
    with pytest.raises(capi.SpvoError) as e:
        ctx.forward(np.zeros((1, 1, 360, 1176), np.float32))
    assert e.value.code == -4
    bad = tmp_path / "bad.spvw"
    bad.write_bytes(b"not a weight file at all, just bytes" * 4)
    with pytest.raises(capi.SpvoError) as e:
        ctx.load_weights(str(bad))
    assert e.value.code == -3
    ctx.close()
