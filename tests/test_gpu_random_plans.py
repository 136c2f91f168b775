"""Randomised convolution stacks through every engine precision (GPU vs oracle, every tensor compared).

The named graphs (VGG, squeeze, mbv1, mbv2) exercise a handful of layer shapes; these seeded random plans cover the
rest of the kernels' parameter space: channel counts that leave partial output-channel tiles, 1x1 and 3x3 layers with and
without ReLU / fused pooling, channel-slice outputs (Concat), stand-alone pooling, ragged image sizes that leave partial
tiles in both directions, one and two images per launch."""
import numpy as np
import pytest

import oracle  # noqa: F401
from oracle import net, net_int8
from spvo import weights as W
from tests.conftest import make_ctx

pytestmark = pytest.mark.gpu


def random_plan(rng, precision):
    chans = [32, 64, 96, 128] if precision == "INT8" else [16, 32, 48, 64, 80, 128]
    p = W.Plan()
    cur = p.input_tensor = p.add_tensor(1, 0)
    level = 0

    def conv(src, dst, c_off, cin, cout, k, relu, pool, in_off=0):
        fan = cin * k * k
        w = (rng.randn(cout, cin, k, k) * np.sqrt(2.0 / fan)).astype(np.float32)
        b = (rng.randn(cout) * 0.1).astype(np.float32)
        op = W.Op(W.OP_CONV, src, dst, c_off, cin, cout, k, (W.FLAG_RELU if relu else 0) | (W.FLAG_POOL if pool else 0), w, b)
        op.in_c_off = in_off
        p.ops.append(op)

    c = int(rng.choice(chans))
    nxt = p.add_tensor(c, 0)
    conv(cur, nxt, 0, 1, c, 3, True, False)
    cur, cc = nxt, c
    while level < 3:
        for _ in range(rng.randint(0, 2)):                                   # layers that keep the resolution
            co = int(rng.choice(chans))
            k = int(rng.choice([1, 3]))
            if precision != "INT8" and rng.rand() < 0.35:                    # two convolutions into channel slices of one tensor
                co2 = int(rng.choice(chans))
                nxt = p.add_tensor(co + co2, level)
                conv(cur, nxt, 0, cc, co, k, True, False)
                conv(cur, nxt, co, cc, co2, 4 - k, True, False)
                co += co2
            else:
                nxt = p.add_tensor(co, level)
                conv(cur, nxt, 0, cc, co, k, bool(rng.rand() < 0.8), False)
            cur, cc = nxt, co
        co = int(rng.choice(chans))
        if precision != "INT8" and rng.rand() < 0.3:                         # stand-alone pool
            nxt = p.add_tensor(cc, level + 1)
            p.ops.append(W.Op(W.OP_MAXPOOL, cur, nxt, 0, cc, cc))
            cur = nxt
        else:                                                                # pool fused into a convolution
            nxt = p.add_tensor(co, level + 1)
            conv(cur, nxt, 0, cc, co, int(rng.choice([1, 3])), True, True)
            cur, cc = nxt, co
        level += 1
    heads = p.add_tensor(512, 3)
    conv(cur, heads, 0, cc, 256, 3, True, False)
    conv(cur, heads, 256, cc, 256, 3, True, False)
    p.det_tensor = p.add_tensor(W.DET_CHANNELS, 3)
    raw = p.add_tensor(W.DESC_CHANNELS, 3)
    p.desc_tensor = p.add_tensor(W.DESC_CHANNELS, 3)
    conv(heads, p.det_tensor, 0, 256, W.DET_CHANNELS, 1, False, False, in_off=0)
    conv(heads, raw, 0, 256, W.DESC_CHANNELS, 1, False, False, in_off=256)
    p.ops.append(W.Op(W.OP_L2NORM, raw, p.desc_tensor, 0, W.DESC_CHANNELS, W.DESC_CHANNELS))
    return p


@pytest.mark.parametrize("precision", ["FP32", "FP16", "INT8"])
@pytest.mark.parametrize("seed,H,W_,batch", [(0, 40, 72, 2), (1, 56, 104, 1), (2, 24, 40, 2), (3, 72, 136, 2), (4, 48, 200, 1),
                                             (5, 64, 64, 2), (6, 8, 520, 2), (7, 136, 8, 1), (8, 88, 328, 2), (9, 32, 96, 2)])
def test_random_plan(precision, seed, H, W_, batch, tmp_path):
    rng = np.random.RandomState(100 + seed)
    plan = random_plan(rng, precision)
    x = rng.rand(batch, 1, H, W_).astype(np.float32)
    plan.precision = precision
    if precision == "INT8":
        plan.act_scales = net_int8.calibrate(plan, [x])
    path = str(tmp_path / W.engine_name("random", 2, H, W_, precision))
    W.save(plan, path)
    ctx = make_ctx(path, net_height=H, net_width=W_)
    det, desc = ctx.forward(x)
    if precision == "INT8":
        rdet, rdesc, vals = net_int8.forward(plan, x, return_all=True)
    else:
        rdet, rdesc, vals = net.forward(plan, x, return_all=True)
    rel = {"FP32": 1e-4, "FP16": 4e-3}.get(precision)
    for tid, (ch, lvl) in enumerate(plan.tensors):
        if tid in (plan.input_tensor, plan.desc_tensor):
            continue
        got = ctx.debug_tensor(tid, batch, ch, lvl)
        if precision == "INT8":
            assert np.array_equal(got, vals[tid].astype(np.float32)), f"tensor {tid} ({ch} ch, level {lvl}): {(got != vals[tid]).sum()} values differ"
        else:
            tol = rel * max(1.0, float(np.abs(vals[tid]).max()))
            assert np.abs(got - vals[tid]).max() <= tol, f"tensor {tid} ({ch} ch, level {lvl}): {np.abs(got - vals[tid]).max()} > {tol}"
    if precision == "INT8":
        assert np.array_equal(det, rdet)
    else:
        assert np.abs(det - rdet).max() <= rel * max(1.0, float(np.abs(rdet).max()))
    assert np.abs(desc - rdesc.transpose(0, 2, 3, 1)).max() <= (1e-4 if precision != "FP16" else 4e-3)
    ctx.close()
