"""Post-training calibration for INT8 engines (BASELINE config 5; the reference has no int8 path).

An INT8 engine file is the canonical fp32 plan plus one symmetric activation scale per tensor
(`Plan.act_scales`, real = q * scale); the library quantises the weights itself when it loads the file
(per output channel, csrc/conv_i8.hip.h).  The scales come from running the FP32 engine of the same plan on
calibration images ON THE DEVICE and taking a high percentile of |activation| per tensor.

    plan = weights.load("sp_mbv1_2_360_1176_FP32.spvw")
    plan.act_scales = quant.calibrate(plan, [x0, x1, ...], 360, 1176)     # x: float32 [B, 1, H, W] in [0, 1]
    weights.save(plan, "sp_mbv1_2_360_1176_INT8.spvw", precision="INT8")
"""
from __future__ import annotations

import os
import tempfile
from typing import Iterable

import numpy as np

from . import capi, weights


def _fp32_context(plan: weights.Plan, height: int, width: int) -> capi.Context:
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "calib_FP32.spvw")
        weights.save(plan, path, precision="FP32")
        ctx = capi.Context(net_height=height, net_width=width)
        ctx.load_weights(path)
    return ctx


def calibration_inputs(plan: weights.Plan, images: Iterable[np.ndarray], height: int, width: int) -> np.ndarray:
    """uint8 grey images -> the network's fp32 input [N, 1, H, W] through the library's own preprocess (crop + resize on
    the device, then x * (1/255) as csrc/post.hip.h does it)."""
    ctx = _fp32_context(plan, height, width)
    try:
        P = np.eye(3, 4)
        out = [ctx.preprocess(im, P)[0].astype(np.float32) * np.float32(1.0 / 255.0) for im in images]
    finally:
        ctx.close()
    return np.stack(out)[:, None]


def calibrate(plan: weights.Plan, inputs: Iterable[np.ndarray], height: int, width: int, percentile: float = 99.999) -> np.ndarray:
    """Per-tensor activation scales = percentile(|activation|) / 127 over the calibration inputs (fp32 engine on the GPU)."""
    ctx = _fp32_context(plan, height, width)
    amax = np.zeros(len(plan.tensors), np.float64)
    try:
        for x in inputs:
            x = np.ascontiguousarray(x, np.float32)
            for b0 in range(0, len(x), 2):
                xb = x[b0:b0 + 2]
                ctx.forward(xb)
                for t, (ch, lvl) in enumerate(plan.tensors):
                    v = xb if t == plan.input_tensor else ctx.debug_tensor(t, len(xb), ch, lvl)
                    amax[t] = max(amax[t], float(np.percentile(np.abs(v), percentile)))
    finally:
        ctx.close()
    amax[amax == 0] = 1.0
    return (amax.astype(np.float32) / np.float32(127.0)).astype(np.float32)
