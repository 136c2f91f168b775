"""Oracle for the CNN: torch-CPU fp32 executor.

Follows what the reference's TensorRT engine computes at
feature_detection_neural_network.cpp:163-176 (`context_->enqueue`): the ONNX
graph {Conv (dense and depthwise), Relu, MaxPool, Concat, BatchNormalization, Add,
ReduceL2, Div} in fp32, input [B,1,H,W] in
[0,1], outputs `output_det` [B,65,H/8,W/8] and `output_desc` [B,256,H/8,W/8]
(L2-normalised over channels in-graph, no epsilon).  TensorRT's own summation
order is unknowable, so parity is within a stated tolerance, not bit-exact.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from spvo import weights as W


def _h(t: torch.Tensor) -> torch.Tensor:
    """round to fp16 (nearest even) and back: the storage precision of an FP16 engine"""
    return t.half().float()


def forward(plan: W.Plan, x: np.ndarray, return_all: bool = False, f64: bool = False):
    """x: float32 [B,1,H,W].  Returns (det, desc) as float32 numpy NCHW (float64 with `f64`: the same graph evaluated in
    double precision from the same fp32 weights and input -- the yardstick the tests use for "fp32 rounding level",
    FP32 plans only).

    plan.precision == "FP16" restates an engine built by engine_generation.py:13-56 with --fp16: the bindings stay
    fp32 (nn.cpp:117) and everything between them is half precision.  TensorRT's internal choices (which layers it
    keeps in fp32, its accumulation width) are not observable, so the restatement fixes the simplest consistent
    reading: weights rounded to fp16, products accumulated in fp32 together with an fp32 bias, every intermediate
    tensor rounded to fp16 when it is stored; the two heads' last convolutions, the L2 normalisation and the
    outputs are fp32."""
    assert x.dtype == np.float32 and x.ndim == 4 and x.shape[1] == 1
    torch.set_grad_enabled(False)
    B, _, H, Wd = x.shape
    vals = {}
    vals[plan.input_tensor] = torch.from_numpy(x)
    fp16 = plan.precision == "FP16"
    assert not (fp16 and f64)
    dt = torch.float64 if f64 else torch.float32
    vals[plan.input_tensor] = vals[plan.input_tensor].to(dt)
    keep_f32 = {plan.input_tensor, plan.det_tensor, plan.desc_tensor} | {op.inp for op in plan.ops if op.type == W.OP_L2NORM}
    for op in plan.ops:
        src = vals[op.inp]
        in_off = getattr(op, "in_c_off", 0)
        if op.type in (W.OP_CONV, W.OP_DWCONV):
            xin = src[:, in_off:in_off + op.cin]
            wt = torch.from_numpy(op.weight).to(dt)
            y = F.conv2d(xin, _h(wt) if fp16 else wt, torch.from_numpy(op.bias).to(dt),
                         stride=1, padding=op.ksize // 2,
                         groups=op.cin if op.type == W.OP_DWCONV else 1)
            if op.flags & W.FLAG_RELU:
                y = F.relu(y)
            if op.flags & W.FLAG_BN:
                # ONNX BatchNormalization (inference): (x - mean) / sqrt(var + eps) * gamma + beta
                c = op.cout
                gam, bet, mean, var = (torch.from_numpy(op.bn[i * c:(i + 1) * c].copy()).to(dt) for i in range(4))
                y = F.relu(F.batch_norm(y, mean, var, gam, bet, training=False, eps=float(op.bn[4 * c])))
            if op.flags & W.FLAG_ADD:
                y = F.relu(y + vals[op.residual])
            if op.flags & W.FLAG_POOL:
                y = F.max_pool2d(y, 2, 2)
            ch, lvl = plan.tensors[op.out]
            if op.out not in vals:
                vals[op.out] = torch.zeros((B, ch, H >> lvl, Wd >> lvl), dtype=dt)
            vals[op.out][:, op.out_c_off:op.out_c_off + op.cout] = _h(y) if fp16 and op.out not in keep_f32 else y
        elif op.type == W.OP_MAXPOOL:
            vals[op.out] = F.max_pool2d(src, 2, 2)
        elif op.type == W.OP_L2NORM:
            # ONNX tail: ReduceL2(axes=[1], keepdims=1) then Div, no epsilon
            vals[op.out] = src / torch.sqrt((src * src).sum(dim=1, keepdim=True))
        else:
            raise NotImplementedError(op.type)
    det = vals[plan.det_tensor].numpy()
    desc = vals[plan.desc_tensor].numpy()
    if return_all:
        return det, desc, {k: v.numpy() for k, v in vals.items()}
    return det, desc
