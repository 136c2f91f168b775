"""Oracle for the INT8 engines (BASELINE config 5: MobileNet-backbone SuperPoint in int8).  TEST INFRASTRUCTURE.

The reference has no int8 path (hpp:124-126 knows FP32 and FP16 only): config 5 is a build-side extension, so this
file DEFINES the arithmetic the HIP kernels must reproduce, and it is chosen to be exactly reproducible: integer
accumulation is exact, and every float operation is ONE IEEE-754 operation of fp32 values in a fixed order -- a fused
multiply-add (correctly rounded: `fma32` below), a multiply, or a maximum -- so the integer tensors and the fp32 detector
output agree with the GPU bit for bit.

Round 6: the requantisation is an AFFINE per channel, evaluated as one fused multiply-add (as int8 inference engines
apply their scale and bias), where rounds 2-5 used a separately rounded multiply and add; and the final multiply by
1 / s_out is folded into the last affine of the chain where the chain allows it.  Half the vector instructions per value:
the fused MobileNet blocks are bound by exactly these chains (csrc/conv_i8_fused.hip.h).

Scheme (symmetric, per-tensor activations / per-output-channel weights, the usual post-training quantisation):
  * a quantised tensor t holds q in [-127, 127] with real value q * s_t; s_t = calibrated |max| / 127 (weights.Plan.act_scales);
  * weights of output channel co: w_q = clip(rint(w / ws)), ws = max|w[co]| / 127 (fp32 division; 1 if the channel is all zero);
  * convolution: acc = sum w_q * x_q (int32, exact);  r = fma(f32(acc), m, bias) with m = f32(ws * s_in);  [ReLU]
    [BatchNorm: r = fma(r, bn_scale, bn_shift); ReLU]  [residual: r = fma(f32(res_q), s_res, r); ReLU]  [2x2 max-pool]
    then q_out = clip(rint(r * (1 / s_out))) -- or r itself for the fp32 bindings (output_det, raw descriptors);
  * FOLDING: when the output is quantised, the op has no residual input, and the chain holds an affine on an int8 accumulator or a
    BatchNorm, the LAST affine's two constants are multiplied by 1 / s_out beforehand -- k' = f32(k * inv_s), b' = f32(b * inv_s), one
    fp32 multiply each -- and the final multiply is dropped: q_out = clip(rint(r'));
  * the single-channel stem (tensors whose channel count is not a multiple of 16, and the layers that read them) stays
    fp32: r = bias, then r = fma(w[tap], x[tap], r) over the taps in row-major order.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from spvo import weights as W

f32 = np.float32


def weight_scales(w: np.ndarray) -> np.ndarray:
    amax = np.abs(w.reshape(w.shape[0], -1)).max(axis=1).astype(f32)
    ws = (amax / f32(127.0)).astype(f32)
    ws[amax == 0] = f32(1.0)
    return ws


def quantize_weights(w: np.ndarray):
    ws = weight_scales(w)
    q = np.clip(np.rint((w / ws[:, None, None, None]).astype(f32)), -127, 127).astype(np.int8)
    return q, ws


def bn_affine(op):
    """scale / shift exactly as the library's loader folds them (double arithmetic, one rounding to fp32 each)"""
    c = op.cout
    g, b, m, v = (op.bn[i * c:(i + 1) * c].astype(np.float64) for i in range(4))
    k = g / np.sqrt(v + np.float64(op.bn[4 * c]))
    return k.astype(f32), (b - m * k).astype(f32)


def fma32(a, b, c) -> np.ndarray:
    """IEEE-754 fusedMultiplyAdd of fp32 arrays, correctly rounded (what v_fma_f32 / std::fmaf compute): the product of two 24-bit
    significands is exact in float64; the sum with c is rounded to 53 bits ROUND-TO-ODD (TwoSum gives the rounding error exactly: when the
    sum is inexact and its last bit is even, step to the neighbour on the error's side), and rounding that to fp32 is then the correctly
    rounded result -- 53 >= 2 * 24 + 2 bits make the double rounding innocuous (Boldo & Melquiond)."""
    a64, b64, c64 = (np.asarray(v, np.float32).astype(np.float64) for v in (a, b, c))
    p = a64 * b64
    s = p + c64
    bb = s - p
    err = (p - (s - bb)) + (c64 - bb)
    s = np.ascontiguousarray(np.broadcast_to(s, np.broadcast(s, err).shape))
    even = (s.view(np.int64) & 1) == 0
    step = np.where(err > 0, np.inf, -np.inf)
    s = np.where((err != 0) & even & np.isfinite(s), np.nextafter(s, step), s)
    return s.astype(f32)


def quantize(r: np.ndarray, s: np.float32, folded: bool = False) -> np.ndarray:
    """clip(rint(r / s)): the multiply by 1 / s is already inside r when the op's last affine was folded"""
    inv = f32(1.0) / f32(s)
    v = r.astype(f32) if folded else (r.astype(f32) * inv).astype(f32)
    return np.clip(np.rint(v), -127, 127).astype(np.int8)


def folds(plan: W.Plan, op) -> bool:
    """the last affine of the op's chain absorbs 1 / s_out (module docstring): quantised output, no residual, an affine to fold into"""
    return is_quantized(plan, op.out) and not (op.flags & W.FLAG_ADD) and (is_quantized(plan, op.inp) or bool(op.flags & W.FLAG_BN))


def is_quantized(plan: W.Plan, t: int) -> bool:
    keep = {plan.input_tensor, plan.det_tensor, plan.desc_tensor} | {op.inp for op in plan.ops if op.type == W.OP_L2NORM}
    return t not in keep and plan.tensors[t][0] % 16 == 0


def calibrate(plan: W.Plan, inputs, percentile: float = 99.999) -> np.ndarray:
    """Per-tensor activation scales from the fp32 network on calibration inputs ([B,1,H,W] fp32 arrays): the given
    percentile of |activation| (clipping the rarest outliers keeps more resolution for the rest: on sp_mbv1 the
    keypoint sets of the int8 and the fp32 network overlap 77 % with 99.999 against 68 % with the plain maximum)."""
    from oracle import net
    p32 = W.Plan(tensors=plan.tensors, ops=plan.ops, input_tensor=plan.input_tensor, det_tensor=plan.det_tensor, desc_tensor=plan.desc_tensor)
    amax = np.zeros(len(plan.tensors), np.float64)
    for x in inputs:
        _, _, vals = net.forward(p32, x, return_all=True)
        for t, v in vals.items():
            amax[t] = max(amax[t], float(np.percentile(np.abs(v), percentile)))
    amax[amax == 0] = 1.0
    return (amax.astype(f32) / f32(127.0)).astype(f32)


def forward(plan: W.Plan, x: np.ndarray, return_all: bool = False):
    """x: float32 [B,1,H,W].  Returns (det, desc) float32 NCHW; with return_all also {tensor id: int8 or float32 array}."""
    assert plan.precision == "INT8" and plan.act_scales is not None
    B, _, H, Wd = x.shape
    s = plan.act_scales.astype(f32)
    vals = {plan.input_tensor: x.astype(f32)}
    for op in plan.ops:
        src = vals[op.inp]
        if op.type in (W.OP_CONV, W.OP_DWCONV):
            dw = op.type == W.OP_DWCONV
            xin = src[:, op.in_c_off:op.in_c_off + op.cin]
            k, pad = op.ksize, op.ksize // 2
            fold = folds(plan, op)
            inv_out = f32(1.0) / f32(s[op.out])
            has_bn = bool(op.flags & W.FLAG_BN)
            if not is_quantized(plan, op.inp):
                # fp32 stem: Cin = 1, bias first, then one fused multiply-add per tap
                assert op.cin == 1 and not dw and not (op.flags & (W.FLAG_POOL | W.FLAG_ADD))
                xp = np.pad(xin[:, 0], ((0, 0), (pad, pad), (pad, pad)))
                r = np.broadcast_to(op.bias.astype(f32)[None, :, None, None], (B, op.cout, H >> plan.tensors[op.inp][1], Wd >> plan.tensors[op.inp][1])).copy()
                hh, ww = r.shape[2], r.shape[3]
                for ky in range(k):
                    for kx in range(k):
                        r = fma32(op.weight[:, 0, ky, kx].astype(f32)[None, :, None, None], xp[:, None, ky:ky + hh, kx:kx + ww], r)
            else:
                wq, ws = quantize_weights(op.weight)
                acc = F.conv2d(torch.from_numpy(xin.astype(np.float64)), torch.from_numpy(wq.astype(np.float64)), None, stride=1, padding=pad,
                               groups=op.cin if dw else 1).numpy()
                assert np.abs(acc).max() < 2 ** 31
                m = (ws * s[op.inp]).astype(f32)
                bias = op.bias.astype(f32)
                if fold and not has_bn:          # the conv's own affine is the chain's last one
                    m, bias = (m * inv_out).astype(f32), (bias * inv_out).astype(f32)
                r = fma32(acc.astype(np.int32).astype(f32), m[None, :, None, None], bias[None, :, None, None])
            if op.flags & W.FLAG_RELU:
                r = np.maximum(r, f32(0))
            if has_bn:
                sc, sh = bn_affine(op)
                if fold:
                    sc, sh = (sc * inv_out).astype(f32), (sh * inv_out).astype(f32)
                r = np.maximum(fma32(r, sc[None, :, None, None], sh[None, :, None, None]), f32(0))
            if op.flags & W.FLAG_ADD:
                r = np.maximum(fma32(vals[op.residual].astype(f32), f32(s[op.residual]), r), f32(0))
            if op.flags & W.FLAG_POOL:
                r = r.reshape(B, op.cout, r.shape[2] // 2, 2, r.shape[3] // 2, 2).max(axis=(3, 5))
            out = quantize(r, s[op.out], fold) if is_quantized(plan, op.out) else r
            ch, lvl = plan.tensors[op.out]
            if op.out not in vals:
                vals[op.out] = np.zeros((B, ch, H >> lvl, Wd >> lvl), out.dtype)
            vals[op.out][:, op.out_c_off:op.out_c_off + op.cout] = out
        elif op.type == W.OP_L2NORM:
            v = torch.from_numpy(src)
            vals[op.out] = (v / torch.sqrt((v * v).sum(dim=1, keepdim=True))).numpy()
        else:
            raise NotImplementedError("INT8 engines: op type %d" % op.type)
    det, desc = vals[plan.det_tensor], vals[plan.desc_tensor]
    return (det, desc, vals) if return_all else (det, desc)
