"""Weight / network-plan packer: ONNX (or the VGG definition) -> one `.spvw` file.

This replaces the reference's offline TensorRT "engine generation" step
(reference: src/odml_visual_odometry/scripts/engine_generation.py:13-56) and its
engine naming convention (reference: src/odml_visual_odometry/src/
feature_detection_neural_network.cpp:44-49).  A `.spvw` file holds a tiny
execution plan (tensors + ops) and the raw fp32 OIHW weights; the C-ABI library
(`spvo_load_weights`) repacks the weights on the device into the MFMA kernel's
own layout, so the file stays canonical and human-checkable.

File layout (little endian):
    char  magic[8]  = b"SPVW0003"
    u32   n_tensors, n_ops, input_tensor, det_tensor, desc_tensor, precision (0 = FP32, 1 = FP16, 2 = INT8),
          act_scale_off (INT8: offset in floats of n_tensors activation scales in the payload), reserved
    n_tensors x { u32 channels, u32 level }          level = log2(downscale)
    n_ops     x { u32 type, in, out, out_c_off, cin | in_c_off << 16, cout, ksize, flags,
                  residual_tensor, reserved[3]; u64 w_off, b_off, bn_off }   offsets in floats
    u64   payload_floats
    f32   payload[payload_floats]

Op types: 1 = CONV (ksize 1|3, pad ksize//2, stride 1), 2 = MAXPOOL2x2,
          3 = L2NORM over channels (descriptor tail: ReduceL2 + Div, no epsilon),
          4 = DWCONV (depthwise 3x3, pad 1: the MobileNet graphs).
Epilogue of CONV / DWCONV, in this order (flags):
          bit0 ReLU after bias; bit2 BatchNorm (gamma, beta, mean, var [cout] each + eps at bn_off)
          followed by ReLU (mbv1 puts BN AFTER the activation, so it cannot be folded into the conv);
          bit3 add `residual_tensor` then ReLU (mbv2); bit1 2x2/2 max-pool last.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import onnx_lite

MAGIC = b"SPVW0003"
OP_CONV, OP_MAXPOOL, OP_L2NORM, OP_DWCONV = 1, 2, 3, 4
FLAG_RELU, FLAG_POOL, FLAG_BN, FLAG_ADD = 1, 2, 4, 8

# reference: feature_detection.hpp:355-359
DET_CHANNELS, DESC_CHANNELS, CELL = 65, 256, 8


@dataclass
class Op:
    type: int
    inp: int
    out: int
    out_c_off: int = 0
    cin: int = 0
    cout: int = 0
    ksize: int = 0
    flags: int = 0
    weight: Optional[np.ndarray] = None  # OIHW fp32
    bias: Optional[np.ndarray] = None
    in_c_off: int = 0
    residual: int = 0                    # tensor added before the final ReLU (FLAG_ADD)
    bn: Optional[np.ndarray] = None      # [4*cout + 1]: gamma, beta, mean, var, eps (FLAG_BN)


@dataclass
class Plan:
    tensors: List[Tuple[int, int]] = field(default_factory=list)  # (channels, level)
    ops: List[Op] = field(default_factory=list)
    input_tensor: int = 0
    det_tensor: int = 0
    desc_tensor: int = 0
    precision: str = "FP32"   # "FP16": the library keeps fp16 between the fp32 network input and the fp32 outputs
    act_scales: Optional[np.ndarray] = None   # "INT8": one symmetric scale per tensor from calibration (real = q * scale)

    def add_tensor(self, channels: int, level: int) -> int:
        self.tensors.append((channels, level))
        return len(self.tensors) - 1

    def n_params(self) -> int:
        return sum(op.weight.size + op.bias.size for op in self.ops if op.weight is not None)


# --------------------------------------------------------------------------
# VGG SuperPoint (MagicLeap layer shapes).  The real
# weights are missing from the reference tree (.MISSING_LARGE_BLOBS:14-15), so
# the values are seeded synthetic; the parameter count (1 300 865) matches the
# reference's report, Table 1.
# --------------------------------------------------------------------------
VGG_LAYERS = [
    # name, cin, cout, k, relu, pool_after
    ("conv1a", 1, 64, 3, True, False),
    ("conv1b", 64, 64, 3, True, True),
    ("conv2a", 64, 64, 3, True, False),
    ("conv2b", 64, 64, 3, True, True),
    ("conv3a", 64, 128, 3, True, False),
    ("conv3b", 128, 128, 3, True, True),
    ("conv4a", 128, 128, 3, True, False),
    ("conv4b", 128, 128, 3, True, False),
]


def vgg_synthetic_weights(seed: int = 0, dustbin_bias: float = 4.7) -> Dict[str, Tuple[np.ndarray, np.ndarray]]:
    """He-normal conv weights, zero biases, one deterministic stream per layer.

    `dustbin_bias` lifts the detector's 65th ("no keypoint") logit so that the
    fraction of pixels above conf_thresh=0.015 is in the range real SuperPoint
    heads give on KITTI imagery (about 3 %); with 0 every cell would hover at the
    uniform 1/65 = 0.0154 and nearly half of all pixels would be candidates.
    """
    out: Dict[str, Tuple[np.ndarray, np.ndarray]] = {}
    shapes = [(n, ci, co, k) for (n, ci, co, k, _, _) in VGG_LAYERS]
    shapes += [("convPa", 128, 256, 3), ("convPb", 256, DET_CHANNELS, 1),
               ("convDa", 128, 256, 3), ("convDb", 256, DESC_CHANNELS, 1)]
    for idx, (name, ci, co, k) in enumerate(shapes):
        rng = np.random.RandomState(seed * 1000 + idx)
        std = np.sqrt(2.0 / (ci * k * k))
        w = (rng.standard_normal((co, ci, k, k)) * std).astype(np.float32)
        b = np.zeros((co,), np.float32)
        if name == "convPb":
            b[DET_CHANNELS - 1] = dustbin_bias
        out[name] = (w, b)
    return out


def vgg_plan(seed: int = 0, dustbin_bias: float = 4.7) -> Plan:
    wts = vgg_synthetic_weights(seed, dustbin_bias)
    p = Plan()
    cur = p.add_tensor(1, 0)
    p.input_tensor = cur
    level = 0
    for name, ci, co, k, relu, pool in VGG_LAYERS:
        if pool:
            level += 1
        nxt = p.add_tensor(co, level)
        w, b = wts[name]
        p.ops.append(Op(OP_CONV, cur, nxt, 0, ci, co, k,
                        (FLAG_RELU if relu else 0) | (FLAG_POOL if pool else 0), w, b))
        cur = nxt
    assert level == 3
    # convPa and convDa read the same tensor: one 128->512 conv feeding both heads
    heads = p.add_tensor(512, 3)
    wpa, bpa = wts["convPa"]
    wda, bda = wts["convDa"]
    p.ops.append(Op(OP_CONV, cur, heads, 0, 128, 256, 3, FLAG_RELU, wpa, bpa))
    p.ops.append(Op(OP_CONV, cur, heads, 256, 128, 256, 3, FLAG_RELU, wda, bda))
    det = p.add_tensor(DET_CHANNELS, 3)
    draw = p.add_tensor(DESC_CHANNELS, 3)
    desc = p.add_tensor(DESC_CHANNELS, 3)
    wpb, bpb = wts["convPb"]
    wdb, bdb = wts["convDb"]
    p.ops.append(Op(OP_CONV, heads, det, 0, 256, DET_CHANNELS, 1, 0, wpb, bpb))
    p.ops[-1].in_c_off = 0
    p.ops.append(Op(OP_CONV, heads, draw, 0, 256, DESC_CHANNELS, 1, 0, wdb, bdb))
    p.ops[-1].in_c_off = 256
    p.ops.append(Op(OP_L2NORM, draw, desc, 0, DESC_CHANNELS, DESC_CHANNELS))
    p.det_tensor, p.desc_tensor = det, desc
    return p


# --------------------------------------------------------------------------
# ONNX graph -> plan.  Ops present in the reference's models: Conv (dense 1x1 / 3x3 and depthwise
# 3x3), Relu, MaxPool, Concat (squeeze), BatchNormalization after Relu (mbv1), Add (mbv2),
# ReduceL2, Div.  Every Relu / BatchNormalization / Add / MaxPool that directly follows a
# convolution is fused into that convolution's epilogue.
# --------------------------------------------------------------------------
_ST_BIAS, _ST_RELU, _ST_BN, _ST_BN_RELU, _ST_ADD, _ST_ADD_RELU, _ST_POOL = range(7)


def onnx_plan(path: str) -> Plan:
    g = onnx_lite.load(path)
    p = Plan()
    consumers: Dict[str, List[onnx_lite.Node]] = {}
    for n in g.nodes:
        for i in n.inputs:
            consumers.setdefault(i, []).append(n)
    produced_by = {o: n for n in g.nodes for o in n.outputs}
    tid: Dict[str, Tuple[int, int]] = {}   # materialised value -> (tensor id, channel offset)
    level_of: Dict[int, int] = {}
    p.input_tensor = p.add_tensor(1, 0)
    tid[g.inputs[0]] = (p.input_tensor, 0)
    level_of[p.input_tensor] = 0

    def channels_of(name: str) -> int:
        n = produced_by[name]
        if n.op == "Conv":
            return g.initializers[n.inputs[1]].shape[0]
        if n.op == "Concat":
            return sum(channels_of(i) for i in n.inputs)
        return channels_of(n.inputs[0])

    concat_slot: Dict[str, Tuple[str, int]] = {}  # value -> (concat output, channel offset)
    for n in g.nodes:
        if n.op == "Concat":
            off = 0
            for i in n.inputs:
                concat_slot[i] = (n.outputs[0], off)
                off += channels_of(i)

    pending: Dict[str, Tuple[Op, int]] = {}   # not yet materialised value -> (op, epilogue stage)

    def can_fuse(value: str, stage: int) -> bool:
        nxt = consumers.get(value, [])
        if len(nxt) != 1 or value in ("output_det", "output_desc") or value in concat_slot:
            return False
        n = nxt[0]
        if n.op == "Relu":
            return stage in (_ST_BIAS, _ST_BN, _ST_ADD)
        if n.op == "BatchNormalization":
            return stage == _ST_RELU
        if n.op == "Add":
            other = [i for i in n.inputs if i != value]
            return stage == _ST_BIAS and len(other) == 1 and other[0] in tid
        if n.op == "MaxPool":
            return stage in (_ST_RELU, _ST_BN_RELU, _ST_ADD_RELU)
        return False

    def settle(value: str, op: Op, stage: int):
        """Keep fusing, or materialise `value` as the output tensor of `op`."""
        if can_fuse(value, stage):
            pending[value] = (op, stage)
            return
        assert stage not in (_ST_BN, _ST_ADD), "BatchNormalization / Add must be followed by Relu in these graphs"
        lvl = level_of[op.inp] + (1 if op.flags & FLAG_POOL else 0)
        if value in concat_slot:
            cname, off = concat_slot[value]
            if cname not in tid:
                t = p.add_tensor(channels_of(cname), lvl)
                level_of[t] = lvl
                tid[cname] = (t, 0)
            op.out, op.out_c_off = tid[cname][0], off
            tid[value] = (op.out, off)
        else:
            t = p.add_tensor(op.cout, lvl)
            level_of[t] = lvl
            op.out, op.out_c_off = t, 0
            tid[value] = (t, 0)

    for n in g.nodes:
        if n.op == "Conv":
            w = g.initializers[n.inputs[1]].astype(np.float32)
            b = g.initializers[n.inputs[2]].astype(np.float32)
            k = w.shape[2]
            assert n.attrs["strides"] == [1, 1] and n.attrs["pads"] == [k // 2] * 4
            src, src_off = tid[n.inputs[0]]
            group = n.attrs.get("group", 1)
            if group == 1:
                op = Op(OP_CONV, src, -1, 0, w.shape[1], w.shape[0], k, 0, w, b)
            else:
                assert group == w.shape[0] and w.shape[1] == 1 and k == 3, "only depthwise 3x3 grouped convs occur"
                op = Op(OP_DWCONV, src, -1, 0, w.shape[0], w.shape[0], k, 0, w, b)
            op.in_c_off = src_off
            p.ops.append(op)
            settle(n.outputs[0], op, _ST_BIAS)
        elif n.op == "Relu":
            op, stage = pending.pop(n.inputs[0])
            if stage == _ST_BIAS:
                op.flags |= FLAG_RELU
                settle(n.outputs[0], op, _ST_RELU)
            else:
                settle(n.outputs[0], op, _ST_BN_RELU if stage == _ST_BN else _ST_ADD_RELU)
        elif n.op == "BatchNormalization":
            op, stage = pending.pop(n.inputs[0])
            gam, bet, mean, var = [g.initializers[i].astype(np.float32) for i in n.inputs[1:5]]
            op.bn = np.concatenate([gam, bet, mean, var, np.array([n.attrs.get("epsilon", 1e-5)], np.float32)])
            op.flags |= FLAG_BN
            settle(n.outputs[0], op, _ST_BN)
        elif n.op == "Add":
            pend = [i for i in n.inputs if i in pending]
            assert len(pend) == 1, "Add must combine one convolution output with one stored tensor"
            op, stage = pending.pop(pend[0])
            other = [i for i in n.inputs if i != pend[0]][0]
            rt, roff = tid[other]
            assert roff == 0 and p.tensors[rt][0] == op.cout and level_of[rt] == level_of[op.inp]
            op.residual = rt
            op.flags |= FLAG_ADD
            settle(n.outputs[0], op, _ST_ADD)
        elif n.op == "MaxPool":
            assert n.attrs["kernel_shape"] == [2, 2] and n.attrs["strides"] == [2, 2] and not n.attrs.get("ceil_mode", 0)
            if n.inputs[0] in pending:
                op, stage = pending.pop(n.inputs[0])
                op.flags |= FLAG_POOL
                settle(n.outputs[0], op, _ST_POOL)
            else:
                src, src_off = tid[n.inputs[0]]
                assert src_off == 0
                ch = p.tensors[src][0]
                dst = p.add_tensor(ch, level_of[src] + 1)
                level_of[dst] = level_of[src] + 1
                p.ops.append(Op(OP_MAXPOOL, src, dst, 0, ch, ch))
                tid[n.outputs[0]] = (dst, 0)
        elif n.op == "Concat":
            assert n.outputs[0] in tid, "concat inputs must all be conv outputs"
        elif n.op == "ReduceL2":
            pass
        elif n.op == "Div":
            src, _ = tid[n.inputs[0]]
            ch = p.tensors[src][0]
            dst = p.add_tensor(ch, level_of[src])
            level_of[dst] = level_of[src]
            p.ops.append(Op(OP_L2NORM, src, dst, 0, ch, ch))
            tid[n.outputs[0]] = (dst, 0)
        else:
            raise NotImplementedError(f"ONNX op {n.op}")
    assert not pending, f"unmaterialised values: {list(pending)}"
    p.det_tensor = tid["output_det"][0]
    p.desc_tensor = tid["output_desc"][0]
    for op in p.ops:
        assert op.out >= 0, "unfinalised conv"
    return p


# --------------------------------------------------------------------------
PRECISIONS = {"FP32": 0, "FP16": 1, "INT8": 2}


def save(plan: Plan, path: str, precision: Optional[str] = None) -> None:
    """Weights stay canonical fp32 in the file for every precision (the library rounds them at load, to nearest even);
    `precision` (default: the plan's) only selects the engine the library builds, like trtexec's --fp16."""
    precision = precision or plan.precision
    payload: List[np.ndarray] = []
    off = 0
    recs = []
    for op in plan.ops:
        w_off = b_off = bn_off = 0
        if op.weight is not None:
            w = np.ascontiguousarray(op.weight, np.float32).ravel()
            b = np.ascontiguousarray(op.bias, np.float32).ravel()
            w_off, off = off, off + w.size
            b_off, off = off, off + b.size
            payload += [w, b]
        if op.bn is not None:
            bn = np.ascontiguousarray(op.bn, np.float32).ravel()
            bn_off, off = off, off + bn.size
            payload.append(bn)
        recs.append(struct.pack("<12I3Q", op.type, op.inp, op.out, op.out_c_off, op.cin | (op.in_c_off << 16), op.cout,
                                op.ksize, op.flags, op.residual, 0, 0, 0, w_off, b_off, bn_off))
    act_off = 0
    if precision == "INT8":
        assert plan.act_scales is not None and len(plan.act_scales) == len(plan.tensors), "INT8 engines need calibrated activation scales"
        act_off, off = off, off + len(plan.tensors)
        payload.append(np.ascontiguousarray(plan.act_scales, np.float32))
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<8I", len(plan.tensors), len(plan.ops), plan.input_tensor,
                             plan.det_tensor, plan.desc_tensor, PRECISIONS[precision], act_off, 0))
        for ch, lvl in plan.tensors:
            fh.write(struct.pack("<2I", ch, lvl))
        for r in recs:
            fh.write(r)
        fh.write(struct.pack("<Q", off))
        for a in payload:
            fh.write(a.tobytes())


def load(path: str) -> Plan:
    with open(path, "rb") as fh:
        buf = fh.read()
    assert buf[:8] == MAGIC, "not a .spvw file of this version"
    nt, no, it, dt, st, prec, act_off, _ = struct.unpack_from("<8I", buf, 8)
    pos = 40
    p = Plan(input_tensor=it, det_tensor=dt, desc_tensor=st, precision={v: k for k, v in PRECISIONS.items()}[prec])
    for _ in range(nt):
        ch, lvl = struct.unpack_from("<2I", buf, pos)
        pos += 8
        p.tensors.append((ch, lvl))
    raw_ops = []
    for _ in range(no):
        raw_ops.append(struct.unpack_from("<12I3Q", buf, pos))
        pos += 72
    (nfl,) = struct.unpack_from("<Q", buf, pos)
    pos += 8
    payload = np.frombuffer(buf, np.float32, nfl, pos)
    for t, i, o, oc, cin, cout, k, fl, res, _r0, _r1, _r2, wo, bo, bno in raw_ops:
        op = Op(t, i, o, oc, cin & 0xFFFF, cout, k, fl)
        op.in_c_off = cin >> 16
        op.residual = res
        if t == OP_CONV:
            n = cout * op.cin * k * k
            op.weight = payload[wo:wo + n].reshape(cout, op.cin, k, k).copy()
            op.bias = payload[bo:bo + cout].copy()
        elif t == OP_DWCONV:
            op.weight = payload[wo:wo + cout * 9].reshape(cout, 1, 3, 3).copy()
            op.bias = payload[bo:bo + cout].copy()
        if fl & FLAG_BN:
            op.bn = payload[bno:bno + 4 * cout + 1].copy()
        p.ops.append(op)
    if p.precision == "INT8":
        p.act_scales = payload[act_off:act_off + nt].copy()
    return p


def engine_name(prefix: str, batch: int, height: int, width: int, precision: str) -> str:
    """Same naming rule as the reference's engines (neural_network.cpp:44-49), new suffix."""
    return f"{prefix}_{batch}_{height}_{width}_{precision}.spvw"
