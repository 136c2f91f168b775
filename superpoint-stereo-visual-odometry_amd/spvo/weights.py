"""Weight / network-plan packer: ONNX (or the VGG definition) -> one `.spvw` file.

This replaces the reference's offline TensorRT "engine generation" step
(reference: src/odml_visual_odometry/scripts/engine_generation.py:13-56) and its
engine naming convention (reference: src/odml_visual_odometry/src/
feature_detection_neural_network.cpp:44-49).  A `.spvw` file holds a tiny
execution plan (tensors + ops) and the raw fp32 OIHW weights; the C-ABI library
(`spvo_load_weights`) repacks the weights on the device into the MFMA kernel's
own layout, so the file stays canonical and human-checkable.

File layout (little endian):
    char  magic[8]  = b"SPVW0002"
    u32   n_tensors, n_ops, input_tensor, det_tensor, desc_tensor, reserved[3]
    n_tensors x { u32 channels, u32 level }          level = log2(downscale)
    n_ops     x { u32 type, in, out, out_c_off, cin, cout, ksize, flags;
                  u64 w_off, b_off }                 offsets in floats into payload
    u64   payload_floats
    f32   payload[payload_floats]

Op types: 1 = CONV (ksize 1|3, pad ksize//2, stride 1), 2 = MAXPOOL2x2,
          3 = L2NORM over channels (descriptor tail: ReduceL2 + Div, no epsilon).
Flags:    bit0 = ReLU after bias, bit1 = fused 2x2/2 max-pool after ReLU.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import onnx_lite

MAGIC = b"SPVW0002"
OP_CONV, OP_MAXPOOL, OP_L2NORM = 1, 2, 3
FLAG_RELU, FLAG_POOL = 1, 2

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


@dataclass
class Plan:
    tensors: List[Tuple[int, int]] = field(default_factory=list)  # (channels, level)
    ops: List[Op] = field(default_factory=list)
    input_tensor: int = 0
    det_tensor: int = 0
    desc_tensor: int = 0

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
# ONNX graph -> plan (ops present in the reference's models: Conv, Relu,
# MaxPool, Concat, ReduceL2, Div; BatchNormalization / Add / depthwise are
# the mbv1 / mbv2 graphs and are rejected here until their kernels exist).
# --------------------------------------------------------------------------
def onnx_plan(path: str) -> Plan:
    g = onnx_lite.load(path)
    p = Plan()
    consumers: Dict[str, List[onnx_lite.Node]] = {}
    for n in g.nodes:
        for i in n.inputs:
            consumers.setdefault(i, []).append(n)
    tid: Dict[str, Tuple[int, int]] = {}  # value name -> (tensor id, channel offset)
    level_of: Dict[int, int] = {}
    p.input_tensor = p.add_tensor(1, 0)
    tid[g.inputs[0]] = (p.input_tensor, 0)
    level_of[p.input_tensor] = 0

    # pre-assign concat outputs so producers can write straight into their slice
    produced_by = {o: n for n in g.nodes for o in n.outputs}

    def channels_of(name: str) -> int:
        n = produced_by[name]
        if n.op == "Conv":
            return g.initializers[n.inputs[1]].shape[0]
        if n.op in ("Relu", "MaxPool", "Div"):
            return channels_of(n.inputs[0])
        if n.op == "Concat":
            return sum(channels_of(i) for i in n.inputs)
        raise NotImplementedError(n.op)

    concat_slot: Dict[str, Tuple[str, int]] = {}  # value -> (concat output, offset)
    for n in g.nodes:
        if n.op == "Concat":
            off = 0
            for i in n.inputs:
                concat_slot[i] = (n.outputs[0], off)
                off += channels_of(i)

    pending_conv: Dict[str, Op] = {}
    for n in g.nodes:
        if n.op == "Conv":
            w = g.initializers[n.inputs[1]].astype(np.float32)
            b = g.initializers[n.inputs[2]].astype(np.float32)
            if n.attrs.get("group", 1) != 1:
                raise NotImplementedError("grouped/depthwise conv (mbv1/mbv2) has no kernel yet")
            k = w.shape[2]
            assert n.attrs["strides"] == [1, 1] and n.attrs["pads"] == [k // 2] * 4
            src, src_off = tid[n.inputs[0]]
            op = Op(OP_CONV, src, -1, 0, w.shape[1], w.shape[0], k, 0, w, b)
            op.in_c_off = src_off
            pending_conv[n.outputs[0]] = op
            p.ops.append(op)
            _finalize_if_terminal(p, g, n.outputs[0], op, consumers, concat_slot, tid, level_of)
        elif n.op == "Relu":
            op = pending_conv.pop(n.inputs[0])
            op.flags |= FLAG_RELU
            pending_conv[n.outputs[0]] = op
            _finalize_if_terminal(p, g, n.outputs[0], op, consumers, concat_slot, tid, level_of)
        elif n.op == "MaxPool":
            assert n.attrs["kernel_shape"] == [2, 2] and n.attrs["strides"] == [2, 2]
            if n.inputs[0] in pending_conv:
                op = pending_conv.pop(n.inputs[0])
                op.flags |= FLAG_POOL
                pending_conv[n.outputs[0]] = op
                _finalize_if_terminal(p, g, n.outputs[0], op, consumers, concat_slot, tid, level_of)
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
    p.det_tensor = tid["output_det"][0]
    p.desc_tensor = tid["output_desc"][0]
    for op in p.ops:
        assert op.out >= 0, "unfinalised conv"
    return p


def _finalize_if_terminal(p, g, value, op, consumers, concat_slot, tid, level_of):
    """Give `op` its output tensor once no more Relu/MaxPool can be fused onto `value`."""
    nxt = consumers.get(value, [])
    fuse_relu = len(nxt) == 1 and nxt[0].op == "Relu" and not (op.flags & (FLAG_RELU | FLAG_POOL))
    fuse_pool = len(nxt) == 1 and nxt[0].op == "MaxPool" and not (op.flags & FLAG_POOL)
    if fuse_relu or fuse_pool:
        return
    lvl = level_of[op.inp] + (1 if op.flags & FLAG_POOL else 0)
    if value in concat_slot:
        cname, off = concat_slot[value]
        if cname not in tid:
            total = off
            # total channel count = sum over all members of this concat
            total = sum(c for c in _concat_members(g, cname))
            t = p.add_tensor(total, lvl)
            level_of[t] = lvl
            tid[cname] = (t, 0)
        op.out, op.out_c_off = tid[cname][0], off
        tid[value] = (op.out, off)
    else:
        t = p.add_tensor(op.cout, lvl)
        level_of[t] = lvl
        op.out, op.out_c_off = t, 0
        tid[value] = (t, 0)


def _concat_members(g, cname):
    produced_by = {o: n for n in g.nodes for o in n.outputs}
    node = produced_by[cname]

    def ch(name):
        n = produced_by[name]
        if n.op == "Conv":
            return g.initializers[n.inputs[1]].shape[0]
        return ch(n.inputs[0])

    return [ch(i) for i in node.inputs]


# --------------------------------------------------------------------------
def save(plan: Plan, path: str) -> None:
    payload: List[np.ndarray] = []
    off = 0
    recs = []
    for op in plan.ops:
        w_off = b_off = 0
        if op.weight is not None:
            w = np.ascontiguousarray(op.weight, np.float32).ravel()
            b = np.ascontiguousarray(op.bias, np.float32).ravel()
            w_off, off = off, off + w.size
            b_off, off = off, off + b.size
            payload += [w, b]
        in_c_off = getattr(op, "in_c_off", 0)
        # the input channel offset travels in the upper half of `cin`
        recs.append(struct.pack("<8I2Q", op.type, op.inp, op.out, op.out_c_off,
                                op.cin | (in_c_off << 16), op.cout, op.ksize, op.flags, w_off, b_off))
    with open(path, "wb") as fh:
        fh.write(MAGIC)
        fh.write(struct.pack("<8I", len(plan.tensors), len(plan.ops), plan.input_tensor,
                             plan.det_tensor, plan.desc_tensor, 0, 0, 0))
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
    assert buf[:8] == MAGIC, "not a .spvw file"
    nt, no, it, dt, st, _, _, _ = struct.unpack_from("<8I", buf, 8)
    pos = 40
    p = Plan(input_tensor=it, det_tensor=dt, desc_tensor=st)
    for _ in range(nt):
        ch, lvl = struct.unpack_from("<2I", buf, pos)
        pos += 8
        p.tensors.append((ch, lvl))
    raw_ops = []
    for _ in range(no):
        raw_ops.append(struct.unpack_from("<8I2Q", buf, pos))
        pos += 48
    (nfl,) = struct.unpack_from("<Q", buf, pos)
    pos += 8
    payload = np.frombuffer(buf, np.float32, nfl, pos)
    for t, i, o, oc, cin, cout, k, fl, wo, bo in raw_ops:
        op = Op(t, i, o, oc, cin & 0xFFFF, cout, k, fl)
        op.in_c_off = cin >> 16
        if t == OP_CONV:
            n = cout * op.cin * k * k
            op.weight = payload[wo:wo + n].reshape(cout, op.cin, k, k).copy()
            op.bias = payload[bo:bo + cout].copy()
        p.ops.append(op)
    return p


def engine_name(prefix: str, batch: int, height: int, width: int, precision: str) -> str:
    """Same naming rule as the reference's engines (neural_network.cpp:44-49), new suffix."""
    return f"{prefix}_{batch}_{height}_{width}_{precision}.spvw"
