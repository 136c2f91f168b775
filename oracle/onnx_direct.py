"""Direct evaluation of an ONNX graph, node by node, with nothing of the product in between.

TEST INFRASTRUCTURE ONLY (see oracle/__init__.py).  oracle/net.py executes the product's execution plan (spvo.weights.Plan:
ReLU / BatchNorm / Add / MaxPool folded into the producing convolution, Concat turned into channel offsets), so a mistake
of the plan packer -- a BatchNorm epsilon, a pad, the order of a Concat -- would be shared by the oracle, the C++
restatement and the HIP kernels alike.  This module is the independent check: its own protobuf wire-format reader (the
`onnx` package is not installed; nothing is shared with spvo/onnx_lite.py) and a literal interpreter of the eight operator
types the reference's graphs use {Conv, Relu, MaxPool, BatchNormalization, Add, Concat, ReduceL2, Div}
(src/odml_visual_odometry/models/sp_*.onnx, opset 11-12), evaluated with torch-CPU fp32 -- what the TensorRT engine of
feature_detection_neural_network.cpp:163-176 is built from.  tests/golden/make_onnx_direct_golden.py freezes its
outputs as fixtures (the ONNX files live under /root/reference and do not travel); tests compare the plan-based
evaluations with them.
"""
from __future__ import annotations

import struct
from typing import Dict, List, Tuple

import numpy as np


# ------------------------------------------------------------------ protobuf wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes):
    """Yields (field number, wire type, value) of one message; length-delimited values come as bytes."""
    pos, n = 0, len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        num, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            v = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            v = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"wire type {wt}")
        yield num, wt, v


def _packed_ints(v: bytes) -> List[int]:
    out, pos = [], 0
    while pos < len(v):
        x, pos = _varint(v, pos)
        out.append(x if x < (1 << 63) else x - (1 << 64))
    return out


def _tensor(buf: bytes):
    dims: List[int] = []
    dtype, name, raw, floats, int64s = 0, "", None, [], []
    for num, wt, v in _fields(buf):
        if num == 1:
            dims += _packed_ints(v) if wt == 2 else [v]
        elif num == 2:
            dtype = v
        elif num == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif num == 7:
            int64s += _packed_ints(v) if wt == 2 else [v]
        elif num == 8:
            name = v.decode()
        elif num == 9:
            raw = v
    if dtype == 1:
        a = np.frombuffer(raw, "<f4").copy() if raw is not None else np.asarray(floats, np.float32)
    elif dtype == 7:
        a = np.frombuffer(raw, "<i8").copy() if raw is not None else np.asarray(int64s, np.int64)
    else:
        raise ValueError(f"tensor {name}: data type {dtype}")
    return name, a.reshape(dims)


def _attribute(buf: bytes):
    name, val, ints, floats = "", None, [], []
    for num, wt, v in _fields(buf):
        if num == 1:
            name = v.decode()
        elif num == 2:
            val = struct.unpack("<f", v)[0]
        elif num == 3:
            val = v if v < (1 << 63) else v - (1 << 64)
        elif num == 4:
            val = v.decode()
        elif num == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif num == 8:
            ints += _packed_ints(v) if wt == 2 else [v]
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def read_graph(path: str):
    """Returns (nodes, initializers, graph input names, graph output names); node = (op_type, inputs, outputs, attrs)."""
    model = open(path, "rb").read()
    graph = next(v for num, _, v in _fields(model) if num == 7)
    nodes, inits, ins, outs = [], {}, [], []
    for num, _, v in _fields(graph):
        if num == 1:
            op, i, o, at = "", [], [], {}
            for n2, _, v2 in _fields(v):
                if n2 == 1:
                    i.append(v2.decode())
                elif n2 == 2:
                    o.append(v2.decode())
                elif n2 == 4:
                    op = v2.decode()
                elif n2 == 5:
                    k, val = _attribute(v2)
                    at[k] = val
            nodes.append((op, i, o, at))
        elif num == 5:
            name, a = _tensor(v)
            inits[name] = a
        elif num in (11, 12):
            name = next(v2.decode() for n2, _, v2 in _fields(v) if n2 == 1)
            (ins if num == 11 else outs).append(name)
    ins = [n for n in ins if n not in inits]
    return nodes, inits, ins, outs


# ------------------------------------------------------------------ interpreter
def run(path: str, x: np.ndarray) -> Dict[str, np.ndarray]:
    """x: float32 [B,1,H,W] -> {graph output name: array}.  ONNX operator semantics, literally."""
    import torch
    import torch.nn.functional as F
    torch.set_grad_enabled(False)
    nodes, inits, ins, outs = read_graph(path)
    assert len(ins) == 1, ins
    vals = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in inits.items()}
    vals[ins[0]] = torch.from_numpy(np.ascontiguousarray(x, np.float32))
    for op, i, o, at in nodes:
        a = [vals[n] for n in i]
        if op == "Conv":
            pads = at.get("pads", [0, 0, 0, 0])
            assert pads[0] == pads[2] and pads[1] == pads[3], pads          # symmetric: F.conv2d's padding
            assert at.get("dilations", [1, 1]) == [1, 1] and at.get("auto_pad", "NOTSET") == "NOTSET"
            y = F.conv2d(a[0], a[1], a[2] if len(a) > 2 else None, stride=tuple(at.get("strides", [1, 1])),
                         padding=(pads[0], pads[1]), groups=at.get("group", 1))
        elif op == "Relu":
            y = F.relu(a[0])
        elif op == "MaxPool":
            assert at.get("pads", [0, 0, 0, 0]) == [0, 0, 0, 0] and at.get("ceil_mode", 0) == 0
            y = F.max_pool2d(a[0], tuple(at["kernel_shape"]), tuple(at.get("strides", at["kernel_shape"])))
        elif op == "BatchNormalization":   # inference form: scale * (x - mean) / sqrt(var + epsilon) + B
            eps = at.get("epsilon", 1e-5)
            sc, b, mean, var = (t.reshape(1, -1, 1, 1) for t in a[1:5])
            y = sc * (a[0] - mean) / torch.sqrt(var + eps) + b
        elif op == "Add":
            y = a[0] + a[1]
        elif op == "Concat":
            y = torch.cat(a, dim=at["axis"])
        elif op == "ReduceL2":
            assert at.get("keepdims", 1) == 1
            y = torch.sqrt((a[0] * a[0]).sum(dim=tuple(at["axes"]), keepdim=True))
        elif op == "Div":
            y = a[0] / a[1]
        else:
            raise NotImplementedError(op)
        vals[o[0]] = y
    return {n: vals[n].numpy() for n in outs}
