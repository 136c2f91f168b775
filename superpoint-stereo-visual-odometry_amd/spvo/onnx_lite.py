"""Minimal ONNX reader (protobuf wire format only, no `onnx` package needed).

Replaces the model-ingest half of the reference's TensorRT engine generator
(reference: src/odml_visual_odometry/scripts/engine_generation.py:25-56, which
shells out to `trtexec --onnx=...`).  Only what the SuperPoint graphs in
src/odml_visual_odometry/models/*.onnx use is decoded: ModelProto.graph,
GraphProto.{node,initializer,input,output}, NodeProto, AttributeProto
(f, i, s, t, floats, ints) and TensorProto (dims, data_type, raw_data,
float_data, int64_data).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

import numpy as np


def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    result = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        result |= (b & 0x7F) << shift
        if not b & 0x80:
            return result, pos
        shift += 7


def _fields(buf: bytes):
    """Yield (field_number, wire_type, value) for one message body."""
    pos = 0
    n = len(buf)
    while pos < n:
        key, pos = _varint(buf, pos)
        fnum, wt = key >> 3, key & 7
        if wt == 0:
            val, pos = _varint(buf, pos)
        elif wt == 1:
            val = buf[pos:pos + 8]
            pos += 8
        elif wt == 2:
            ln, pos = _varint(buf, pos)
            val = buf[pos:pos + ln]
            pos += ln
        elif wt == 5:
            val = buf[pos:pos + 4]
            pos += 4
        else:
            raise ValueError(f"unsupported wire type {wt}")
        yield fnum, wt, val


def _signed(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(val: bytes) -> List[int]:
    out = []
    pos = 0
    while pos < len(val):
        v, pos = _varint(val, pos)
        out.append(_signed(v))
    return out


_DTYPES = {1: np.float32, 6: np.int32, 7: np.int64, 10: np.float16, 11: np.float64}


def _tensor(buf: bytes) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype = 1
    name = ""
    raw = None
    floats: List[float] = []
    int64s: List[int] = []
    for f, wt, v in _fields(buf):
        if f == 1:
            dims += _packed_varints(v) if wt == 2 else [_signed(v)]
        elif f == 2:
            dtype = v
        elif f == 8:
            name = v.decode()
        elif f == 9:
            raw = v
        elif f == 4:
            if wt == 2:
                floats += list(struct.unpack(f"<{len(v) // 4}f", v))
            else:
                floats.append(struct.unpack("<f", v)[0])
        elif f == 7:
            int64s += _packed_varints(v) if wt == 2 else [_signed(v)]
    np_dt = _DTYPES[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np_dt).copy()
    elif floats:
        arr = np.asarray(floats, dtype=np_dt)
    else:
        arr = np.asarray(int64s, dtype=np_dt)
    return name, arr.reshape(dims) if dims else arr


@dataclass
class Node:
    op: str
    name: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)


def _attr(buf: bytes) -> Tuple[str, object]:
    name = ""
    val: object = None
    ints: List[int] = []
    floats: List[float] = []
    for f, wt, v in _fields(buf):
        if f == 1:
            name = v.decode()
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = v.decode(errors="replace")
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 7:
            floats += list(struct.unpack(f"<{len(v) // 4}f", v)) if wt == 2 else [struct.unpack("<f", v)[0]]
        elif f == 8:
            ints += _packed_varints(v) if wt == 2 else [_signed(v)]
    if ints:
        val = ints
    elif floats:
        val = floats
    return name, val


def _node(buf: bytes) -> Node:
    n = Node("", "", [], [])
    for f, _, v in _fields(buf):
        if f == 1:
            n.inputs.append(v.decode())
        elif f == 2:
            n.outputs.append(v.decode())
        elif f == 3:
            n.name = v.decode()
        elif f == 4:
            n.op = v.decode()
        elif f == 5:
            k, a = _attr(v)
            n.attrs[k] = a
    return n


def _value_info_name(buf: bytes) -> str:
    for f, _, v in _fields(buf):
        if f == 1:
            return v.decode()
    return ""


@dataclass
class Graph:
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]
    inputs: List[str]
    outputs: List[str]


def load(path: str) -> Graph:
    with open(path, "rb") as fh:
        buf = fh.read()
    graph_buf = None
    for f, _, v in _fields(buf):
        if f == 7:
            graph_buf = v
    if graph_buf is None:
        raise ValueError(f"{path}: no GraphProto found")
    g = Graph([], {}, [], [])
    for f, _, v in _fields(graph_buf):
        if f == 1:
            g.nodes.append(_node(v))
        elif f == 5:
            name, arr = _tensor(v)
            g.initializers[name] = arr
        elif f == 11:
            g.inputs.append(_value_info_name(v))
        elif f == 12:
            g.outputs.append(_value_info_name(v))
    g.inputs = [i for i in g.inputs if i not in g.initializers]
    return g
