"""Dependency-free reader for the ``.onnx`` files the reference deploys.

The reference's deployable artefact is the ONNX file written by ``export_onnx_model``
(reference: nanowakeword/_export/onnx.py:157-229, ``torch.onnx.export`` of torch 2.8 = the TorchScript exporter,
opset 17, input ``input`` / output ``output``); its interpreter feeds it to onnxruntime
(nanowakeword/interpreter/nanointerpreter.py:955-959).  The ``onnx`` package is not required here: an ONNX file
is a protobuf ``ModelProto`` and this module decodes the wire format directly (varint / 64-bit / length-delimited
/ 32-bit fields; field numbers from the published onnx.proto3), keeping only what weight ingestion needs -
nodes with attributes, initialisers, graph inputs/outputs with shapes, ``metadata_props``.
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, Iterator, List, Optional, Tuple

import numpy as np


# ---------------------------------------------------------------- protobuf wire format
def _varint(buf: bytes, pos: int) -> Tuple[int, int]:
    out = 0
    shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if b < 0x80:
            return out, pos
        shift += 7
        if shift > 70:
            raise ValueError("malformed varint")


def _fields(buf) -> Iterator[Tuple[int, int, object]]:
    """Yield (field number, wire type, value); value is an int (varint / fixed) or a memoryview (bytes)."""
    buf = memoryview(buf)
    pos, end = 0, len(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        fno, wt = key >> 3, key & 7
        if wt == 0:
            v, pos = _varint(buf, pos)
        elif wt == 1:
            v = bytes(buf[pos:pos + 8]); pos += 8
        elif wt == 2:
            n, pos = _varint(buf, pos)
            if pos + n > end:
                raise ValueError("truncated protobuf message")
            v = buf[pos:pos + n]; pos += n
        elif wt == 5:
            v = bytes(buf[pos:pos + 4]); pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wt}")
        yield fno, wt, v


def _sint64(v: int) -> int:
    return v - (1 << 64) if v >= (1 << 63) else v


def _packed_varints(v) -> List[int]:
    out, pos, n = [], 0, len(v)
    while pos < n:
        x, pos = _varint(v, pos)
        out.append(_sint64(x))
    return out


# ---------------------------------------------------------------- ONNX messages
_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 4: np.uint16, 5: np.int16, 6: np.int32, 7: np.int64,
           9: np.bool_, 10: np.float16, 11: np.float64, 12: np.uint32, 13: np.uint64}


def _tensor(buf) -> Tuple[str, np.ndarray]:
    dims: List[int] = []
    dtype = 1
    name = ""
    raw = None
    floats: List[float] = []
    ints: List[int] = []
    doubles: List[float] = []
    for fno, wt, v in _fields(buf):
        if fno == 1:
            dims.extend(_packed_varints(v) if wt == 2 else [_sint64(v)])
        elif fno == 2:
            dtype = v
        elif fno == 8:
            name = bytes(v).decode()
        elif fno == 9:
            raw = bytes(v)
        elif fno == 4:
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else struct.unpack("<f", v))
        elif fno in (5, 7):
            ints.extend(_packed_varints(v) if wt == 2 else [_sint64(v)])
        elif fno == 10:
            doubles.extend(np.frombuffer(bytes(v), "<f8").tolist() if wt == 2 else struct.unpack("<d", v))
        elif fno == 14 and v == 1:
            raise NotImplementedError(f"tensor '{name}' uses external data, which the reference's exporter never writes")
    if dtype not in _DTYPES:
        raise NotImplementedError(f"tensor '{name}': ONNX data type {dtype}")
    dt = np.dtype(_DTYPES[dtype])
    if raw is not None:
        arr = np.frombuffer(raw, dt.newbyteorder("<")).astype(dt)
    elif dtype == 1:
        arr = np.asarray(floats, dt)
    elif dtype == 11:
        arr = np.asarray(doubles, dt)
    else:
        arr = np.asarray(ints, dt)
    n = int(np.prod(dims)) if dims else 1
    if arr.size != n:
        raise ValueError(f"tensor '{name}': {arr.size} elements for shape {dims}")
    return name, arr.reshape(dims)


@dataclass
class Node:
    op_type: str
    name: str
    inputs: List[str]
    outputs: List[str]
    attrs: Dict[str, object] = field(default_factory=dict)


def _attribute(buf) -> Tuple[str, object]:
    name = ""
    val: object = None
    ints: List[int] = []
    floats: List[float] = []
    have_ints = have_floats = False
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:
            val = struct.unpack("<f", v)[0]
        elif fno == 3:
            val = _sint64(v)
        elif fno == 4:
            val = bytes(v)
        elif fno == 5:
            val = _tensor(v)[1]
        elif fno == 7:
            have_floats = True
            floats.extend(np.frombuffer(bytes(v), "<f4").tolist() if wt == 2 else struct.unpack("<f", v))
        elif fno == 8:
            have_ints = True
            ints.extend(_packed_varints(v) if wt == 2 else [_sint64(v)])
    if have_ints:
        val = ints
    elif have_floats:
        val = floats
    return name, val


def _node(buf) -> Node:
    n = Node("", "", [], [])
    for fno, wt, v in _fields(buf):
        if fno == 1:
            n.inputs.append(bytes(v).decode())
        elif fno == 2:
            n.outputs.append(bytes(v).decode())
        elif fno == 3:
            n.name = bytes(v).decode()
        elif fno == 4:
            n.op_type = bytes(v).decode()
        elif fno == 5:
            k, a = _attribute(v)
            n.attrs[k] = a
    return n


def _value_info(buf) -> Tuple[str, Optional[List[object]]]:
    name = ""
    shape: Optional[List[object]] = None
    for fno, wt, v in _fields(buf):
        if fno == 1:
            name = bytes(v).decode()
        elif fno == 2:                                         # TypeProto
            for f2, _, v2 in _fields(v):
                if f2 != 1:                                    # tensor_type
                    continue
                for f3, _, v3 in _fields(v2):
                    if f3 != 2:                                # shape
                        continue
                    shape = []
                    for f4, _, v4 in _fields(v3):              # dim
                        if f4 != 1:
                            continue
                        d: object = None
                        for f5, _, v5 in _fields(v4):
                            if f5 == 1:
                                d = _sint64(v5)
                            elif f5 == 2:
                                d = bytes(v5).decode()
                        shape.append(d)
    return name, shape


@dataclass
class OnnxGraph:
    nodes: List[Node]
    initializers: Dict[str, np.ndarray]
    inputs: List[Tuple[str, Optional[List[object]]]]           # graph inputs that are not initialisers
    outputs: List[Tuple[str, Optional[List[object]]]]
    metadata: Dict[str, str]
    opset: int
    producer: str

    def producer_of(self, tensor: str) -> Optional[Node]:
        for n in self.nodes:
            if tensor in n.outputs:
                return n
        return None

    def constant(self, tensor: str) -> Optional[np.ndarray]:
        """Value of an initialiser or of a Constant node's output, else None."""
        if tensor in self.initializers:
            return self.initializers[tensor]
        n = self.producer_of(tensor)
        if n is not None and n.op_type == "Constant" and isinstance(n.attrs.get("value"), np.ndarray):
            return n.attrs["value"]
        return None


def read_onnx(path_or_bytes) -> OnnxGraph:
    if isinstance(path_or_bytes, (bytes, bytearray, memoryview)):
        data = bytes(path_or_bytes)
    else:
        with open(path_or_bytes, "rb") as f:
            data = f.read()
    graph_buf = None
    metadata: Dict[str, str] = {}
    opset = 0
    producer = ""
    for fno, wt, v in _fields(data):
        if fno == 7:
            graph_buf = v
        elif fno == 2:
            producer = bytes(v).decode()
        elif fno == 8:                                         # opset_import
            dom, ver = "", 0
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    dom = bytes(v2).decode()
                elif f2 == 2:
                    ver = v2
            if dom in ("", "ai.onnx"):
                opset = ver
        elif fno == 14:                                        # metadata_props
            k = val = ""
            for f2, _, v2 in _fields(v):
                if f2 == 1:
                    k = bytes(v2).decode()
                elif f2 == 2:
                    val = bytes(v2).decode()
            metadata[k] = val
    if graph_buf is None:
        raise ValueError("not an ONNX ModelProto: no graph")
    nodes: List[Node] = []
    inits: Dict[str, np.ndarray] = {}
    ins: List[Tuple[str, Optional[List[object]]]] = []
    outs: List[Tuple[str, Optional[List[object]]]] = []
    for fno, wt, v in _fields(graph_buf):
        if fno == 1:
            nodes.append(_node(v))
        elif fno == 5:
            k, a = _tensor(v)
            inits[k] = a
        elif fno == 11:
            ins.append(_value_info(v))
        elif fno == 12:
            outs.append(_value_info(v))
    ins = [(n, s) for n, s in ins if n not in inits]
    return OnnxGraph(nodes, inits, ins, outs, metadata, opset, producer)
