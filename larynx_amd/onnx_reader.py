"""Dependency-free reader for the tensors of an ONNX file.

Released Larynx voices ship `generator.onnx` (`larynx/utils.py:203-209` accepts a voice
directory that has one; `larynx/glow_tts.py:98-100`, `larynx/hifi_gan.py:103-105` hand it to
onnxruntime).  The HIP backend needs the checkpoint TENSORS, not a runtime, and the `onnx`
package is not a dependency of Larynx — so this module decodes the protobuf wire format itself:
`ModelProto.graph` (field 7) -> `GraphProto.node` (1), `.initializer` (5), `.input` (11), `.output` (12);
`NodeProto.input` (1), `.output` (2), `.name` (3), `.op_type` (4), `.attribute` (5);
`AttributeProto.name` (1), `.f` (2), `.i` (3), `.t` (5), `.ints` (8), `.type` (20);
`TensorProto.dims` (1), `.data_type` (2), `.float_data` (4), `.int32_data` (5), `.int64_data` (7),
`.name` (8), `.raw_data` (9), `.double_data` (10) — the field numbers of onnx.proto3 (IR version 3+).
"""
from __future__ import annotations

import struct
import typing
from pathlib import Path

import numpy as np

_DTYPES = {1: np.float32, 2: np.uint8, 3: np.int8, 5: np.int16, 6: np.int32, 7: np.int64, 9: np.bool_, 10: np.float16, 11: np.float64}


def _varint(buf: bytes, pos: int) -> typing.Tuple[int, int]:
    out = shift = 0
    while True:
        b = buf[pos]
        pos += 1
        out |= (b & 0x7F) << shift
        if not b & 0x80:
            return out, pos
        shift += 7


def _fields(buf: bytes) -> typing.Iterator[typing.Tuple[int, int, typing.Any]]:
    """(field number, wire type, value) for every field of one message; length-delimited
    values come back as memoryview slices (no copies of multi-megabyte weight blobs)."""
    pos, end = 0, len(buf)
    mv = memoryview(buf)
    while pos < end:
        key, pos = _varint(buf, pos)
        field, wire = key >> 3, key & 7
        if wire == 0:
            val, pos = _varint(buf, pos)
        elif wire == 1:
            val = bytes(mv[pos : pos + 8])
            pos += 8
        elif wire == 2:
            n, pos = _varint(buf, pos)
            val = mv[pos : pos + n]
            pos += n
        elif wire == 5:
            val = bytes(mv[pos : pos + 4])
            pos += 4
        else:
            raise ValueError(f"unsupported protobuf wire type {wire}")
        yield field, wire, val


def _packed_varints(v) -> typing.List[int]:
    b = bytes(v)
    out, pos = [], 0
    while pos < len(b):
        x, pos = _varint(b, pos)
        out.append(x)
    return out


def _signed(x: int) -> int:
    return x - (1 << 64) if x >= 1 << 63 else x


class Unreadable(str):
    """Why an initializer could not be read (external data, element-count mismatch, unknown type).  Only tensors the
    weight loader actually NEEDS turn this into an error (`onnx_weights.state_dict_from_onnx`): a voice file may carry an
    unrelated tensor this reader cannot take."""


def _tensor(buf, strict: bool = True) -> typing.Tuple[str, typing.Union[np.ndarray, Unreadable]]:
    b = bytes(buf)
    dims: typing.List[int] = []
    dtype, name, raw = 1, "", None
    external = False
    floats: typing.List[float] = []
    ints: typing.List[int] = []
    doubles: typing.List[float] = []
    for f, w, v in _fields(b):
        if f == 1:
            dims += _packed_varints(v) if w == 2 else [v]
        elif f == 2:
            dtype = v
        elif f == 4:
            floats += list(struct.unpack(f"<{len(v) // 4}f", bytes(v))) if w == 2 else [struct.unpack("<f", v)[0]]
        elif f in (5, 7):
            ints += [_signed(x) for x in _packed_varints(v)] if w == 2 else [_signed(v)]
        elif f == 8:
            name = bytes(v).decode("utf-8")
        elif f == 9:
            raw = bytes(v)
        elif f == 10:
            doubles += list(struct.unpack(f"<{len(v) // 8}d", bytes(v))) if w == 2 else [struct.unpack("<d", v)[0]]
        elif f == 13:  # external_data entries
            external = True
        elif f == 14:  # data_location: 1 = EXTERNAL
            external = external or v == 1
    def bad(msg):
        if strict:
            raise ValueError(msg)
        return name, Unreadable(msg)

    if dtype not in _DTYPES:
        return bad(f"tensor '{name}': unsupported ONNX data type {dtype}")
    if external and raw is None:
        return bad(f"tensor '{name}': its data lives in an external file (data_location = EXTERNAL), which this reader "
                   "does not follow — re-export the model with the weights embedded")
    if dtype == 10 and raw is None and ints:
        # float16 values in int32_data are BIT PATTERNS (onnx.proto: "float16 values must be bit-wise converted to an uint16_t")
        raw = np.asarray(ints, np.uint16).tobytes()
    np_t = _DTYPES[dtype]
    if raw is not None:
        arr = np.frombuffer(raw, dtype=np.dtype(np_t).newbyteorder("<")).astype(np_t)
    elif floats:
        arr = np.asarray(floats, np_t)
    elif doubles:
        arr = np.asarray(doubles, np_t)
    else:
        arr = np.asarray(ints, np_t)
    shape = tuple(int(d) for d in dims)
    want = int(np.prod(shape, dtype=np.int64))
    if arr.size != want:
        return bad(f"tensor '{name}': {arr.size} elements stored for dims {list(shape)} ({want} expected)")
    return name, arr.reshape(shape)


class Node(typing.NamedTuple):
    op_type: str
    name: str
    inputs: typing.Tuple[str, ...]
    outputs: typing.Tuple[str, ...]
    attrs: typing.Dict[str, typing.Any]


def _attribute(buf) -> typing.Tuple[str, typing.Any]:
    name, val, ints = "", None, []
    for f, w, v in _fields(bytes(buf)):
        if f == 1:
            name = bytes(v).decode("utf-8")
        elif f == 2:
            val = struct.unpack("<f", v)[0]
        elif f == 3:
            val = _signed(v)
        elif f == 4:
            val = bytes(v)
        elif f == 5:
            val = _tensor(v)[1]
        elif f == 8:
            ints += [_signed(x) for x in _packed_varints(v)] if w == 2 else [_signed(v)]
    return name, (ints if ints else val)


def _node(buf) -> Node:
    ins, outs, name, op, attrs = [], [], "", "", {}
    for f, w, v in _fields(bytes(buf)):
        if f == 1:
            ins.append(bytes(v).decode("utf-8"))
        elif f == 2:
            outs.append(bytes(v).decode("utf-8"))
        elif f == 3:
            name = bytes(v).decode("utf-8")
        elif f == 4:
            op = bytes(v).decode("utf-8")
        elif f == 5:
            k, a = _attribute(v)
            attrs[k] = a
    return Node(op, name, tuple(ins), tuple(outs), attrs)


class OnnxGraph(typing.NamedTuple):
    nodes: typing.List[Node]                       # in graph (= execution) order
    initializers: typing.Dict[str, np.ndarray]     # name -> tensor, incl. the value of every Constant node
    inputs: typing.List[str]
    outputs: typing.List[str]
    unreadable: typing.Dict[str, str] = {}         # initializer name -> why it could not be read (see `Unreadable`)


def read_onnx(path: typing.Union[str, Path]) -> OnnxGraph:
    data = Path(path).read_bytes()
    graph = None
    for f, w, v in _fields(data):
        if f == 7 and w == 2:
            graph = bytes(v)
    if graph is None:
        raise ValueError(f"{path}: not an ONNX ModelProto (no graph)")
    nodes: typing.List[Node] = []
    inits: typing.Dict[str, np.ndarray] = {}
    bad: typing.Dict[str, str] = {}
    inputs: typing.List[str] = []
    outputs: typing.List[str] = []

    def value_name(buf) -> str:
        for f2, _, v2 in _fields(bytes(buf)):
            if f2 == 1:
                return bytes(v2).decode("utf-8")
        return ""

    for f, w, v in _fields(graph):
        if f == 1:
            nodes.append(_node(v))
        elif f == 5:
            name, arr = _tensor(v, strict=False)
            if isinstance(arr, Unreadable):
                bad[name] = str(arr)
            else:
                inits[name] = arr
        elif f == 11:
            inputs.append(value_name(v))
        elif f == 12:
            outputs.append(value_name(v))
    for n in nodes:  # Constant nodes carry tensors the exporter did not hoist into initializers
        if n.op_type == "Constant" and "value" in n.attrs and isinstance(n.attrs["value"], np.ndarray) and n.outputs:
            inits.setdefault(n.outputs[0], n.attrs["value"])
    return OnnxGraph(nodes, inits, [i for i in inputs if i not in inits], outputs, bad)
