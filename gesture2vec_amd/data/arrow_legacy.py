"""pyarrow's LEGACY object serialisation (`pyarrow.serialize(obj).to_buffer()` / `pyarrow.deserialize`, deprecated in 2.0
and later removed; the reference pins pyarrow==11.0.0 and stores every cached sample with it:
data_loader/data_preprocessor.py:318-333, read back at lmdb_data_loader.py:236,637) decoded with the Arrow IPC
primitives that current pyarrow still ships.

Layout of a serialised buffer (arrow/python/serialize.cc, SerializedPyObject::WriteTo):
    i32 num_tensors | i32 num_sparse_tensors | i32 num_ndarrays | i32 num_buffers
    Arrow IPC stream: schema + ONE record batch with one column "list" = dense union over the items of [obj]
    (64-byte aligned) the tensors, then the ndarrays, each as an IPC Tensor message; then the raw buffers (i64 size + bytes)
The union's children are created on first use and NAMED by the value's type tag (decimal string):
    0 None  1 bool  2 int64  3 py2 int  4 bytes  5 str  6 float16  7 float32  8 float64  9 date64  10 list  11 dict
    12 tuple  13 set  14 tensor  15 ndarray  16 buffer
list / tuple / set children are list<dense_union>, dict is list<struct<keys: dense_union, vals: dense_union>>, tensor /
ndarray / buffer children hold an int32 index into the arrays that follow the stream.

`serialize` writes the same layout for the value types the reference stores (None, bool, int, float, str, bytes, list,
tuple, dict, numpy arrays): it exists so that tests and synthetic caches can be produced without the removed API.
NOTE (parity): no buffer written by a real pyarrow<=11 is available in the build container (`pyarrow.serialize` is gone
from the installed 25.0); decoder and encoder agree with each other and with the published layout above."""
from __future__ import annotations

import struct
from typing import Any, List

import numpy as np
import pyarrow as pa

NONE, BOOL, INT, PY2INT, BYTES, STRING, HALF, FLOAT, DOUBLE, DATE64, LIST, DICT, TUPLE, SET, TENSOR, NDARRAY, BUFFER = range(17)


# ------------------------------------------------------------------------------------------------------------ decode
def _child_tags(union_type) -> List[int]:
    return [int(union_type.field(i).name) for i in range(union_type.num_fields)]


def _decode_union(arr, lo: int, hi: int, blobs) -> List[Any]:
    """items lo..hi-1 of a dense union array"""
    if hi <= lo:
        return []
    t = arr.type
    tags = _child_tags(t)
    code_to_pos = {c: i for i, c in enumerate(t.type_codes)}
    codes = arr.type_codes.to_numpy(zero_copy_only=False)
    offs = arr.offsets.to_numpy(zero_copy_only=False)
    kids = [arr.field(i) for i in range(t.num_fields)]
    out = []
    for j in range(lo, hi):
        pos = code_to_pos[int(codes[j])]
        out.append(_decode_child(kids[pos], tags[pos], int(offs[j]), blobs))
    return out


def _decode_child(child, tag: int, k: int, blobs):
    if not child[k].is_valid:
        return None
    if tag in (BOOL, INT, PY2INT, BYTES, STRING, HALF, FLOAT, DOUBLE, DATE64):
        v = child[k].as_py()
        return float(v) if tag in (HALF, FLOAT, DOUBLE) else v
    if tag in (LIST, TUPLE, SET):
        o = child.offsets.to_numpy(zero_copy_only=False)
        items = _decode_union(child.values, int(o[k]), int(o[k + 1]), blobs)
        return items if tag == LIST else (tuple(items) if tag == TUPLE else set(items))
    if tag == DICT:
        o = child.offsets.to_numpy(zero_copy_only=False)
        st = child.values
        keys = _decode_union(st.field("keys"), int(o[k]), int(o[k + 1]), blobs)
        vals = _decode_union(st.field("vals"), int(o[k]), int(o[k + 1]), blobs)
        return dict(zip(keys, vals))
    if tag == TENSOR:
        return blobs["tensors"][child[k].as_py()]
    if tag == NDARRAY:
        return blobs["ndarrays"][child[k].as_py()]
    if tag == BUFFER:
        return blobs["buffers"][child[k].as_py()]
    raise ValueError(f"legacy pyarrow serialisation: unsupported type tag {tag}")


def deserialize(buf) -> Any:
    """`pyarrow.deserialize(buf)` for buffers written by pyarrow <= 11 `serialize(...).to_buffer()`."""
    data = bytes(buf) if not isinstance(buf, (bytes, bytearray, memoryview)) else buf
    data = bytes(data)
    n_tensors, n_sparse, n_ndarrays, n_buffers = struct.unpack_from("<iiii", data, 0)
    if n_sparse != 0:
        raise ValueError("sparse tensors are not used by the reference's samples")
    src = pa.BufferReader(pa.py_buffer(data))
    src.seek(16)
    reader = pa.ipc.open_stream(src)
    batch = reader.read_next_batch()
    try:
        reader.read_next_batch()          # consume the end-of-stream marker
    except StopIteration:
        pass

    def read_arrays(n):
        out = []
        for _ in range(n):
            pos = src.tell()
            src.seek((pos + 63) // 64 * 64)
            out.append(pa.ipc.read_tensor(src).to_numpy())
        return out

    blobs = {"tensors": read_arrays(n_tensors), "ndarrays": read_arrays(n_ndarrays), "buffers": []}
    for _ in range(n_buffers):
        pos = src.tell()
        src.seek((pos + 63) // 64 * 64)
        (size,) = struct.unpack("<q", src.read(8))
        blobs["buffers"].append(src.read(size))
    col = batch.column(0)
    return _decode_union(col, 0, len(col), blobs)[0]


# ------------------------------------------------------------------------------------------------------------ encode
class _Seq:
    """a dense union under construction (SequenceBuilder)"""

    def __init__(self):
        self.codes: List[int] = []
        self.offsets: List[int] = []
        self.tags: List[int] = []          # child position -> tag
        self.store: List[Any] = []         # child position -> builder state

    def _child(self, tag: int, init):
        if tag not in self.tags:
            self.tags.append(tag)
            self.store.append(init())
        return self.tags.index(tag)

    def append(self, v, blobs):
        if v is None or isinstance(v, (bool, np.bool_)) or isinstance(v, (int, np.integer, float, np.floating, str, bytes)):
            tag = (BOOL if isinstance(v, (bool, np.bool_)) else INT if isinstance(v, (int, np.integer)) else
                   FLOAT if isinstance(v, np.float32) else HALF if isinstance(v, np.float16) else
                   DOUBLE if isinstance(v, (float, np.floating)) else STRING if isinstance(v, str) else
                   BYTES if isinstance(v, bytes) else None)
            if v is None:      # DenseUnionBuilder::AppendNull: a null in the first child
                if not self.tags:
                    self._child(INT, list)
                pos = 0
                self.store[0].append(None) if isinstance(self.store[0], list) else self._append_null_nested(0)
            else:
                pos = self._child(tag, list)
                self.store[pos].append(v.item() if isinstance(v, np.generic) else v)
            self.codes.append(pos)
            self.offsets.append(self._len(pos) - 1)
            return
        if isinstance(v, np.ndarray):
            pos = self._child(NDARRAY, list)
            blobs.append(np.ascontiguousarray(v))
            self.store[pos].append(len(blobs) - 1)
            self.codes.append(pos); self.offsets.append(len(self.store[pos]) - 1)
            return
        if isinstance(v, (list, tuple, set)):
            tag = LIST if isinstance(v, list) else TUPLE if isinstance(v, tuple) else SET
            pos = self._child(tag, lambda: {"offsets": [0], "values": _Seq()})
            st = self.store[pos]
            for item in v:
                st["values"].append(item, blobs)
            st["offsets"].append(len(st["values"].codes))
            self.codes.append(pos); self.offsets.append(len(st["offsets"]) - 2)
            return
        if isinstance(v, dict):
            pos = self._child(DICT, lambda: {"offsets": [0], "keys": _Seq(), "vals": _Seq()})
            st = self.store[pos]
            for k_, v_ in v.items():
                st["keys"].append(k_, blobs)
                st["vals"].append(v_, blobs)
            st["offsets"].append(len(st["keys"].codes))
            self.codes.append(pos); self.offsets.append(len(st["offsets"]) - 2)
            return
        raise TypeError(f"cannot serialise {type(v).__name__}")

    def _append_null_nested(self, pos):
        raise TypeError("None as the first item of a sequence whose first child is a container is not supported by this writer")

    def _len(self, pos):
        st = self.store[pos]
        return len(st) if isinstance(st, list) else len(st["offsets"]) - 1

    def finish(self):
        kids, names = [], []
        for tag, st in zip(self.tags, self.store):
            names.append(str(tag))
            if tag == BOOL:
                kids.append(pa.array(st, pa.bool_()))
            elif tag in (INT, PY2INT):
                kids.append(pa.array(st, pa.int64()))
            elif tag == BYTES:
                kids.append(pa.array(st, pa.binary()))
            elif tag == STRING:
                kids.append(pa.array(st, pa.string()))
            elif tag == HALF:
                kids.append(pa.array(np.asarray(st, np.float16)))
            elif tag == FLOAT:
                kids.append(pa.array(st, pa.float32()))
            elif tag == DOUBLE:
                kids.append(pa.array(st, pa.float64()))
            elif tag in (TENSOR, NDARRAY, BUFFER):
                kids.append(pa.array(st, pa.int32()))
            elif tag in (LIST, TUPLE, SET):
                kids.append(pa.ListArray.from_arrays(pa.array(st["offsets"], pa.int32()), st["values"].finish()))
            elif tag == DICT:
                pairs = pa.StructArray.from_arrays([st["keys"].finish(), st["vals"].finish()], names=["keys", "vals"])
                kids.append(pa.ListArray.from_arrays(pa.array(st["offsets"], pa.int32()), pairs))
        return pa.UnionArray.from_dense(pa.array(self.codes, pa.int8()), pa.array(self.offsets, pa.int32()), kids, names)


def serialize(obj) -> bytes:
    """Counterpart of `pyarrow.serialize(obj).to_buffer()` for None / bool / int / float / str / bytes / list / tuple / dict /
    numpy arrays (what data_preprocessor.py stores)."""
    blobs: List[np.ndarray] = []
    seq = _Seq()
    seq.append(obj, blobs)
    col = seq.finish()
    batch = pa.RecordBatch.from_arrays([col], names=["list"])
    sink = pa.BufferOutputStream()
    sink.write(struct.pack("<iiii", 0, 0, len(blobs), 0))
    with pa.ipc.new_stream(sink, batch.schema) as w:
        w.write_batch(batch)
    for a in blobs:
        pos = sink.tell()
        sink.write(b"\0" * ((pos + 63) // 64 * 64 - pos))
        pa.ipc.write_tensor(pa.Tensor.from_numpy(a), sink)
    return sink.getvalue().to_pybytes()
