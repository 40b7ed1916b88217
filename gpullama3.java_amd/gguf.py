"""Minimal GGUF v3 reader / writer for the tensors the forward pass consumes.

The Java host of the reference keeps its own loader (J/tensor/GGUF.java:43-92, 217-311); this
module exists so the harness in THIS repo can write synthetic random-weight models in the same
wire format and hand the raw tensor bytes to ``gl3_upload_tensor`` exactly as
the Java FFM shim would hand over slices of its mmap'd tensor-data segment
(J/tensor/GGUF.java:105-137).  Block layouts follow J/tensor/GGMLType.java:5-21:
F32 = 4 B, F16 = 2 B, Q4_0 = 18 B / 32 elems, Q8_0 = 34 B / 32 elems.
"""
from __future__ import annotations

import mmap
import struct

import numpy as np

GGUF_MAGIC = 0x46554747
ALIGNMENT = 32

GGML_F32, GGML_F16, GGML_Q4_0, GGML_Q8_0 = 0, 1, 2, 8
GGML_Q4_K, GGML_Q5_K, GGML_Q6_K = 12, 13, 14        # K-quants: 256-element super-blocks (J/tensor/GGMLType.java:18-20)
TYPE_SIZE = {GGML_F32: (1, 4), GGML_F16: (1, 2), GGML_Q4_0: (32, 18), GGML_Q8_0: (32, 34),
             GGML_Q4_K: (256, 144), GGML_Q5_K: (256, 176), GGML_Q6_K: (256, 210)}

# gguf_metadata_value_type
_U8, _I8, _U16, _I16, _U32, _I32, _F32, _BOOL, _STR, _ARR, _U64, _I64, _F64 = range(13)
_SCALAR = {_U8: "<B", _I8: "<b", _U16: "<H", _I16: "<h", _U32: "<I", _I32: "<i", _F32: "<f",
           _BOOL: "<?", _U64: "<Q", _I64: "<q", _F64: "<d"}


def byte_size(ggml_type: int, n_elems: int) -> int:
    bs, ts = TYPE_SIZE[ggml_type]
    assert n_elems % bs == 0
    return n_elems // bs * ts


def _w_str(f, s: str):
    b = s.encode("utf-8")
    f.write(struct.pack("<Q", len(b)))
    f.write(b)


def _w_value(f, v):
    if isinstance(v, bool):
        f.write(struct.pack("<I?", _BOOL, v))
    elif isinstance(v, int):
        f.write(struct.pack("<Ii", _I32, v)) if -2**31 <= v < 2**31 else f.write(struct.pack("<Iq", _I64, v))
    elif isinstance(v, float):
        f.write(struct.pack("<If", _F32, v))
    elif isinstance(v, str):
        f.write(struct.pack("<I", _STR))
        _w_str(f, v)
    elif isinstance(v, (list, tuple)):
        f.write(struct.pack("<I", _ARR))
        if len(v) and isinstance(v[0], str) or not len(v):
            f.write(struct.pack("<IQ", _STR, len(v)))
            for s in v:
                _w_str(f, s)
        elif isinstance(v[0], float):
            f.write(struct.pack("<IQ", _F32, len(v)))
            f.write(np.asarray(v, "<f4").tobytes())
        else:
            f.write(struct.pack("<IQ", _I32, len(v)))
            f.write(np.asarray(v, "<i4").tobytes())
    else:
        raise TypeError(type(v))


def write_gguf(path: str, metadata: dict, tensors: list):
    """tensors: list of (name, dims(list, GGUF order: fastest first), ggml_type, raw_bytes ndarray uint8)."""
    with open(path, "wb") as f:
        f.write(struct.pack("<IIQQ", GGUF_MAGIC, 3, len(tensors), len(metadata)))
        for k, v in metadata.items():
            _w_str(f, k)
            _w_value(f, v)
        off = 0
        offsets = []
        for name, dims, ty, raw in tensors:
            _w_str(f, name)
            f.write(struct.pack("<I", len(dims)))
            for d in dims:
                f.write(struct.pack("<Q", d))
            f.write(struct.pack("<IQ", ty, off))
            offsets.append(off)
            off += (len(raw) + ALIGNMENT - 1) // ALIGNMENT * ALIGNMENT
        pad = (ALIGNMENT - f.tell() % ALIGNMENT) % ALIGNMENT
        f.write(b"\0" * pad)
        for (_, _, _, raw) in tensors:
            b = raw.tobytes() if isinstance(raw, np.ndarray) else bytes(raw)
            f.write(b)
            f.write(b"\0" * ((ALIGNMENT - len(b) % ALIGNMENT) % ALIGNMENT))


class GGUFFile:
    """mmap-backed reader: ``metadata`` dict + ``tensors`` name -> (dims, ggml_type, uint8 view)."""

    def __init__(self, path: str):
        self._f = open(path, "rb")
        self._mm = mmap.mmap(self._f.fileno(), 0, access=mmap.ACCESS_READ)
        self._buf = np.frombuffer(self._mm, dtype=np.uint8)
        self._p = 0
        magic, version, n_t, n_kv = self._unpack("<IIQQ")
        if magic != GGUF_MAGIC:
            raise ValueError("unsupported header.magic %#x" % magic)
        if version not in (2, 3):
            raise ValueError("unsupported header.version %d" % version)
        self.metadata = {}
        for _ in range(n_kv):
            k = self._str()
            self.metadata[k] = self._value(self._unpack("<I")[0])
        infos = []
        for _ in range(n_t):
            name = self._str()
            nd = self._unpack("<I")[0]
            dims = [self._unpack("<Q")[0] for _ in range(nd)]
            ty, off = self._unpack("<IQ")
            infos.append((name, dims, ty, off))
        align = self.metadata.get("general.alignment", ALIGNMENT)
        self._p += (align - self._p % align) % align
        self.tensor_data_offset = self._p
        self.tensors = {}
        for name, dims, ty, off in infos:
            n = int(np.prod(dims))
            if ty not in TYPE_SIZE:
                raise ValueError("unsupported ggml type %d for %s" % (ty, name))
            sz = byte_size(ty, n)
            a = self.tensor_data_offset + off
            self.tensors[name] = (dims, ty, self._buf[a:a + sz])

    def _unpack(self, fmt):
        v = struct.unpack_from(fmt, self._mm, self._p)
        self._p += struct.calcsize(fmt)
        return v

    def _str(self):
        n = self._unpack("<Q")[0]
        s = bytes(self._mm[self._p:self._p + n]).decode("utf-8")
        self._p += n
        return s

    def _value(self, ty):
        if ty in _SCALAR:
            return self._unpack(_SCALAR[ty])[0]
        if ty == _STR:
            return self._str()
        if ty == _ARR:
            ety, n = self._unpack("<IQ")
            return [self._value(ety) for _ in range(n)]
        raise ValueError("bad metadata value type %d" % ty)
