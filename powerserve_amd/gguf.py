"""Minimal GGUF v3 writer/reader (numpy only).

File layout follows what the reference reads with gguf_init_from_file (libs/ggml/src/ggml.c:23249) and
writes with gguf_write_to_file (ggml.c:24220): header, KV section, tensor infos, padding to
`general.alignment` (default 32), tensor data each aligned to the same value.  Tensor names are the ones the
reference looks up (src/model/common/weights.hpp:26-69).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass

import numpy as np

GGUF_MAGIC = b"GGUF"
GGUF_VERSION = 3
ALIGNMENT = 32

# ggml_type values (libs/ggml/include/ggml.h:361-398)
F32, F16, Q4_0, Q8_0, Q4_K, Q5_K, Q6_K, Q8_K, I32 = 0, 1, 2, 8, 12, 13, 14, 15, 26
TYPE_NAME = {F32: "F32", F16: "F16", Q4_0: "Q4_0", Q8_0: "Q8_0", Q4_K: "Q4_K", Q5_K: "Q5_K", Q6_K: "Q6_K"}
NAME_TYPE = {v: k for k, v in TYPE_NAME.items()}
BLOCK = {F32: (1, 4), F16: (1, 2), Q4_0: (32, 18), Q8_0: (32, 34), Q4_K: (256, 144), Q5_K: (256, 176), Q6_K: (256, 210), I32: (1, 4)}

# gguf_type enum
_U8, _I8, _U16, _I16, _U32, _I32, _F32, _BOOL, _STR, _ARR, _U64, _I64, _F64 = range(13)


def row_size(t: int, k: int) -> int:
    blk, ts = BLOCK[t]
    assert k % blk == 0, f"K={k} not a multiple of block {blk} for {TYPE_NAME[t]}"
    return k // blk * ts


def tensor_nbytes(t: int, ne) -> int:
    n = row_size(t, ne[0])
    for d in ne[1:]:
        n *= d
    return n


def _pad(n: int, a: int = ALIGNMENT) -> int:
    return (n + a - 1) // a * a


def _s(b: str) -> bytes:
    e = b.encode()
    return struct.pack("<Q", len(e)) + e


def _kv(key: str, val) -> bytes:
    out = _s(key)
    if isinstance(val, bool):
        out += struct.pack("<IB", _BOOL, int(val))
    elif isinstance(val, int):
        out += struct.pack("<II", _U32, val) if 0 <= val < 2**32 else struct.pack("<Iq", _I64, val)
    elif isinstance(val, float):
        out += struct.pack("<If", _F32, val)
    elif isinstance(val, str):
        out += struct.pack("<I", _STR) + _s(val)
    else:
        raise TypeError(type(val))
    return out


@dataclass
class TensorInfo:
    name: str
    type: int
    ne: tuple
    offset: int = 0  # relative to data section

    @property
    def nbytes(self) -> int:
        return tensor_nbytes(self.type, self.ne)


class GGUFWriter:
    """Two-phase writer: declare tensors, then stream their bytes in declaration order."""

    def __init__(self, path: str):
        self.path, self.kv, self.infos = path, [], []

    def add_kv(self, key: str, val) -> None:
        self.kv.append((key, val))

    def add_tensor(self, name: str, t: int, ne) -> None:
        self.infos.append(TensorInfo(name, t, tuple(int(x) for x in ne)))

    def write(self, producer) -> None:
        """producer(info) -> np.ndarray (uint8 or any dtype) holding exactly info.nbytes bytes."""
        off = 0
        for ti in self.infos:
            ti.offset = off
            off = _pad(off + ti.nbytes)
        hdr = GGUF_MAGIC + struct.pack("<IQQ", GGUF_VERSION, len(self.infos), len(self.kv))
        for k, v in self.kv:
            hdr += _kv(k, v)
        for ti in self.infos:
            hdr += _s(ti.name) + struct.pack("<I", len(ti.ne)) + struct.pack(f"<{len(ti.ne)}Q", *ti.ne)
            hdr += struct.pack("<IQ", ti.type, ti.offset)
        with open(self.path, "wb") as f:
            f.write(hdr)
            f.write(b"\0" * (_pad(len(hdr)) - len(hdr)))
            for ti in self.infos:
                a = np.ascontiguousarray(producer(ti))
                b = a.view(np.uint8).reshape(-1)
                assert b.size == ti.nbytes, (ti.name, b.size, ti.nbytes)
                f.write(memoryview(b))
                f.write(b"\0" * (_pad(ti.nbytes) - ti.nbytes))


class GGUFReader:
    def __init__(self, path: str):
        self.path = path
        self.mm = np.memmap(path, dtype=np.uint8, mode="r")
        buf = self.mm
        assert bytes(buf[:4]) == GGUF_MAGIC
        ver, nt, nkv = struct.unpack_from("<IQQ", buf, 4)
        assert ver in (2, 3)
        p = 24

        def rd_s():
            nonlocal p
            (n,) = struct.unpack_from("<Q", buf, p)
            s = bytes(buf[p + 8 : p + 8 + n]).decode()
            p += 8 + n
            return s

        fmt = {_U8: "<B", _I8: "<b", _U16: "<H", _I16: "<h", _U32: "<I", _I32: "<i", _F32: "<f", _BOOL: "<B",
               _U64: "<Q", _I64: "<q", _F64: "<d"}

        def rd_v(t):
            nonlocal p
            if t == _STR:
                return rd_s()
            if t == _ARR:
                et, n = struct.unpack_from("<IQ", buf, p)
                p += 12
                return [rd_v(et) for _ in range(n)]
            (v,) = struct.unpack_from(fmt[t], buf, p)
            p += struct.calcsize(fmt[t])
            return v

        self.kv = {}
        for _ in range(nkv):
            k = rd_s()
            (t,) = struct.unpack_from("<I", buf, p)
            p += 4
            self.kv[k] = rd_v(t)
        self.tensors = {}
        for _ in range(nt):
            name = rd_s()
            (nd,) = struct.unpack_from("<I", buf, p)
            p += 4
            ne = struct.unpack_from(f"<{nd}Q", buf, p)
            p += 8 * nd
            t, off = struct.unpack_from("<IQ", buf, p)
            p += 12
            self.tensors[name] = TensorInfo(name, t, tuple(ne), off)
        self.data_off = _pad(p, int(self.kv.get("general.alignment", ALIGNMENT)))

    def data(self, name: str) -> np.ndarray:
        ti = self.tensors[name]
        a = self.mm[self.data_off + ti.offset : self.data_off + ti.offset + ti.nbytes]
        return a.view(np.float32) if ti.type == F32 else a
