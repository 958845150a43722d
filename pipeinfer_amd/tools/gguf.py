"""Minimal GGUF v3 writer/reader (numpy only).

Host-side tooling: used to build the shape-exact synthetic model files that bench.py and the tests load through the
engine's own GGUF loader (pipeinfer_amd/csrc/engine/gguf.cpp).  File layout follows the format the reference reads
(ggml.c:18083-18330): magic, version, counts, KV pairs, tensor infos, aligned data section."""
import struct

import numpy as np

GGUF_MAGIC = 0x46554747
ALIGN = 32

# gguf value types
U8, I8, U16, I16, U32, I32, F32, BOOL, STR, ARR, U64, I64, F64 = range(13)
# ggml tensor types
T_F32, T_F16, T_Q4_0, T_Q4_1, T_Q5_0, T_Q5_1, T_Q8_0, T_Q2_K, T_Q3_K, T_Q4_K, T_Q5_K, T_Q6_K = 0, 1, 2, 3, 6, 7, 8, 10, 11, 12, 13, 14
BLCK = {T_F32: 1, T_F16: 1, T_Q4_0: 32, T_Q4_1: 32, T_Q5_0: 32, T_Q5_1: 32, T_Q8_0: 32, T_Q2_K: 256, T_Q3_K: 256, T_Q4_K: 256, T_Q5_K: 256, T_Q6_K: 256}
TSIZE = {T_F32: 4, T_F16: 2, T_Q4_0: 18, T_Q4_1: 20, T_Q5_0: 22, T_Q5_1: 24, T_Q8_0: 34, T_Q2_K: 84, T_Q3_K: 110, T_Q4_K: 144, T_Q5_K: 176, T_Q6_K: 210}
TYPE_BY_NAME = {"f32": T_F32, "f16": T_F16, "q4_0": T_Q4_0, "q4_1": T_Q4_1, "q5_0": T_Q5_0, "q5_1": T_Q5_1, "q8_0": T_Q8_0, "q2_k": T_Q2_K, "q3_k": T_Q3_K, "q4_k": T_Q4_K, "q5_k": T_Q5_K, "q6_k": T_Q6_K}

_SCALAR_FMT = {U8: "<B", I8: "<b", U16: "<H", I16: "<h", U32: "<I", I32: "<i", F32: "<f", BOOL: "<?", U64: "<Q", I64: "<q", F64: "<d"}


def nbytes(ttype, shape):
    n = int(np.prod(shape))
    return n // BLCK[ttype] * TSIZE[ttype]


def _s(b):
    b = b if isinstance(b, bytes) else b.encode("utf-8")
    return struct.pack("<Q", len(b)) + b


class GGUFWriter:
    def __init__(self, path):
        self.path = path
        self.kv = []        # (key, type, payload bytes)
        self.tensors = []   # (name, shape(ne order), ttype, data or callable)

    def add(self, key, vtype, value):
        if vtype == STR:
            payload = _s(value)
        elif vtype == ARR:
            etype, items = value
            if etype == STR:
                payload = struct.pack("<IQ", STR, len(items)) + b"".join(_s(x) for x in items)
            else:
                arr = np.asarray(items, dtype={F32: "<f4", I32: "<i4", U32: "<u4"}[etype])
                payload = struct.pack("<IQ", etype, arr.size) + arr.tobytes()
        else:
            payload = struct.pack(_SCALAR_FMT[vtype], value)
        self.kv.append((key, vtype, payload))

    def add_tensor(self, name, ne, ttype, data):
        """ne: ggml order (ne[0] = contiguous dim). data: bytes-like / ndarray, a callable returning one (lazy), or None:
        the tensor keeps its place in the file but its bytes are a hole (pipeline stages materialise only their own layers)."""
        self.tensors.append((name, tuple(int(x) for x in ne), ttype, data))

    def write(self):
        with open(self.path, "wb") as f:
            f.write(struct.pack("<IIQQ", GGUF_MAGIC, 3, len(self.tensors), len(self.kv)))
            for key, vtype, payload in self.kv:
                f.write(_s(key) + struct.pack("<I", vtype) + payload)
            off = 0
            for name, ne, ttype, _ in self.tensors:
                f.write(_s(name) + struct.pack("<I", len(ne)) + b"".join(struct.pack("<Q", d) for d in ne))
                f.write(struct.pack("<IQ", ttype, off))
                off += (nbytes(ttype, ne) + ALIGN - 1) // ALIGN * ALIGN
            pad = (-f.tell()) % ALIGN
            f.write(b"\0" * pad)
            for name, ne, ttype, data in self.tensors:
                if data is None:
                    f.seek((nbytes(ttype, ne) + ALIGN - 1) // ALIGN * ALIGN, 1)      # sparse: never read by this process
                    continue
                if callable(data):
                    data = data()
                buf = np.ascontiguousarray(data).view(np.uint8).reshape(-1)
                assert buf.size == nbytes(ttype, ne), (name, buf.size, nbytes(ttype, ne))
                buf.tofile(f)
                f.write(b"\0" * ((-buf.size) % ALIGN))
            f.truncate(f.tell())                 # a trailing hole still counts towards the file size


def read_gguf(path):
    """-> (kv dict, {name: (ne, ttype, np.memmap uint8)}) — used by tests to inspect fixtures"""
    mm = np.memmap(path, dtype=np.uint8, mode="r")
    pos = 0

    def rd(fmt):
        nonlocal pos
        v = struct.unpack_from(fmt, mm, pos)
        pos += struct.calcsize(fmt)
        return v[0] if len(v) == 1 else v

    def rstr():
        nonlocal pos
        n = rd("<Q")
        s = bytes(mm[pos:pos + n])
        pos += n
        return s

    def rval(t):
        nonlocal pos
        if t == STR:
            return rstr()
        if t == ARR:
            et, n = rd("<I"), rd("<Q")
            if et == STR:
                return [rstr() for _ in range(n)]
            dt = {U8: "u1", I8: "i1", U16: "<u2", I16: "<i2", U32: "<u4", I32: "<i4", F32: "<f4", BOOL: "u1", U64: "<u8", I64: "<i8", F64: "<f8"}[et]
            a = np.frombuffer(mm, dtype=dt, count=n, offset=pos).copy()
            pos += a.nbytes
            return a
        return rd(_SCALAR_FMT[t])

    magic, version, n_tensors, n_kv = rd("<I"), rd("<I"), rd("<Q"), rd("<Q")
    assert magic == GGUF_MAGIC and version in (2, 3)
    kv = {}
    for _ in range(n_kv):
        k = rstr().decode()
        kv[k] = rval(rd("<I"))
    infos = []
    for _ in range(n_tensors):
        name = rstr().decode()
        nd = rd("<I")
        ne = tuple(rd("<Q") for _ in range(nd))
        ttype, off = rd("<I"), rd("<Q")
        infos.append((name, ne, ttype, off))
    align = int(kv.get("general.alignment", ALIGN))
    base = (pos + align - 1) // align * align
    tensors = {name: (ne, ttype, mm[base + off: base + off + nbytes(ttype, ne)]) for name, ne, ttype, off in infos}
    return kv, tensors
