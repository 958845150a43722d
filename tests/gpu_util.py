"""helpers for the -m gpu tests: numpy <-> HBM through the C ABI (no torch involved)"""
import ctypes as C

import numpy as np

from pipeinfer_amd import lib as L


class Dev:
    """a device buffer with numpy round trips"""

    def __init__(self, arr=None, nbytes=None):
        self.lib = L.load()
        if arr is not None:
            arr = np.ascontiguousarray(arr)
            nbytes = arr.nbytes
        self.nbytes = int(nbytes)
        self.ptr = self.lib.pi_malloc(max(self.nbytes, 16))
        assert self.ptr, self.lib.pi_last_error()
        if arr is not None and self.nbytes:
            L.check(self.lib.pi_memcpy_h2d(self.ptr, arr.ctypes.data_as(C.c_void_p), self.nbytes, None), "h2d")
            L.check(self.lib.pi_stream_sync(None))
        elif self.nbytes:
            L.check(self.lib.pi_memset(self.ptr, 0, self.nbytes, None))

    def get(self, dtype, shape=None, offset=0, count=None):
        dtype = np.dtype(dtype)
        n = (self.nbytes - offset) // dtype.itemsize if count is None else count
        out = np.zeros(n, dtype)
        L.check(self.lib.pi_memcpy_d2h(out.ctypes.data_as(C.c_void_p), self.ptr + offset, out.nbytes, None), "d2h")
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            self.lib.pi_free(self.ptr)
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def init():
    lib = L.load()
    L.check(lib.pi_device_init(0), "pi_device_init")
    return lib


def q8_planes(lib, vdt, k, n_tok, dev):
    """read back the device activation planes (k-major, see pi_q8_layout) and undo the permutation
    -> (qs int8 [n_tok,k] in natural order, d, bsums)"""
    oq, od, ob = C.c_size_t(), C.c_size_t(), C.c_size_t()
    lib.pi_q8_layout(vdt, k, n_tok, C.byref(oq), C.byref(od), C.byref(ob))
    if vdt == 15:
        nb = k // 256
        raw = dev.get(np.int8, (n_tok, nb, 8, 4, 2, 4), oq.value, n_tok * k)            # [k][j][lo/hi][byte]
        qs = raw.transpose(0, 1, 3, 4, 2, 5).reshape(n_tok, k)                          # value e = 64j + 32h + 4k + byte
        d = dev.get(np.float32, (n_tok, nb), od.value, n_tok * nb)
        bs = dev.get(np.int16, (n_tok, k // 16), ob.value, n_tok * (k // 16))
        return qs, d, bs
    nb = k // 32
    ng = (nb + 3) // 4
    raw = dev.get(np.int8, (n_tok, ng, 8, 4, 4), oq.value, n_tok * ng * 128)            # [group][k][block][byte]
    qs = raw.transpose(0, 1, 3, 2, 4).reshape(n_tok, ng * 128)[:, :k]
    if vdt == 9:                                                                         # Q8_1: d and s as floats
        d = dev.get(np.float32, (n_tok, ng * 4), od.value, n_tok * ng * 4)[:, :nb]
        sm = dev.get(np.float32, (n_tok, ng * 4), ob.value, n_tok * ng * 4)[:, :nb]
        return qs, d, sm
    d = dev.get(np.uint16, (n_tok, ng * 4), od.value, n_tok * ng * 4)[:, :nb]
    return qs, d, None


def q8_as_ref_blocks(vdt, qs, d, bs):
    """re-assemble the reference block format (block_q8_K 292 B / block_q8_0 34 B) for byte comparison"""
    n_tok, k = qs.shape
    qs = np.ascontiguousarray(qs); d = np.ascontiguousarray(d)
    if vdt == 15:
        nb = k // 256
        out = np.zeros((n_tok, nb, 292), np.uint8)
        out[:, :, 0:4] = d.reshape(n_tok, nb, 1).view(np.uint8).reshape(n_tok, nb, 4)
        out[:, :, 4:260] = qs.reshape(n_tok, nb, 256).view(np.uint8)
        out[:, :, 260:292] = bs.reshape(n_tok, nb, 16).view(np.uint8).reshape(n_tok, nb, 32)
        return out.reshape(n_tok, -1)
    nb = k // 32
    if vdt == 9:                                                                         # block_q8_1: float d, float s, 32 quants
        out = np.zeros((n_tok, nb, 40), np.uint8)
        out[:, :, 0:4] = d.reshape(n_tok, nb, 1).view(np.uint8).reshape(n_tok, nb, 4)
        out[:, :, 4:8] = np.ascontiguousarray(bs).reshape(n_tok, nb, 1).view(np.uint8).reshape(n_tok, nb, 4)
        out[:, :, 8:40] = qs.reshape(n_tok, nb, 32).view(np.uint8)
        return out.reshape(n_tok, -1)
    out = np.zeros((n_tok, nb, 34), np.uint8)
    out[:, :, 0:2] = d.reshape(n_tok, nb, 1).view(np.uint8).reshape(n_tok, nb, 2)
    out[:, :, 2:34] = qs.reshape(n_tok, nb, 32).view(np.uint8)
    return out.reshape(n_tok, -1)
