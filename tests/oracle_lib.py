"""ctypes bindings for the TEST-ONLY oracle (oracle/liboracle.so) and, where it was built, the real reference
library behind oracle/_ref/libref_harness.so.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg import this module."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")

F32, F16, Q4_0, Q4_1, Q5_0, Q5_1, Q8_0, Q8_1, Q2_K, Q3_K, Q4_K, Q5_K, Q6_K, Q8_K = 0, 1, 2, 3, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15
TYPE_NAMES = {F32: "f32", F16: "f16", Q4_0: "q4_0", Q4_1: "q4_1", Q5_0: "q5_0", Q5_1: "q5_1", Q8_0: "q8_0", Q8_1: "q8_1", Q2_K: "q2_K", Q3_K: "q3_K", Q4_K: "q4_K", Q5_K: "q5_K", Q6_K: "q6_K", Q8_K: "q8_K"}
BLCK = {F32: 1, F16: 1, Q4_0: 32, Q4_1: 32, Q5_0: 32, Q5_1: 32, Q8_0: 32, Q8_1: 32, Q2_K: 256, Q3_K: 256, Q4_K: 256, Q5_K: 256, Q6_K: 256, Q8_K: 256}
TSIZE = {F32: 4, F16: 2, Q4_0: 18, Q4_1: 20, Q5_0: 22, Q5_1: 24, Q8_0: 34, Q8_1: 40, Q2_K: 84, Q3_K: 110, Q4_K: 144, Q5_K: 176, Q6_K: 210, Q8_K: 292}
VEC_DOT_TYPE = {Q4_0: Q8_0, Q4_1: Q8_1, Q5_0: Q8_0, Q5_1: Q8_1, Q8_0: Q8_0, Q2_K: Q8_K, Q3_K: Q8_K, Q4_K: Q8_K, Q5_K: Q8_K, Q6_K: Q8_K, F16: F16, F32: F32}


def row_size(t, n):
    return n // BLCK[t] * TSIZE[t]


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)


def build_oracle():
    so = os.path.join(ORACLE_DIR, "liboracle.so")
    srcs = [os.path.join(ORACLE_DIR, f) for f in ("quants.c", "ops.c", "kvcache.c", "model.c", "oracle.h", "Makefile")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["make", "-C", ORACLE_DIR, "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


_oracle = None
_ref = None


def oracle():
    global _oracle
    if _oracle is None:
        lib = C.CDLL(build_oracle())
        lib.o_fp16_to_fp32.restype = C.c_float
        lib.o_fp16_to_fp32.argtypes = [C.c_uint16]
        lib.o_fp32_to_fp16.restype = C.c_uint16
        lib.o_fp32_to_fp16.argtypes = [C.c_float]
        lib.o_vec_dot.restype = C.c_float
        lib.o_vec_dot.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.o_vec_dot_f16.restype = C.c_float
        lib.o_vec_dot_f16.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
        for name in ("o_table_exp_f16", "o_table_silu_f16", "o_table_gelu_f16", "o_kv_new", "o_model_load", "o_model_kv"):
            getattr(lib, name).restype = C.c_void_p
        lib.o_kv_new.argtypes = [C.c_int, C.c_int]
        lib.o_model_load.argtypes = [C.c_char_p, C.c_int, C.c_int]
        lib.o_model_kv.argtypes = [C.c_void_p]
        lib.o_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        lib.o_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        lib.o_rope.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float]
        lib.o_soft_max.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.o_silu.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.o_gelu.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.o_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.o_get_rows.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.o_quantize_act.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.o_dequantize_row.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.o_block_isums.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.o_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
        lib.o_attention_alibi.argtypes = lib.o_attention.argtypes[:-1] + [C.c_float, C.c_void_p]
        lib.o_rope_ex.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float]
        for name, nargs in (("o_kv_free", 1), ("o_kv_clear", 1), ("o_kv_head", 1), ("o_kv_used", 1), ("o_kv_cell_max", 1), ("o_kv_n", 1)):
            getattr(lib, name).argtypes = [C.c_void_p]
        lib.o_kv_pre_decode.argtypes = [C.c_void_p, C.c_int]
        lib.o_kv_post_decode.argtypes = [C.c_void_p, C.c_int]
        lib.o_kv_find_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        lib.o_kv_seq_rm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.o_kv_seq_cp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.o_kv_seq_keep.argtypes = [C.c_void_p, C.c_int]
        lib.o_kv_cell_pos.argtypes = [C.c_void_p, C.c_int]
        lib.o_kv_cell_has.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.o_kv_cell_nseq.argtypes = [C.c_void_p, C.c_int]
        lib.o_kv_build_mask.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.o_model_free.argtypes = [C.c_void_p]
        lib.o_model_info.argtypes = [C.c_void_p, C.c_int]
        lib.o_model_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        lib.o_model_forward.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.o_argmax.argtypes = [C.c_void_p, C.c_int]
        lib.o_softmax_topk.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        lib.o_set_threads.argtypes = [C.c_int]
        lib.o_tables_init()
        _oracle = lib
    return _oracle


def have_ref():
    return os.path.exists(os.path.join(ORACLE_DIR, "_ref", "libref_harness.so"))


_ref_kv = None


def ref_kv():
    """The REAL reference KV-cell functions: the reference's llama.cpp compiled with oracle/ref_kv_harness.cpp (oracle/_ref/libref_kv.so); None where it
    was never built.  The object keeps the MPI layer's functions undefined (nothing here reaches them), so it is opened with lazy binding — ctypes itself
    always asks for RTLD_NOW."""
    global _ref_kv
    if _ref_kv is None:
        path = os.path.join(ORACLE_DIR, "_ref", "libref_kv.so")
        if not os.path.exists(path):
            if os.path.isdir("/root/reference"):
                subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
            else:
                return None
        ref()                                                                # libggml_ref.so first, from the same directory
        libc = C.CDLL(None)
        libc.dlopen.restype, libc.dlopen.argtypes = C.c_void_p, [C.c_char_p, C.c_int]
        h = libc.dlopen(path.encode(), os.RTLD_LAZY | os.RTLD_LOCAL)
        assert h, path
        lib = C.CDLL(path, handle=h)
        lib.refkv_new.restype, lib.refkv_new.argtypes = C.c_void_p, [C.c_int]
        lib.refkv_free.argtypes = [C.c_void_p]
        lib.refkv_find_slot.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int]
        for f in ("refkv_cell_max", "refkv_clear", "refkv_head", "refkv_used"):
            getattr(lib, f).argtypes = [C.c_void_p]
        lib.refkv_set_head.argtypes = [C.c_void_p, C.c_int]
        lib.refkv_seq_rm.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.refkv_seq_cp.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int]
        lib.refkv_seq_keep.argtypes = [C.c_void_p, C.c_int]
        lib.refkv_cell_pos.argtypes = lib.refkv_cell_nseq.argtypes = [C.c_void_p, C.c_int]
        lib.refkv_cell_has.argtypes = [C.c_void_p, C.c_int, C.c_int]
        lib.reftok_load.restype, lib.reftok_load.argtypes = C.c_void_p, [C.c_char_p]
        lib.reftok_free.argtypes = lib.reftok_n_vocab.argtypes = [C.c_void_p]
        lib.reftok_tokenize.argtypes = [C.c_void_p, C.c_char_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int]
        lib.reftok_piece.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int]
        lib.refmodel_load.restype, lib.refmodel_load.argtypes = C.c_void_p, [C.c_char_p, C.c_int, C.c_int]
        lib.refmodel_kv.restype, lib.refmodel_kv.argtypes = C.c_void_p, [C.c_void_p]
        lib.refmodel_free.argtypes = lib.refmodel_n_vocab.argtypes = [C.c_void_p]
        lib.refmodel_decode.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        _ref_kv = lib
    return _ref_kv


def ref():
    """The REAL reference kernels (ggml) behind our harness; None where oracle/_ref was never built."""
    global _ref
    if _ref is None:
        if not have_ref():
            if os.path.isdir("/root/reference"):
                subprocess.check_call(["make", "-C", ORACLE_DIR, "ref"], stdout=subprocess.DEVNULL)
            else:
                return None
        lib = C.CDLL(os.path.join(ORACLE_DIR, "_ref", "libref_harness.so"))
        lib.ref_vec_dot.restype = C.c_float
        lib.ref_vec_dot.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        lib.ref_quantize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_quantize_reference.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_dequantize.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_fp32_to_fp16_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_fp16_to_fp32_row.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_mul_mat.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
        lib.ref_rms_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        lib.ref_norm.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float]
        lib.ref_rope.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float]
        lib.ref_soft_max.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int]
        lib.ref_unary.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_get_rows.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
        lib.ref_attention.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int]
        lib.ref_model_load.restype = C.c_void_p
        lib.ref_model_load.argtypes = [C.c_char_p, C.c_int]
        lib.ref_model_free.argtypes = [C.c_void_p]
        lib.ref_model_info.argtypes = [C.c_void_p, C.c_int]
        lib.ref_model_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int]
        lib.ref_time_us.restype = C.c_int64
        _ref = lib
    return _ref


# ---- numpy-friendly wrappers (same signatures for oracle and ref) -------------------------------------------
class Impl:
    def __init__(self, which):
        self.which = which
        self.lib = oracle() if which == "oracle" else ref()

    def quantize_act(self, vdt, x):
        x = np.ascontiguousarray(x, np.float32)
        out = np.zeros(row_size(vdt, x.size), np.uint8)
        if self.which == "oracle":
            self.lib.o_quantize_act(vdt, _ptr(x), _ptr(out), x.size)
        else:
            self.lib.ref_quantize(vdt, _ptr(x), _ptr(out), x.size)
        return out

    def dequantize(self, t, blocks, k):
        out = np.zeros(k, np.float32)
        blocks = np.ascontiguousarray(blocks)
        if self.which == "oracle":
            self.lib.o_dequantize_row(t, _ptr(blocks), _ptr(out), k)
        else:
            self.lib.ref_dequantize(t, _ptr(blocks), _ptr(out), k)
        return out

    def vec_dot(self, t, n, x, y):
        x = np.ascontiguousarray(x)
        y = np.ascontiguousarray(y)
        if self.which == "oracle":
            if t == F16:
                return self.lib.o_vec_dot_f16(n, _ptr(x), _ptr(y))
            return self.lib.o_vec_dot(t, n, _ptr(x), _ptr(y))
        return self.lib.ref_vec_dot(t, n, _ptr(x), _ptr(y))

    def mul_mat(self, t, w, ne00, ne01, act, n_threads=4):
        act = np.ascontiguousarray(act, np.float32)
        n_tok = act.size // ne00
        out = np.zeros((n_tok, ne01), np.float32)
        w = np.ascontiguousarray(w)
        if self.which == "oracle":
            self.lib.o_mul_mat(t, _ptr(w), ne00, ne01, _ptr(act), n_tok, _ptr(out))
        else:
            self.lib.ref_mul_mat(t, _ptr(w), ne00, ne01, _ptr(act), n_tok, _ptr(out), n_threads)
        return out

    def rms_norm(self, x, eps):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        fn = self.lib.o_rms_norm if self.which == "oracle" else self.lib.ref_rms_norm
        fn(_ptr(x), _ptr(y), x.shape[-1], x.size // x.shape[-1], eps)
        return y

    def norm(self, x, eps):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        fn = self.lib.o_norm if self.which == "oracle" else self.lib.ref_norm
        fn(_ptr(x), _ptr(y), x.shape[-1], x.size // x.shape[-1], eps)
        return y

    def rope(self, x, pos, mode, freq_base=10000.0, freq_scale=1.0):
        """x: [n_tok, n_head, head_dim]"""
        x = np.ascontiguousarray(x, np.float32)
        pos = np.ascontiguousarray(pos, np.int32)
        y = np.zeros_like(x)
        n_tok, n_head, hd = x.shape
        if self.which == "oracle":
            self.lib.o_rope(_ptr(x), _ptr(y), hd, n_head, n_tok, _ptr(pos), mode, freq_base, freq_scale)
        else:
            self.lib.ref_rope(_ptr(x), _ptr(y), hd, n_head, n_tok, _ptr(pos), mode, 4096, freq_base, freq_scale)
        return y

    def rope_ex(self, x, pos, mode, n_dims, freq_base=10000.0, freq_scale=1.0):
        """x: [n_tok, n_head, head_dim]; rotates blocks of n_dims values (oracle only: the NeoX block walk of ggml.c:11133-11160)"""
        x = np.ascontiguousarray(x, np.float32)
        pos = np.ascontiguousarray(pos, np.int32)
        y = np.zeros_like(x)
        n_tok, n_head, hd = x.shape
        assert self.which == "oracle"
        self.lib.o_rope_ex(_ptr(x), _ptr(y), hd, n_dims, n_head, n_tok, _ptr(pos), mode, freq_base, freq_scale)
        return y

    def soft_max(self, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        fn = self.lib.o_soft_max if self.which == "oracle" else self.lib.ref_soft_max
        fn(_ptr(x), _ptr(y), x.shape[-1], x.size // x.shape[-1])
        return y

    def silu(self, x):
        return self._unary(0, x)

    def gelu(self, x):
        return self._unary(1, x)

    def _unary(self, which, x):
        x = np.ascontiguousarray(x, np.float32)
        y = np.zeros_like(x)
        if self.which == "oracle":
            (self.lib.o_silu if which == 0 else self.lib.o_gelu)(_ptr(x), _ptr(y), x.size)
        else:
            self.lib.ref_unary(which, _ptr(x), _ptr(y), x.size)
        return y

    def get_rows(self, t, w, ne00, ne01, ids):
        ids = np.ascontiguousarray(ids, np.int32)
        out = np.zeros((ids.size, ne00), np.float32)
        w = np.ascontiguousarray(w)
        if self.which == "oracle":
            self.lib.o_get_rows(t, _ptr(w), ne00, _ptr(ids), ids.size, _ptr(out))
        else:
            self.lib.ref_get_rows(t, _ptr(w), ne00, ne01, _ptr(ids), ids.size, _ptr(out))
        return out

    def attention(self, k_cache, v_cache, n_ctx, q, k, v, mask, hd, n_head, n_head_kv, kv_head, n_kv, n_threads=4, max_alibi_bias=0.0):
        """k_cache [n_ctx, n_embd_gqa] f16 (uint16), v_cache [n_embd_gqa, n_ctx]; updated in place. q [n_tok,n_head,hd]..."""
        q = np.ascontiguousarray(q, np.float32)
        k = np.ascontiguousarray(k, np.float32)
        v = np.ascontiguousarray(v, np.float32)
        mask = np.ascontiguousarray(mask, np.float32)
        n_tok = q.shape[0]
        out = np.zeros((n_tok, n_head * hd), np.float32)
        args = [_ptr(k_cache), _ptr(v_cache), n_ctx, _ptr(q), _ptr(k), _ptr(v), _ptr(mask), hd, n_head, n_head_kv, n_tok, kv_head, n_kv, _ptr(out)]
        if max_alibi_bias > 0.0:
            assert self.which == "oracle"
            self.lib.o_attention_alibi(*args[:-1], C.c_float(max_alibi_bias), args[-1])
        elif self.which == "oracle":
            self.lib.o_attention(*args)
        else:
            self.lib.ref_attention(*args, n_threads)
        return out


def f32_to_f16_bits(x):
    """RNE conversion through the oracle (numpy's astype(float16) is also RNE; kept for cross-checks)."""
    return np.ascontiguousarray(x, np.float32).astype(np.float16).view(np.uint16)


# ---- model-level backends for tests/scenarios.py ----------------------------------------------------------------
class OracleModel:
    """whole path through the restatement: cell placement + mask + forward (oracle/model.c o_model_decode)"""

    def __init__(self, path, n_ctx=128, max_seq=8):
        self.lib = oracle()
        self.h = self.lib.o_model_load(path.encode(), n_ctx, max_seq)
        assert self.h, path
        self.kv = self.lib.o_model_kv(self.h)
        self.n_vocab = self.lib.o_model_info(self.h, 0)
        self.stride = max_seq

    def decode(self, tokens, pos, seq_ids):
        from tests.scenarios import flatten_seq_ids
        n = len(tokens)
        cnt, flat = flatten_seq_ids(seq_ids, self.stride)
        t = np.asarray(tokens, np.int32); p = np.asarray(pos, np.int32)
        out = np.zeros((n, self.n_vocab), np.float32)
        rc = self.lib.o_model_decode(self.h, n, _ptr(t), _ptr(p), _ptr(cnt), _ptr(flat), self.stride, _ptr(out))
        assert rc == 0
        return out

    def seq_rm(self, s, p0, p1):
        self.lib.o_kv_seq_rm(self.kv, s, p0, p1)

    def seq_cp(self, a, b, p0, p1):
        self.lib.o_kv_seq_cp(self.kv, a, b, p0, p1)

    def seq_keep(self, s):
        self.lib.o_kv_seq_keep(self.kv, s)

    def cells(self):
        n = self.lib.o_model_info(self.h, 7)
        return [(self.lib.o_kv_cell_pos(self.kv, i), tuple(s for s in range(self.stride) if self.lib.o_kv_cell_has(self.kv, i, s))) for i in range(n)]

    def close(self):
        self.lib.o_model_free(self.h)


class RefLlama:
    """the reference's OWN model code: llama.cpp's loader, KV cache, graph builders (build_llama / build_falcon) and ggml kernels, compiled from the reference
    sources into oracle/_ref/libref_kv.so (oracle/ref_kv_harness.cpp: only the MPI-calling shells of llama_new_context_with_model / llama_decode_internal are
    replaced by their own few lines).  Same interface as OracleModel."""

    def __init__(self, path, n_ctx=128, max_seq=8, n_threads=2):
        self.lib = ref_kv()
        self.h = self.lib.refmodel_load(path.encode(), n_ctx, n_threads)
        assert self.h, path
        self.kv = self.lib.refmodel_kv(self.h)
        self.n_vocab = self.lib.refmodel_n_vocab(self.h)
        self.stride = max_seq

    def decode(self, tokens, pos, seq_ids):
        from tests.scenarios import flatten_seq_ids
        n = len(tokens)
        cnt, flat = flatten_seq_ids(seq_ids, self.stride)
        t = np.asarray(tokens, np.int32); p = np.asarray(pos, np.int32)
        out = np.zeros((n, self.n_vocab), np.float32)
        assert self.lib.refmodel_decode(self.h, n, _ptr(t), _ptr(p), _ptr(cnt), _ptr(flat), self.stride, _ptr(out)) >= 0
        return out

    def seq_rm(self, s, p0, p1):
        self.lib.refkv_seq_rm(self.kv, s, p0, p1)

    def seq_cp(self, a, b, p0, p1):
        if a != b:                                   # the public wrapper returns early (llama.cpp:9303-9305)
            self.lib.refkv_seq_cp(self.kv, a, b, p0, p1)

    def seq_keep(self, s):
        self.lib.refkv_seq_keep(self.kv, s)

    def close(self):
        self.lib.refmodel_free(self.h)


class RefModel:
    """the REAL reference kernels for the arithmetic (ggml graph through oracle/ref_harness.c); cell placement and mask
    from oracle/kvcache.c because llama.cpp itself cannot be built here."""

    def __init__(self, path, n_ctx=128, max_seq=8, n_threads=4):
        self.o = oracle()
        self.r = ref()
        self.h = self.r.ref_model_load(path.encode(), n_ctx)
        assert self.h, path
        self.kv = self.o.o_kv_new(n_ctx, max_seq)
        self.n_vocab = self.r.ref_model_info(self.h, 0)
        self.stride = max_seq
        self.n_threads = n_threads
        self.n_ctx = n_ctx

    def decode(self, tokens, pos, seq_ids):
        from tests.scenarios import flatten_seq_ids
        n = len(tokens)
        cnt, flat = flatten_seq_ids(seq_ids, self.stride)
        t = np.asarray(tokens, np.int32); p = np.asarray(pos, np.int32)
        self.o.o_kv_pre_decode(self.kv, n)
        head = self.o.o_kv_find_slot(self.kv, n, _ptr(p), _ptr(cnt), _ptr(flat), self.stride)
        assert head >= 0
        n_kv = self.o.o_kv_n(self.kv)
        mask = np.zeros((n, n_kv), np.float32)
        self.o.o_kv_build_mask(self.kv, n_kv, n, _ptr(p), _ptr(cnt), _ptr(flat), self.stride, _ptr(mask))
        out = np.zeros((n, self.n_vocab), np.float32)
        self.r.ref_model_decode(self.h, _ptr(t), _ptr(p), n, head, n_kv, _ptr(mask), _ptr(out), self.n_threads)
        self.o.o_kv_post_decode(self.kv, n)
        return out

    def seq_rm(self, s, p0, p1):
        self.o.o_kv_seq_rm(self.kv, s, p0, p1)

    def seq_cp(self, a, b, p0, p1):
        self.o.o_kv_seq_cp(self.kv, a, b, p0, p1)

    def seq_keep(self, s):
        self.o.o_kv_seq_keep(self.kv, s)

    def close(self):
        self.r.ref_model_free(self.h)
        self.o.o_kv_free(self.kv)
