"""ctypes loader for libpipeinfer_amd.so (HIP kernels + engine, C ABI in include/*.h).

There is no fallback: if the shared library is missing or no MI355X is visible, the compute entry points raise."""
import ctypes as C
import os
import re
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
SO = os.environ.get("PIPEINFER_LIB") or os.path.join(PKG, "libpipeinfer_amd.so")        # (override: A/B runs of two builds on one box, tools_dev/)

_lib = None


class PipeInferError(RuntimeError):
    pass


def build(force=False):
    """compile the HIP extension in-tree (hipcc cross-compiles gfx950 without a GPU)"""
    if force or not os.path.exists(SO) or _stale():
        subprocess.check_call(["make", "-C", os.path.join(PKG, "csrc"), "-j8"], stdout=subprocess.DEVNULL)
    return SO


def _stale():
    t = os.path.getmtime(SO)
    for d in (os.path.join(PKG, "csrc", "kernels"), os.path.join(PKG, "csrc", "engine"), os.path.join(ROOT, "include")):
        if os.path.isdir(d):
            for f in os.listdir(d):
                if os.path.getmtime(os.path.join(d, f)) > t:
                    return os.path.exists("/opt/rocm/bin/hipcc")
    return False


def declared_symbols():
    """every PI_API / LLAMA_API / GGML_API function name declared in include/*.h"""
    names = []
    inc = os.path.join(ROOT, "include")
    for f in sorted(os.listdir(inc)):
        if f.endswith(".h"):
            text = open(os.path.join(inc, f)).read()
            text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
            text = re.sub(r"//[^\n]*", "", text)
            text = re.sub(r"#define[^\n]*", "", text)
            text = text.replace("DEPRECATED(", "")
            names += re.findall(r"(?:PI_API|LLAMA_API|GGML_API)\s+[^;{]*?\b(\w+)\s*\(", text)
    return sorted(set(names))


def load():
    global _lib
    if _lib is None:
        if not os.path.exists(SO):
            if os.path.exists("/opt/rocm/bin/hipcc"):
                build()
            else:
                raise PipeInferError(f"{SO} is missing and hipcc is not available: the MI355X path has no fallback")
        lib = C.CDLL(SO)
        _declare(lib)
        _lib = lib
    return _lib


def check(rc, what=""):
    if rc != 0:
        raise PipeInferError(f"{what} failed: {load().pi_last_error().decode()}")


def _declare(lib):
    vp, i32, i64, f32, sz = C.c_void_p, C.c_int, C.c_int64, C.c_float, C.c_size_t
    lib.pi_last_error.restype = C.c_char_p
    lib.pi_device_count.restype = i32
    lib.pi_device_init.argtypes = [i32]
    lib.pi_malloc.restype = vp
    lib.pi_malloc.argtypes = [sz]
    lib.pi_free.argtypes = [vp]
    lib.pi_memcpy_h2d.argtypes = [vp, vp, sz, vp]
    lib.pi_memcpy_d2h.argtypes = [vp, vp, sz, vp]
    lib.pi_memset.argtypes = [vp, i32, sz, vp]
    lib.pi_stream_sync.argtypes = [vp]
    lib.pi_set_abort_word.argtypes = [vp]; lib.pi_set_abort_word.restype = None
    lib.pi_op_abort_begin.argtypes = [vp, i32, vp, vp]
    lib.pi_op_abort_cancel.argtypes = [vp, i32, vp]
    lib.pi_op_copy_out.argtypes = [vp, vp, C.c_int64, vp, vp, vp, vp, vp, vp, i32, vp]
    lib.pi_kernel_timing.argtypes = [i32, i32]
    lib.pi_kernel_timing_rows.argtypes = [vp, i32]
    lib.pi_kernel_timing_rows.restype = i32
    lib.pi_kernel_lifetime_rows.argtypes = [vp, i32]
    lib.pi_kernel_lifetime_rows.restype = i32
    lib.pi_weight_upload.restype = vp
    lib.pi_weight_upload.argtypes = [i32, i64, i64, vp, vp]
    lib.pi_weight_free.argtypes = [vp]
    lib.pi_weight_nbytes.restype = sz
    lib.pi_weight_nbytes.argtypes = [vp]
    lib.pi_weight_type.argtypes = [vp]
    lib.pi_vec_dot_type.argtypes = [i32]
    lib.pi_q8_nbytes.restype = sz
    lib.pi_q8_nbytes.argtypes = [i32, i64, i32]
    lib.pi_q8_layout.argtypes = [i32, i64, i32, C.POINTER(sz), C.POINTER(sz), C.POINTER(sz)]
    lib.pi_op_get_rows.argtypes = [vp, vp, i32, vp, vp]
    lib.pi_op_norm.argtypes = [vp, vp, vp, vp, i32, i32, f32, i32, vp]
    lib.pi_op_quantize.argtypes = [i32, vp, i64, i32, vp, vp]
    lib.pi_op_norm_quantize.argtypes = [vp, vp, vp, vp, i32, i32, f32, i32, i32, vp, vp]
    lib.pi_op_act_quantize.argtypes = [vp, vp, i32, i64, i32, i32, vp, vp]
    lib.pi_op_mul_mat.argtypes = [vp, vp, i32, vp, vp, vp]
    lib.pi_op_mul_mat_multi.argtypes = [i32, vp, vp, i32, vp, vp, vp]
    lib.pi_op_mul_mat_qkv_rope_ok.argtypes = [vp, vp, vp, i32, i32]
    lib.pi_op_mul_mat_qkv_rope.argtypes = [vp, vp, vp, vp, i32, vp, vp, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.pi_rope_table_create.restype = vp
    lib.pi_rope_table_create.argtypes = [i32, i32, f32, f32, i32]
    lib.pi_rope_table_create_ex.restype = vp
    lib.pi_rope_table_create_ex.argtypes = [i32, i32, i32, f32, f32, i32]
    lib.pi_rope_table_free.argtypes = [vp]
    lib.pi_alibi_slopes.argtypes = [i32, f32, vp]; lib.pi_alibi_slopes.restype = None
    lib.pi_op_attention_alibi.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, i32, vp, vp]
    lib.pi_op_persimmon_qkv_store.argtypes = [vp, i32, i32, vp, vp, vp, vp, f32, vp, vp, i32, i32, vp, vp, vp, vp, vp]
    lib.pi_op_add_bias.argtypes = [vp, vp, i64, i32, vp]
    lib.pi_op_clamp.argtypes = [vp, f32, f32, i64, vp]
    lib.pi_op_rope_kv_store.argtypes = [vp, vp, vp, i64, i64, i64, vp, vp, i32, i32, i32, i32, i32, i32, vp, vp, vp, vp, vp]
    lib.pi_op_attention.argtypes = [vp, vp, vp, vp, i32, i32, i32, i32, i32, vp, vp, vp]
    lib.pi_op_attention_reserve.argtypes = [i32, i32]
    lib.pi_op_silu_mul.argtypes = [vp, vp, vp, i64, vp]
    lib.pi_op_gelu.argtypes = [vp, vp, i64, vp]
    lib.pi_op_add.argtypes = [vp, vp, vp, i64, vp]
    lib.pi_op_topk_scratch_bytes.argtypes = [i32]; lib.pi_op_topk_scratch_bytes.restype = C.c_size_t
    lib.pi_op_topk.argtypes = [vp, i32, i32, i32, vp, vp, vp, vp, vp]
