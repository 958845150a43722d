"""CPU: the offline quantisers (csrc/engine/quantize.cpp: pipeinfer_quantize_row, llama_model_quantize) against the reference's
quantize_row_*_reference (ggml-quants.c:428-640, 1458-2316) — committed golden bytes (tests/golden/quantize.npz, made by make_golden.py quantize
from the real reference functions) and, where oracle/_ref is built, the reference functions themselves on fresh random rows — and the file
driver against the mixture rules of llama.cpp:7896-8006 as restated independently in pipeinfer_amd/tools/synth.py."""
import ctypes as C
import os

import numpy as np
import pytest

from pipeinfer_amd import lib as L
from pipeinfer_amd.tools import gguf as G, synth
from tests import oracle_lib as O

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TYPES = [(2, "q4_0"), (3, "q4_1"), (6, "q5_0"), (7, "q5_1"), (8, "q8_0"), (10, "q2_K"), (11, "q3_K"), (12, "q4_K"), (13, "q5_K"), (14, "q6_K")]
FTYPES = {"q4_0": 2, "q4_1": 3, "q8_0": 7, "q5_0": 8, "q5_1": 9, "q2_k": 10, "q3_k_m": 12, "q4_k_s": 14, "q4_k_m": 15, "q5_k_s": 16, "q5_k_m": 17, "q6_k": 18}


class QuantizeParams(C.Structure):                                   # llama.h:205-212
    _fields_ = [("nthread", C.c_int), ("ftype", C.c_int), ("allow_requantize", C.c_bool), ("quantize_output_tensor", C.c_bool), ("only_copy", C.c_bool), ("pure", C.c_bool)]


@pytest.fixture(scope="module")
def lib():
    lib = L.load()
    lib.pipeinfer_quantize_row.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    lib.pipeinfer_quantize_row.restype = C.c_int64
    lib.llama_model_quantize.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(QuantizeParams)]
    lib.llama_model_quantize.restype = C.c_int
    lib.llama_model_quantize_default_params.restype = QuantizeParams
    return lib


def quantize(lib, t, x):
    x = np.ascontiguousarray(x, np.float32)
    out = np.full(O.row_size(t, x.size), 0xAA, np.uint8)
    assert lib.pipeinfer_quantize_row(t, x.ctypes.data, out.ctypes.data, x.size) == out.size
    return out


@pytest.mark.parametrize("t,name", TYPES)
def test_rows_equal_the_reference_golden_bytes(lib, t, name):
    g = np.load(os.path.join(GOLD, "quantize.npz"))
    for i, x in enumerate(g["x"]):
        assert np.array_equal(quantize(lib, t, x), g[name][i]), (name, i)


@pytest.mark.parametrize("t,name", TYPES)
def test_rows_equal_the_reference_functions(lib, t, name):
    if not O.have_ref() or O.ref() is None:
        pytest.skip("oracle/_ref is not built here")
    r = O.ref()
    rng = np.random.default_rng(100 + t)
    k = 256 * 24
    for trial in range(16):
        x = (rng.standard_normal(k) * 10.0 ** rng.integers(-7, 4)).astype(np.float32)
        if trial % 4 == 1:
            x *= rng.choice([0.0, 1.0], k, p=[0.7, 0.3]).astype(np.float32)                  # sparse rows: groups whose scale rounds to zero keep the first guess
        if trial % 4 == 2:
            x = np.abs(x)                                                                       # one-sided: min = 0 branch of the affine fit
        want = np.zeros(O.row_size(t, k), np.uint8)
        r.ref_quantize_reference(t, O._ptr(x), O._ptr(want), k)
        assert np.array_equal(quantize(lib, t, x), want), (name, trial)


@pytest.mark.parametrize("t,name", TYPES)
def test_dequantize_equals_the_reference(lib, t, name):
    """pipeinfer_dequantize_row (what --allow-requantize reads with) against dequantize_row_* of the reference: the golden blocks, and blocks of random bytes"""
    lib.pipeinfer_dequantize_row.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    lib.pipeinfer_dequantize_row.restype = C.c_int64
    g = np.load(os.path.join(GOLD, "quantize.npz"))
    k = g["x"].shape[1]
    for i in range(g["x"].shape[0]):
        blocks = np.ascontiguousarray(g[name][i])
        y = np.zeros(k, np.float32)
        assert lib.pipeinfer_dequantize_row(t, blocks.ctypes.data, y.ctypes.data, k) == k
        x = g["x"][i]
        tol = np.abs(x).max() * {"q2_K": 1.1, "q3_K": 0.6, "q8_0": 0.01}.get(name, 0.3) + 1e-6      # (a sanity bound: it IS the inverse of the quantiser; tiny rows sit in the f16 subnormals of d)
        assert np.abs(y - x).max() <= tol, (name, i)
    if not O.have_ref() or O.ref() is None:
        return
    r = O.ref()
    rng = np.random.default_rng(300 + t)
    k = 256 * 16
    n = O.row_size(t, k)
    for trial in range(8):
        blocks = rng.integers(0, 256, n, dtype=np.uint8)
        h = blocks.view(np.uint16) if n % 2 == 0 else None
        if h is not None:                                         # keep every f16 finite (random bytes make NaN / inf scales, whose payload bits are not the point)
            h[(h & 0x7c00) == 0x7c00] &= 0xbfff
        want, got = np.zeros(k, np.float32), np.zeros(k, np.float32)
        r.ref_dequantize(t, O._ptr(blocks), O._ptr(want), k)
        assert lib.pipeinfer_dequantize_row(t, blocks.ctypes.data, got.ctypes.data, k) == k
        assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (name, trial)


def test_bad_arguments(lib):
    x = np.zeros(300, np.float32)
    out = np.zeros(4096, np.uint8)
    assert lib.pipeinfer_quantize_row(12, x.ctypes.data, out.ctypes.data, 300) == -1           # not whole super-blocks
    assert lib.pipeinfer_quantize_row(5, x.ctypes.data, out.ctypes.data, 256) == -1            # not a type
    assert lib.pipeinfer_quantize_row(1, x.ctypes.data, out.ctypes.data, 300) == 600           # f16: any length


@pytest.mark.parametrize("spec,ftype", [("tiny-llama-b", "q4_k_m"), ("tiny-llama", "q3_k_m"), ("tiny-llama", "q2_k"), ("tiny-llama", "q5_k_s"), ("tiny-llama", "q6_k"),
                                        ("tiny-llama", "q5_1"), ("tiny-falcon40", "q4_k_m"), ("tiny-falcon40", "q5_k_m"), ("tiny-falcon", "q4_0")])
def test_model_quantize_follows_the_mixture_rules(lib, tmp_path, spec, ftype):
    """an fp32 model goes through llama_model_quantize: every 2-D *.weight tensor gets the type the reference's rules give it (independent
    restatement: synth.tensor_plan) and the bytes the row quantiser gives its rows; everything else and all metadata pass through"""
    src, dst = str(tmp_path / "f32.gguf"), str(tmp_path / "q.gguf")
    synth.write_model(src, spec, "f32", seed=5, structure="random")
    p = lib.llama_model_quantize_default_params()
    p.ftype, p.nthread = FTYPES[ftype], 3
    assert lib.llama_model_quantize(src.encode(), dst.encode(), C.byref(p)) == 0
    kv0, t0 = G.read_gguf(src)
    kv1, t1 = G.read_gguf(dst)
    assert kv1["general.file_type"] == FTYPES[ftype] and kv1["general.quantization_version"] == 2
    for k, v in kv0.items():
        if k not in ("general.file_type", "general.quantization_version"):
            assert np.array_equal(np.asarray(kv1[k]), np.asarray(v)), k
    assert list(t0) == list(t1)
    want_types = {name: tt for name, _, tt, _ in synth.tensor_plan(spec, ftype)}
    n_q = 0
    for name, (ne, tt, data) in t1.items():
        ne0, tt0, data0 = t0[name]
        assert ne == ne0
        assert tt == want_types[name], (name, tt, want_types[name])
        if tt == tt0:
            assert np.array_equal(data, data0), name
        else:
            n_q += 1
            x = np.frombuffer(bytes(data0), np.float32)
            assert np.array_equal(np.asarray(data), quantize(lib, tt, x)), name
    assert n_q > 0
    # pure: one type for every weight matrix, no mixture; output.weight stays when asked to
    p.pure, p.quantize_output_tensor = True, False
    dst2 = str(tmp_path / "pure.gguf")
    assert lib.llama_model_quantize(src.encode(), dst2.encode(), C.byref(p)) == 0
    _, t2 = G.read_gguf(dst2)
    base = {"q4_0": 2, "q4_1": 3, "q5_0": 6, "q5_1": 7, "q8_0": 8, "q2_k": 10, "q3_k_m": 11, "q4_k_s": 12, "q4_k_m": 12, "q5_k_s": 13, "q5_k_m": 13, "q6_k": 14}[ftype]
    for name, (ne, tt, _) in t2.items():
        if name == "output.weight" or len(ne) != 2 or not name.endswith("weight"):
            assert tt == t0[name][1], name
        else:
            assert tt == base, (name, tt)


def test_model_quantize_refuses_what_it_cannot_do(lib, tmp_path):
    src, dst = str(tmp_path / "q.gguf"), str(tmp_path / "qq.gguf")
    synth.write_model(src, "tiny-llama", "q4_0", seed=5)
    p = lib.llama_model_quantize_default_params()
    p.ftype = FTYPES["q4_k_m"]
    assert lib.llama_model_quantize(src.encode(), dst.encode(), C.byref(p)) == 1               # requantising is disabled by default (llama.cpp:8159-8161)
    p.ftype = 99
    assert lib.llama_model_quantize(src.encode(), dst.encode(), C.byref(p)) == 1
    p.ftype = FTYPES["q4_0"]
    assert lib.llama_model_quantize(str(tmp_path / "missing.gguf").encode(), dst.encode(), C.byref(p)) == 1
    p.ftype, p.allow_requantize = FTYPES["q4_k_m"], True                                          # --allow-requantize: dequantise, then quantise (llama.cpp:7838-7893, 8162-8165)
    assert lib.llama_model_quantize(src.encode(), dst.encode(), C.byref(p)) == 0
    lib.pipeinfer_dequantize_row.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int64]
    _, t0 = G.read_gguf(src)
    _, t1 = G.read_gguf(dst)
    ne, tt, data = t1["blk.0.ffn_up.weight"]
    assert tt == 12 and t0["blk.0.ffn_up.weight"][1] == 2
    src_blocks = np.ascontiguousarray(t0["blk.0.ffn_up.weight"][2])
    f32 = np.zeros(ne[0] * ne[1], np.float32)
    assert lib.pipeinfer_dequantize_row(2, src_blocks.ctypes.data, f32.ctypes.data, f32.size) == f32.size
    assert np.array_equal(np.asarray(data), quantize(lib, 12, f32))
    p.ftype, p.allow_requantize = FTYPES["q4_0"], False
    p.only_copy = True                                                                            # a copy keeps every tensor and the input's file type
    assert lib.llama_model_quantize(src.encode(), dst.encode(), C.byref(p)) == 0
    kv0, t0 = G.read_gguf(src)
    kv1, t1 = G.read_gguf(dst)
    assert kv1["general.file_type"] == kv0["general.file_type"]
    assert all(np.array_equal(t1[n][2], t0[n][2]) and t1[n][1] == t0[n][1] for n in t0)


def test_quantize_command_line(lib, tmp_path):
    """examples/quantize_amd.cpp: the argument forms of the reference's examples/quantize (type by name or id, optional output name, thread count)"""
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "quantize_amd")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "quantize_amd.cpp"),
                    "-L" + os.path.join(root, "pipeinfer_amd"), "-lpipeinfer_amd", "-Wl,-rpath," + os.path.join(root, "pipeinfer_amd"), "-o", exe], check=True)
    src = str(tmp_path / "f32.gguf")
    synth.write_model(src, "tiny-llama", "f32", seed=5, structure="random")
    assert subprocess.run([exe, src, str(tmp_path / "a.gguf"), "Q4_K_M", "2"], capture_output=True).returncode == 0
    assert subprocess.run([exe, "--pure", src, "15"], capture_output=True).returncode == 0          # default output name beside the input, type by id
    kv_a, t_a = G.read_gguf(str(tmp_path / "a.gguf"))
    kv_b, t_b = G.read_gguf(str(tmp_path / "ggml-model-Q4_K_M.gguf"))
    assert kv_a["general.file_type"] == kv_b["general.file_type"] == 15
    assert t_a["blk.1.attn_v.weight"][1] == 14 and t_b["blk.1.attn_v.weight"][1] == 12             # mixture vs --pure
    assert subprocess.run([exe, src, "Q9_X"], capture_output=True).returncode == 1
