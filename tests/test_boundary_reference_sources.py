"""CPU: the drop-in claim, checked with the reference's OWN sources (skipped where /root/reference does not exist, i.e. on the GPU box):
  * examples/speculative/speculative.cpp + common/{common,sampling,grammar-parser}.cpp compile unmodified against include/llama.h
    (-std=c++11, as the reference builds them, -DGGML_USE_MPI like its PipeInfer build) and LINK against libpipeinfer_amd.so;
  * the reference's tests/test-sampling.cpp (known-answer probabilities for top-k / top-p / tail-free / typical / repetition
    penalties) runs green against the library's host-side samplers.
and, everywhere, the same known answers (data copied from that test's main(): input probabilities -> expected probabilities)
through the C ABI by ctypes."""
import ctypes as C
import os
import shutil
import subprocess

import numpy as np
import pytest

from pipeinfer_amd import lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
LIBDIR = os.path.join(ROOT, "pipeinfer_amd")
need_ref = pytest.mark.skipif(not os.path.isdir(REF) or shutil.which("g++") is None, reason="needs the reference sources and a host compiler")


@need_ref
def test_reference_driver_compiles_and_links_against_the_boundary():
    L.load()                                                          # builds the library if needed
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "refdriver"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    exe = os.path.join(ROOT, "oracle", "_ref", "speculative_ref")
    assert os.path.exists(exe)
    # every llama_* / ggml_* symbol the reference's objects import is provided by our library
    und = subprocess.run(["nm", "-D", "--undefined-only", exe], capture_output=True, text=True).stdout
    wanted = sorted({l.split()[-1] for l in und.splitlines() if l.split() and l.split()[-1].startswith(("llama_", "ggml_"))})
    assert len(wanted) >= 40, wanted
    lib = L.load()
    assert not [s for s in wanted if not hasattr(lib, s)]
    # without a GPU the binary must fail loudly at model load / first decode, not crash in the dynamic linker
    if lib.pi_device_count() == 0:
        p = subprocess.run([exe, "-m", os.path.join(ROOT, "tests", "golden", "tiny_llama_b_q4_k_m.gguf"), "-md", os.path.join(ROOT, "tests", "golden", "tiny_llama_q4_k_m.gguf"),
                            "-p", "hello", "-n", "4", "--mpi-layer-split", "1.0/1.0", "--log-disable"], capture_output=True, text=True, errors="replace", timeout=120)
        assert p.returncode != 0 or "no HIP device" in (p.stderr + p.stdout)


def test_boundary_struct_layouts_are_compile_time_checked():
    """examples/speculative_min.cpp carries static_asserts on sizeof / offsetof of llama_batch (passed by value: llama.h:142-162) and llama_token_data: they hold
    against include/llama.h, and — where the reference's sources exist — the very same asserts hold against the reference's own llama.h"""
    if shutil.which("g++") is None:
        pytest.skip("no host compiler")
    src = os.path.join(ROOT, "examples", "speculative_min.cpp")
    text = open(src).read()
    assert "static_assert(sizeof(llama_batch) == 80" in text and "offsetof(llama_batch, batch_id) == 68" in text
    r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), src], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr[-3000:]
    if os.path.isdir(REF):
        asserts = text[text.index("static_assert(sizeof(llama_batch)"):text.index("static void batch_add")]
        probe = "#include <cstddef>\n" + asserts
        r = subprocess.run(["g++", "-std=c++11", "-fsyntax-only", "-I" + REF, "-include", os.path.join(REF, "llama.h"), "-x", "c++", "-"], input=probe, capture_output=True, text=True, timeout=120)
        assert r.returncode == 0, r.stderr[-3000:]


@need_ref
def test_reference_sampling_known_answers_run_against_our_samplers(tmp_path):
    exe = str(tmp_path / "test-sampling")
    L.load()
    subprocess.run(["g++", "-std=c++11", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(REF, "tests", "test-sampling.cpp"),
                    "-L" + LIBDIR, "-lpipeinfer_amd", "-Wl,-rpath," + LIBDIR, "-o", exe], check=True, timeout=300)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert r.returncode == 0 and r.stdout.strip().endswith("OK"), (r.stdout[-1500:], r.stderr[-1500:])


# ---- the same known answers as data (tests/test-sampling.cpp:131-158), through ctypes -------------------------------------------------
class TokenData(C.Structure):
    _fields_ = [("id", C.c_int32), ("logit", C.c_float), ("p", C.c_float)]


class TokenDataArray(C.Structure):
    _fields_ = [("data", C.POINTER(TokenData)), ("size", C.c_size_t), ("sorted", C.c_bool)]


def _cands(probs):
    arr = (TokenData * len(probs))(*[TokenData(i, float(np.log(np.float32(p))), 0.0) for i, p in enumerate(probs)])
    return arr, TokenDataArray(arr, len(probs), False)


def _probs(c):
    return [c.data[i].p for i in range(c.size)]


@pytest.fixture(scope="module")
def lib():
    lib = L.load()
    vp = C.c_void_p
    lib.llama_sample_softmax.argtypes = [vp, C.POINTER(TokenDataArray)]
    lib.llama_sample_top_k.argtypes = [vp, C.POINTER(TokenDataArray), C.c_int, C.c_size_t]
    for n in ("llama_sample_top_p", "llama_sample_tail_free", "llama_sample_typical", "llama_sample_min_p"):
        getattr(lib, n).argtypes = [vp, C.POINTER(TokenDataArray), C.c_float, C.c_size_t]
    lib.llama_sample_temp.argtypes = [vp, C.POINTER(TokenDataArray), C.c_float]
    lib.llama_sample_repetition_penalties.argtypes = [vp, C.POINTER(TokenDataArray), C.POINTER(C.c_int32), C.c_size_t, C.c_float, C.c_float, C.c_float]
    lib.llama_sample_token_greedy.argtypes = [vp, C.POINTER(TokenDataArray)]
    lib.llama_sample_token_greedy.restype = C.c_int32
    return lib


@pytest.mark.parametrize("probs,want,k", [([0.1, 0.2, 0.3, 0.4], [0.4], 1), ([0.1, 0.2, 0.3, 0.4], [0.4, 0.3, 0.2], 3)])
def test_top_k(lib, probs, want, k):
    keep, c = _cands(probs)
    lib.llama_sample_softmax(None, C.byref(c)); lib.llama_sample_top_k(None, C.byref(c), k, 1)
    assert np.allclose(_probs(c), want, atol=1e-5) and c.size == len(want)


@pytest.mark.parametrize("probs,want,p", [([0.1, 0.2, 0.3, 0.4], [0.4], 0.0), ([0.1, 0.2, 0.3, 0.4], [0.4, 0.3], 0.7), ([0.1, 0.2, 0.3, 0.4], [0.4, 0.3, 0.2], 0.8),
                                          ([0.1, 0.2, 0.3, 0.4], [0.4, 0.3, 0.2, 0.1], 1.0)])
def test_top_p(lib, probs, want, p):
    keep, c = _cands(probs)
    lib.llama_sample_softmax(None, C.byref(c)); lib.llama_sample_top_p(None, C.byref(c), p, 1)
    assert c.size == len(want) and np.allclose(_probs(c), want, atol=1e-3)


@pytest.mark.parametrize("probs,want,z", [([0.1, 0.15, 0.2, 0.25, 0.3], [0.3], 0.25), ([0.1, 0.15, 0.2, 0.25, 0.3], [0.3, 0.25], 0.75), ([0.1, 0.15, 0.2, 0.25, 0.3], [0.3, 0.25], 0.99)])
def test_tail_free(lib, probs, want, z):
    keep, c = _cands(probs)
    lib.llama_sample_tail_free(None, C.byref(c), z, 1)
    assert c.size == len(want) and np.allclose(_probs(c), want, atol=1e-3)


@pytest.mark.parametrize("probs,want,p", [([0.97, 0.01, 0.01, 0.01], [0.97], 0.5), ([0.4, 0.2, 0.2, 0.2], [0.2, 0.2, 0.2], 0.5)])
def test_typical(lib, probs, want, p):
    keep, c = _cands(probs)
    lib.llama_sample_typical(None, C.byref(c), p, 1)
    assert c.size == len(want) and np.allclose(_probs(c), want, atol=1e-3)


@pytest.mark.parametrize("last,want,rep,freq,pres", [
    ([0], [0.25, 0.25, 0.25, 0.25, 0], 50.0, 0.0, 0.0), ([0, 1, 2], [0.5, 0.5, 0, 0, 0], 50.0, 0.0, 0.0), ([0, 1, 2, 0, 0], [0.5, 0.5, 0, 0, 0], 50.0, 0.0, 0.0),
    ([0], [0.249997, 0.249997, 0.249997, 0.249997, 0.000011], 1.0, 5.0, 5.0), ([0, 1, 2], [0.499966, 0.499966, 0.000023, 0.000023, 0.000023], 1.0, 5.0, 5.0),
    ([0, 1, 2, 0, 0], [0.499977, 0.499977, 0.000023, 0.000023, 0.000000], 1.0, 5.0, 5.0)])
def test_repetition_penalties(lib, last, want, rep, freq, pres):
    keep, c = _cands([0.2] * 5)
    lib.llama_sample_softmax(None, C.byref(c))
    lt = (C.c_int32 * len(last))(*last)
    lib.llama_sample_repetition_penalties(None, C.byref(c), lt, len(last), rep, freq, pres)
    lib.llama_sample_softmax(None, C.byref(c))
    assert c.size == 5 and np.allclose(_probs(c), want, atol=1e-3)


def test_greedy_takes_the_first_maximum(lib):
    arr = (TokenData * 4)(TokenData(7, 1.0, 0), TokenData(3, 2.5, 0), TokenData(9, 2.5, 0), TokenData(1, -1.0, 0))
    c = TokenDataArray(arr, 4, False)
    assert lib.llama_sample_token_greedy(None, C.byref(c)) == 3          # std::max_element: first of equal maxima (llama.cpp:7496-7510)
