"""CPU (-m "not gpu"): the C-ABI library builds, loads without a GPU, exports every symbol include/*.h declares, and its
compute entry points fail loudly (no CPU fallback) when no MI355X is visible."""
import ctypes as C
import os

import pytest

from pipeinfer_amd import lib as L


def test_library_loads_and_exports_every_declared_symbol():
    lib = L.load()
    names = L.declared_symbols()
    assert len(names) >= 25
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_no_gpu_means_loud_failure():
    lib = L.load()
    if lib.pi_device_count() > 0:
        pytest.skip("a GPU is visible")
    assert lib.pi_device_init(0) != 0
    assert b"no HIP device" in lib.pi_last_error()


def test_product_never_touches_the_oracle():
    """the oracle is test infrastructure: nothing under pipeinfer_amd/ may reference it"""
    root = os.path.dirname(os.path.abspath(L.__file__))
    for dp, _, files in os.walk(root):
        if "build" in dp.split(os.sep):
            continue
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h", "Makefile")):
                text = open(os.path.join(dp, f), errors="ignore").read()
                assert "liboracle" not in text and "oracle/" not in text and "oracle_lib" not in text, os.path.join(dp, f)
