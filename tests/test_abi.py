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


def test_hand_counted_weight_ring_of_k_mmq_g2_is_intact_in_the_compiled_code(tmp_path):
    """k_mmq_g2 (mmq_mfma.hip) loads its weight ring with inline asm the compiler cannot see and waits for it with counted `s_waitcnt vmcnt` statements that name the ring registers.
    The compiler must not copy, spill, or reuse such a register while its load is in flight: tools_dev/check_asm_ring.py audits the ISA hipcc emits (every register a ring load
    targets stays untouched until the `landed:` statement that names it), with the library's own flags."""
    import shutil, subprocess, sys
    hipcc = "/opt/rocm/bin/hipcc"
    if not os.path.exists(hipcc):
        pytest.skip("no hipcc on this box (the audit runs where the library is built)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    asm = tmp_path / "mmq.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fvisibility=hidden", "-Wno-unused-function", "-Wno-unused-result",
                    "--cuda-device-only", "-S", os.path.join(root, "pipeinfer_amd", "csrc", "kernels", "mmq_mfma.hip"), "-o", str(asm)], check=True, timeout=900)
    r = subprocess.run([sys.executable, os.path.join(root, "tools_dev", "check_asm_ring.py"), str(asm)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-3000:]
    n = int(r.stdout.strip().splitlines()[-1].split()[0])
    assert n >= 5 * 2 * 4 * 4, r.stdout[-500:]                  # five instances x two row groups x four loads x at least the prologue's four slots
    text = asm.read_text()
    import re
    for m in re.finditer(r"\.name:\s+(_Z8k_mmq_g2\S+)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", text):
        assert int(m.group(3)) == 0, (m.group(1), "spills: a spilled ring register would be stored before its load has landed")
