"""micro-benchmark of the mat-vec kernel per shape (not a test): python tests/perf_mmvq.py"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from tests import gpu_util as U, oracle_lib as O
from pipeinfer_amd.tools import synth
from pipeinfer_amd import lib as L
lib = U.init()
shapes = [(O.Q4_K, 4096, 4096), (O.Q4_K, 11008, 4096), (O.Q4_K, 4096, 11008), (O.Q6_K, 32000, 4096), (O.Q4_K, 2048, 2048), (O.Q4_K, 5632, 2048), (O.Q4_K, 256, 2048), (O.Q6_K, 4096, 11008), (O.Q4_0, 4096, 4096), (O.Q4_K, 22016, 4096), (O.Q3_K, 4096, 4096), (O.Q3_K, 22016, 4096), (O.Q3_K, 4096, 11008), (O.Q5_K, 4096, 4096), (O.Q8_0, 4096, 4096), (O.Q4_K, 8192, 8192), (O.Q4_K, 28672, 8192), (O.Q4_K, 8192, 28672), (O.Q2_K, 4096, 4096), (O.Q2_K, 22016, 4096)]
if len(sys.argv) > 1:                               # indices into the list above, or explicit shapes "q5_K:8192:8192"
    names = {v: k for k, v in O.TYPE_NAMES.items()}
    shapes = [shapes[int(i)] if i.isdigit() else (names[i.split(":")[0]], int(i.split(":")[1]), int(i.split(":")[2])) for i in sys.argv[1].split(",")]
nts = [int(x) for x in os.environ.get("NTS", "1,2,4,8").split(",")]
for t, rows, k in shapes:
    rng = np.random.default_rng(1)
    blocks = synth.random_blocks(t, rows * k, rng, 0.05)
    # rotate over enough copies to exceed the 256 MB infinity cache: decode streams the whole model, it never re-reads
    ncopy = int(os.environ["NCOPY"]) if os.environ.get("NCOPY") else max(1, int(400e6 // blocks.nbytes) + 1)
    ws = [lib.pi_weight_upload(t, k, rows, blocks.ctypes.data_as(C.c_void_p), None) for _ in range(ncopy)]
    w = ws[0]
    for nt in nts:
        act = rng.standard_normal((nt, k)).astype(np.float32)
        vdt = O.VEC_DOT_TYPE[t]
        xd = U.Dev(act); q8 = U.Dev(nbytes=lib.pi_q8_nbytes(vdt, k, nt)); out = U.Dev(nbytes=4 * nt * rows)
        L.check(lib.pi_op_quantize(vdt, xd.ptr, k, nt, q8.ptr, None))
        for _ in range(5):
            lib.pi_op_mul_mat(w, q8.ptr, nt, out.ptr, None, None)
        lib.pi_device_sync()
        n = int(os.environ.get("NLAUNCH", "200"))
        t0 = time.perf_counter()
        for i in range(n):
            lib.pi_op_mul_mat(ws[i % ncopy], q8.ptr, nt, out.ptr, None, None)
        lib.pi_device_sync()
        dt = (time.perf_counter() - t0) / n
        nbytes = lib.pi_weight_nbytes(w)
        print(f"{O.TYPE_NAMES[t]:5s} M={rows:6d} K={k:6d} NT={nt}: {dt*1e6:7.2f} us/launch  {nbytes/dt/1e9:8.1f} GB/s  ({nbytes/1e6:.1f} MB)")
    for w in ws:
        lib.pi_weight_free(w)
