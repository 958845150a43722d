"""one shape, one width, N launches over rotating weight copies (for rocprofv3 --pmc / --kernel-trace runs): python tools_dev/perf_one.py M K NT [n]"""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U, oracle_lib as O
from pipeinfer_amd.tools import synth
from pipeinfer_amd import lib as L
M, K, NT = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]); n = int(sys.argv[4]) if len(sys.argv) > 4 else 40
t = {"q4": O.Q4_K, "q6": O.Q6_K, "q5": O.Q5_K}[os.environ.get("TYPE", "q4")]
lib = U.init()
rng = np.random.default_rng(1)
blocks = synth.random_blocks(t, M * K, rng, 0.05)
ncopy = max(1, int(400e6 // blocks.nbytes) + 1)
ws = [lib.pi_weight_upload(t, K, M, blocks.ctypes.data_as(C.c_void_p), None) for _ in range(ncopy)]
act = rng.standard_normal((NT, K)).astype(np.float32)
xd = U.Dev(act); q8 = U.Dev(nbytes=lib.pi_q8_nbytes(O.Q8_K, K, NT)); out = U.Dev(nbytes=4 * NT * M)
L.check(lib.pi_op_quantize(O.Q8_K, xd.ptr, K, NT, q8.ptr, None))
for i in range(n):
    lib.pi_op_mul_mat(ws[i % ncopy], q8.ptr, NT, out.ptr, None, None)
lib.pi_device_sync()
print("done", M, K, NT, n)
