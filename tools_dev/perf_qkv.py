"""micro-benchmark of the attention projections as ONE pi_op_mul_mat_multi call (q, k in Q4_K; v in Q5_K / Q6_K: the Q4_K_M mixture of a 70B layer) at verify and prompt widths:
    NTS=6,64 python3 tools_dev/perf_qkv.py          (PI_MMQ_2T=0: attn_v as a launch of its own)"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U, oracle_lib as O
from pipeinfer_amd.tools import synth
from pipeinfer_amd import lib as L
lib = U.init()
k = int(os.environ.get("K", "8192")); rows = [int(x) for x in os.environ.get("ROWS", "8192,1024,1024").split(",")]
nts = [int(x) for x in os.environ.get("NTS", "1,6,64").split(",")]
for vt in (O.Q6_K, O.Q5_K):
    types = (O.Q4_K, O.Q4_K, vt)
    rng = np.random.default_rng(1)
    ncopy = 8
    sets = []
    for _ in range(ncopy):
        sets.append([lib.pi_weight_upload(t, k, m, synth.random_blocks(t, m * k, rng, 0.05).ctypes.data_as(C.c_void_p), None) for t, m in zip(types, rows)])
    for nt in nts:
        act = rng.standard_normal((nt, k)).astype(np.float32)
        xd = U.Dev(act); q8 = U.Dev(nbytes=lib.pi_q8_nbytes(O.Q8_K, k, nt)); outs = [U.Dev(nbytes=4 * nt * m) for m in rows]
        L.check(lib.pi_op_quantize(O.Q8_K, xd.ptr, k, nt, q8.ptr, None))
        ol = (C.c_void_p * 3)(*[o.ptr for o in outs])
        wls = [(C.c_void_p * 3)(*ws) for ws in sets]
        for i in range(5):
            L.check(lib.pi_op_mul_mat_multi(3, wls[i % ncopy], q8.ptr, nt, ol, None, None))
        lib.pi_device_sync()
        n = 300
        t0 = time.perf_counter()
        for i in range(n):
            lib.pi_op_mul_mat_multi(3, wls[i % ncopy], q8.ptr, nt, ol, None, None)
        lib.pi_device_sync()
        dt = (time.perf_counter() - t0) / n
        print(f"q+k Q4_K + v {O.TYPE_NAMES[vt]} rows={rows} K={k} NT={nt}: {dt * 1e6:7.2f} us per call (PI_MMQ_2T={os.environ.get('PI_MMQ_2T', '1')})")
