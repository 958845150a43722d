"""micro-benchmark of the attention kernel as a function of n_kv (not a test): python tools_dev/perf_att.py"""
import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))  # repo root
from tests import gpu_util as U
from pipeinfer_amd import lib as L
lib = U.init()
hd, n_head, n_head_kv, n_ctx = 128, 32, 32, 1024
rng = np.random.default_rng(2)
n_layer = int(os.environ.get("NLAYER", "32"))               # rotate over layers as a decode does
kcs = [U.Dev(rng.standard_normal((n_ctx, n_head_kv * hd)).astype(np.float16)) for _ in range(n_layer)]
vcs = [U.Dev(rng.standard_normal((n_ctx, n_head_kv * hd)).astype(np.float16)) for _ in range(n_layer)]
q = U.Dev(rng.standard_normal((1, n_head * hd)).astype(np.float16))
out = U.Dev(nbytes=4 * n_head * hd)
for n_kv in [int(x) for x in os.environ.get("NKV", "32,64,96,128,129,136,144,152,159,160,161,176,192,224,256,384,512,1024").split(",")]:
    words = (n_kv + 31) // 32
    mask = U.Dev(np.full((1, words), 0xFFFFFFFF, dtype=np.uint32))
    for i in range(64):
        lib.pi_op_attention(q.ptr, kcs[i % n_layer].ptr, vcs[i % n_layer].ptr, mask.ptr, hd, n_head, n_head_kv, 1, n_kv, out.ptr, None, None)
    lib.pi_device_sync()
    n = 640
    t0 = time.perf_counter()
    for i in range(n):
        lib.pi_op_attention(q.ptr, kcs[i % n_layer].ptr, vcs[i % n_layer].ptr, mask.ptr, hd, n_head, n_head_kv, 1, n_kv, out.ptr, None, None)
    lib.pi_device_sync()
    dt = (time.perf_counter() - t0) / n
    print(f"n_kv={n_kv:5d}: {dt*1e6:7.2f} us/launch (back-to-back launches, includes launch gaps)")
