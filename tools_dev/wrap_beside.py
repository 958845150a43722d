#!/usr/bin/env python3
"""Which part of a draft CALL keeps it from running beside an unsegmented target run?  The synthetic step of persist_probe.hip wrapped in what the engine's call has around
its graph (pp_wrap bits: 1 H2D copies ahead, 2 D2H copies behind, 4 a one-thread kernel first, 8 copies by kernels on mapped pinned memory)."""
import ctypes as C
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pipeinfer_amd import lib as L, llama as LL  # noqa: E402

tspec, tft, dspec, dft = bench.WORKLOADS["70b"]
cache = "/tmp/pipeinfer_models"
tp = bench.ensure_model(cache, tspec, tft, 0x5EED, 1.0, 0.8)
lib = L.load()
L.check(lib.pi_device_init(0))
pp = C.CDLL(os.path.join(ROOT, "tools_dev", "libpersist_probe.so"))
pp.pp_step.restype = C.c_float
assert pp.pp_init() > 0
mt = LL.Model(tp)
ct = LL.Context(mt, n_ctx=1024, n_batch=64)
ct.prepare()
rng = np.random.default_rng(1)
prompt = [int(x) for x in rng.integers(3, mt.n_vocab, 32)]
ct.decode(prompt, list(range(32)), [[0]] * 32)
for i in range(3):
    ct.decode([7], [32 + i], [[0]])
for wrap in (0, 4, 1, 2, 3, 7, 8 + 7, 8 + 3):
    pp.pp_wrap(wrap)
    for _ in range(3):
        pp.pp_step(0, 256)
    alone = np.mean([pp.pp_step(0, 256) for _ in range(20)])
    pos_t = 100
    fl = [ct.start([7], [pos_t], [[0]])]; pos_t += 1
    lat, host, n_t = [], [], 0
    t_begin = time.time()
    for i in range(40):
        t0 = time.time()
        lat.append(pp.pp_step(0, 256)); host.append((time.time() - t0) * 1e3)
        while ct.iprobe() and fl:
            ct.finish(fl.pop(0)); n_t += 1
            fl.append(ct.start([7], [pos_t], [[0]])); pos_t += 1
            if pos_t > 900:
                pos_t = 100
    dt = (time.time() - t_begin) * 1e3
    for f in fl:
        ct.finish(f)
    print(f"wrap {wrap:2d}: step alone {alone:5.2f} ms | beside one unsegmented target run: {np.mean(lat):6.2f} ms GPU span, {np.mean(host):6.2f} ms host (max {np.max(host):.2f}) | target {dt / max(1, n_t):6.2f} ms per run", flush=True)
