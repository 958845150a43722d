#!/usr/bin/env python3
"""How long does a draft decode call take while target runs keep the GPU busy?  (one GPU, two contexts, two streams)
    python tools_dev/contention.py [7b|70b]
Variants: stream priorities, a private CU range for the draft."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pipeinfer_amd import lib as L, llama as LL  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "7b"
tspec, tft, dspec, dft = bench.WORKLOADS[wl]
cache = "/tmp/pipeinfer_models"
tp = bench.ensure_model(cache, tspec, tft, 0x5EED, 1.0, 0.8)
dp = bench.ensure_model(cache, dspec, dft, 0xD4AF, 1.0, 0.8)
lib = L.load()
L.check(lib.pi_device_init(0))
mt, md = LL.Model(tp), LL.Model(dp)


def probe(name, prio_t, prio_d, cus=0, n_tok_t=1):
    ct, cd = LL.Context(mt, n_ctx=1024, n_batch=64), LL.Context(md, n_ctx=1024, n_batch=64)
    ct.set_priority(prio_t); cd.set_priority(prio_d)
    if cus:
        cd.set_cu_range(0, cus); ct.set_cu_range(cus, 256)
    ct.prepare(); cd.prepare()
    rng = np.random.default_rng(1)
    prompt = [int(x) for x in rng.integers(3, mt.n_vocab, 32)]
    ct.decode(prompt, list(range(32)), [[0]] * 32); cd.decode(prompt, list(range(32)), [[0]] * 32)
    # alone
    for i in range(3):
        cd.decode([5], [32 + i], [[0]])
    t0 = time.time()
    for i in range(20):
        cd.decode([5], [35 + i], [[0]])
    alone = (time.time() - t0) / 20
    t0 = time.time()
    for i in range(5):
        ct.decode([7] * n_tok_t, [32 + i * n_tok_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)
    t_alone = (time.time() - t0) / 5
    # under load: keep two target runs in flight, time draft calls
    pos_t = 40 + 5 * n_tok_t
    fl = []
    for _ in range(2):
        fl.append(ct.start([7] * n_tok_t, [pos_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)); pos_t += n_tok_t
    lat = []
    n_t = 0
    t_begin = time.time()
    for i in range(40):
        t0 = time.time()
        cd.decode([5], [60 + i], [[0]])
        lat.append(time.time() - t0)
        while ct.iprobe() and fl:
            ct.finish(fl.pop(0)); n_t += 1
            fl.append(ct.start([7] * n_tok_t, [pos_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)); pos_t += n_tok_t
            if pos_t > 900:
                pos_t = 40
    dt = time.time() - t_begin
    for f in fl:
        ct.finish(f)
    print(f"{name:40s} draft alone {alone * 1e3:6.2f} ms | target alone {t_alone * 1e3:6.2f} ms | draft under load {np.mean(lat) * 1e3:6.2f} ms (min {np.min(lat) * 1e3:.2f}) | "
          f"target runs finished meanwhile {n_t} in {dt * 1e3:.0f} ms -> {dt / max(1, n_t) * 1e3:.2f} ms per run", flush=True)
    ct.free(); cd.free()


probe("default priorities", 0, 0)
probe("draft high", 0, 1)
probe("draft high, target low", -1, 1)
probe("draft 32 CUs private", 0, 0, cus=32)
probe("draft 64 CUs private", 0, 0, cus=64)
