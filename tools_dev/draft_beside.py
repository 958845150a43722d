#!/usr/bin/env python3
"""The REAL draft decode call beside real target runs: host clock and GPU span per call, what the target pays.
    python tools_dev/draft_beside.py [70b] [target tokens per run]
Variants: target as one graph or in segments of 20 layers, one or two target runs in flight, draft stream priority on / off.  (PI_NO_GRAPHS=1 in the environment: eager launches.)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from pipeinfer_amd import lib as L, llama as LL  # noqa: E402

wl = sys.argv[1] if len(sys.argv) > 1 else "70b"
n_tok_t = int(sys.argv[2]) if len(sys.argv) > 2 else 1
tspec, tft, dspec, dft = bench.WORKLOADS[wl]
cache = "/tmp/pipeinfer_models"
tp = bench.ensure_model(cache, tspec, tft, 0x5EED, 1.0, 0.8)
dp = bench.ensure_model(cache, dspec, dft, 0xD4AF, 1.0, 0.8)
lib = L.load()
L.check(lib.pi_device_init(0))
mt, md = LL.Model(tp), LL.Model(dp)
rng = np.random.default_rng(1)
prompt = [int(x) for x in rng.integers(3, mt.n_vocab, 32)]
print(f"graphs {'off' if os.environ.get('PI_NO_GRAPHS') else 'on'}; target runs of {n_tok_t} token(s)", flush=True)


def probe(name, seg, inflight, prio_d):
    ct, cd = LL.Context(mt, n_ctx=1024, n_batch=64), LL.Context(md, n_ctx=1024, n_batch=64)
    cd.set_priority(prio_d)
    if seg:
        ct.set_segments(seg)
    ct.prepare(); cd.prepare()
    ct.decode(prompt, list(range(32)), [[0]] * 32); cd.decode(prompt, list(range(32)), [[0]] * 32)
    for i in range(3):
        cd.decode([5], [32 + i], [[0]])
        ct.decode([7] * n_tok_t, [32 + i * n_tok_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)
    cd.reset_stats()
    t0 = time.time()
    for i in range(20):
        cd.decode([5], [35 + i], [[0]])
    alone = (time.time() - t0) / 20 * 1e3
    sd = cd.stats(); alone_gpu = sd["gpu_ms"] / max(1, sd["n_calls"])
    t0 = time.time()
    for i in range(5):
        ct.decode([7] * n_tok_t, [48 + i * n_tok_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)
    t_alone = (time.time() - t0) / 5 * 1e3
    pos_t = 100
    fl = []
    for _ in range(inflight):
        fl.append(ct.start([7] * n_tok_t, [pos_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)); pos_t += n_tok_t
    cd.reset_stats(); ct.reset_stats()
    lat, n_t = [], 0
    t_begin = time.time()
    for i in range(60):
        t0 = time.time()
        cd.decode([5], [60 + i], [[0]])
        lat.append((time.time() - t0) * 1e3)
        while ct.iprobe() and fl:
            ct.finish(fl.pop(0)); n_t += 1
            fl.append(ct.start([7] * n_tok_t, [pos_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)); pos_t += n_tok_t
            if pos_t > 900:
                pos_t = 100
    dt = (time.time() - t_begin) * 1e3
    for f in fl:
        ct.finish(f)
    sd, st = cd.stats(), ct.stats()
    print(f"{name:58s} draft alone {alone:5.2f} ms host / {alone_gpu:5.2f} GPU | target alone {t_alone:6.2f} | beside: draft {np.mean(lat):6.2f} ms host (min {np.min(lat):.2f}) / "
          f"{sd['gpu_ms'] / max(1, sd['n_calls']):5.2f} GPU span | target {dt / max(1, n_t):6.2f} ms per run ({st['gpu_ms'] / max(1, st['n_calls']):.2f} GPU span) | "
          f"work done vs one after the other: {(60 * alone_gpu + n_t * t_alone) / dt:.2f}x", flush=True)
    ct.free(); cd.free()


probe("one graph per target run, 2 in flight, draft high", 0, 2, 1)
probe("one graph per target run, 1 in flight, draft high", 0, 1, 1)
probe("segments of 20 layers, 2 runs in flight, draft high", 20, 2, 1)
probe("segments of 20 layers, 1 run in flight, draft high", 20, 1, 1)
probe("segments of 20 layers, 2 runs in flight, draft default", 20, 2, 0)
probe("segments of 10 layers, 2 runs in flight, draft high", 10, 2, 1)
