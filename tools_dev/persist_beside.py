#!/usr/bin/env python3
"""A one-kernel draft step beside the target: what would it take?  (tools_dev/persist_probe.hip holds the synthetic step.)
    python tools_dev/persist_beside.py [7b|70b]
Keeps two real target runs in flight (like tools_dev/contention.py) and times, beside them: the real draft decode call; the synthetic step as 203 launches; the
synthetic step as one persistent kernel of 64 / 128 / 256 workgroups.  Also reports how many target runs finish meanwhile (what the draft costs the target)."""
import ctypes as C
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
pp = C.CDLL(os.path.join(ROOT, "tools_dev", "libpersist_probe.so"))
pp.pp_step.restype = C.c_float
pp.pp_bytes.restype = C.c_double
n_ph = pp.pp_init()
assert n_ph > 0
print(f"synthetic step: {n_ph} phases, {pp.pp_bytes() / 1e9:.3f} GB", flush=True)
mt, md = LL.Model(tp), LL.Model(dp)
ct, cd = LL.Context(mt, n_ctx=1024, n_batch=64), LL.Context(md, n_ctx=1024, n_batch=64)
cd.set_priority(1)
ct.prepare(); cd.prepare()
rng = np.random.default_rng(1)
prompt = [int(x) for x in rng.integers(3, mt.n_vocab, 32)]
ct.decode(prompt, list(range(32)), [[0]] * 32); cd.decode(prompt, list(range(32)), [[0]] * 32)
dpos = [40]


def real_draft():
    t0 = time.time(); cd.decode([5], [dpos[0]], [[0]]); dpos[0] += 1
    if dpos[0] > 900:
        dpos[0] = 40
    return (time.time() - t0) * 1e3


def synth(mode, wg):
    def f():
        ms = pp.pp_step(mode, wg)
        assert ms > 0
        return ms
    return f


variants = [("real draft decode call (host clock)", real_draft), ("synthetic, 203 launches of 256 workgroups", synth(0, 256)), ("synthetic, 203 launches of 1024 workgroups", synth(0, 1024)),
            ("synthetic, persistent 64 workgroups", synth(1, 64)), ("synthetic, persistent 128 workgroups", synth(1, 128)), ("synthetic, persistent 256 workgroups", synth(1, 256)),
            ("synthetic, persistent 512 workgroups", synth(1, 512))]
for i in range(3):
    ct.decode([7] * n_tok_t, [32 + i * n_tok_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)
t0 = time.time()
for i in range(5):
    ct.decode([7] * n_tok_t, [40 + i * n_tok_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)
t_alone = (time.time() - t0) / 5 * 1e3
print(f"target call of {n_tok_t} token(s) alone: {t_alone:.2f} ms", flush=True)
for name, fn in variants:
    for _ in range(3):
        fn()
    alone = np.mean([fn() for _ in range(20)])
    pos_t = 100
    fl = []
    for _ in range(2):
        fl.append(ct.start([7] * n_tok_t, [pos_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)); pos_t += n_tok_t
    lat, n_t = [], 0
    t_begin = time.time()
    for i in range(60):
        lat.append(fn())
        while ct.iprobe() and fl:
            ct.finish(fl.pop(0)); n_t += 1
            fl.append(ct.start([7] * n_tok_t, [pos_t + j for j in range(n_tok_t)], [[0]] * n_tok_t)); pos_t += n_tok_t
            if pos_t > 900:
                pos_t = 100
    dt = (time.time() - t_begin) * 1e3
    for f in fl:
        ct.finish(f)
    print(f"{name:46s} alone {alone:6.2f} ms | beside the target {np.mean(lat):6.2f} ms (min {np.min(lat):.2f}, max {np.max(lat):.2f}) | target: {n_t} runs in {dt:.0f} ms = {dt / max(1, n_t):.2f} ms per run "
          f"| draft steps + target runs per second: {60 / dt * 1e3:.0f} + {n_t / dt * 1e3:.0f}", flush=True)
