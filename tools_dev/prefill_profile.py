#!/usr/bin/env python3
"""Where a prompt pass spends its GPU time: decodes an n-token prompt on the target model with graphs off and EVERY launch timed (kernel_timing.h), prints the per-instance
table (family, type, tokens of the launch, launch bytes, launches, avg us, GB/s) and the total.   usage: prefill_profile.py [workload=70b] [n_prompt=128] [reps=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench as B
from pipeinfer_amd import lib as L, llama as LL

workload = sys.argv[1] if len(sys.argv) > 1 else "70b"
n_prompt = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 3
tspec, tft, _, _ = B.WORKLOADS[workload]
path = B.ensure_model(os.environ.get("PIPEINFER_CACHE", "/tmp/pipeinfer_models"), tspec, tft, 0x5EED, 1.0, 0.25)
lib = L.load()
L.check(lib.pi_device_init(0))
m = LL.Model(path)
c = LL.Context(m, n_ctx=1024, n_batch=512)
c.prepare()
rng = np.random.default_rng(1)
prompt = [int(x) for x in rng.integers(3, m.n_vocab, n_prompt)]
for graphs in (1, 0):
    c.set_graphs(graphs)
    c.kv_clear(); c.decode(prompt, list(range(n_prompt)), [[0]] * n_prompt)          # warm-up (capture)
    lib.pi_device_sync()
    if not graphs:
        lib.pi_kernel_timing(1, 1)
    t0 = time.time()
    for _ in range(reps):
        c.kv_clear(); c.decode(prompt, list(range(n_prompt)), [[0]] * n_prompt)
    lib.pi_device_sync()
    dt = (time.time() - t0) / reps
    print(f"{workload}: {n_prompt}-token prompt, graphs={graphs}: {dt * 1e3:.1f} ms per decode = {n_prompt / dt:.0f} tok/s (host clock)")
rows = B.timing_rows(lib)
lib.pi_kernel_timing(0, 8)
tot = sum(r["avg_us"] * r["launches"] for r in rows) / reps
print(f"sum of kernel durations per decode: {tot / 1e3:.2f} ms")
fam = {}
for r in sorted(rows, key=lambda r: -r["avg_us"] * r["launches"]):
    us = r["avg_us"] * r["launches"] / reps
    f = B.family_of(r); fam[f] = fam.get(f, 0.0) + us
    print(f"  {f:13s} {B.TYPE_NAMES.get(r['type'], '-'):5s} nt={r['n_tok']:3d} launch_MB={r['shape_bytes'] / 1e6:8.2f} launches/decode={r['launches'] / reps:7.1f} avg_us={r['avg_us']:8.2f} "
          f"GB/s={r['avg_sampled_bytes'] / max(1e-9, r['avg_us']) / 1e3:7.1f} ms/decode={us / 1e3:7.3f} ({100 * us / tot:4.1f} %)")
print({k: round(v / 1e3, 2) for k, v in sorted(fam.items(), key=lambda kv: -kv[1])})
