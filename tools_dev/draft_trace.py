#!/usr/bin/env python3
"""Kernel-trace companion of draft_beside.py: a few draft decode calls beside ONE unsegmented (or segmented: argv[1] = layers per segment) target run in flight.
   rocprofv3 --kernel-trace -d gpurun_out/dtrace -o dt --output-format csv -- python3 tools_dev/draft_trace.py 0
   python3 tools_dev/draft_trace.py --analyse gpurun_out/dtrace/.../dt_kernel_trace.csv"""
import csv
import os
import sys
import time
from collections import defaultdict

if len(sys.argv) > 2 and sys.argv[1] == "--analyse":
    rows = list(csv.DictReader(open(sys.argv[2])))
    by_q = defaultdict(list)
    for r in rows:
        by_q[(r.get("Queue_Id"), r.get("Stream_Id"))].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60]))
    print("queues / streams:", {k: len(v) for k, v in by_q.items()})
    qs = sorted(by_q, key=lambda k: -len(by_q[k]))
    tq, dq = qs[0], qs[1]                                   # most launches: the target; next: the draft
    T = sorted(by_q[tq]); D = sorted(by_q[dq])
    t_end = T[-1][1]
    # split the draft's kernels into calls: a gap of more than 300 us between consecutive kernels' starts... (calls are back to back: use the top-k kernel as the end marker)
    calls, cur = [], []
    for k in D:
        cur.append(k)
        if "k_topk2" in k[2]:
            calls.append(cur); cur = []
    print(f"target kernels {len(T)}, draft kernels {len(D)}, draft calls {len(calls)}")
    for ci, c in enumerate(calls[-12:]):
        if len(c) < 50:
            continue
        s0 = c[0][0]
        busy = sum(e - s for s, e, _ in c) / 1e3
        inside = sum(1 for s, e, _ in T if s < c[-1][1] and e > s0)
        gaps = sorted(((c[i + 1][0] - c[i][1]) / 1e3, i, c[i][2]) for i in range(len(c) - 1))
        print(f"call {ci}: {len(c)} kernels, span {(c[-1][1] - s0) / 1e3:8.1f} us, kernels' own time {busy:7.1f} us, target kernels overlapping the span: {inside}; largest gaps (us, after kernel #, name): "
              + "; ".join(f"{g:.0f} after #{i} {n[:28]}" for g, i, n in gaps[-3:]))
    sys.exit(0)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import bench  # noqa: E402
from pipeinfer_amd import lib as L, llama as LL  # noqa: E402

seg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
tspec, tft, dspec, dft = bench.WORKLOADS["70b"]
cache = "/tmp/pipeinfer_models"
tp = bench.ensure_model(cache, tspec, tft, 0x5EED, 1.0, 0.8)
dp = bench.ensure_model(cache, dspec, dft, 0xD4AF, 1.0, 0.8)
lib = L.load()
L.check(lib.pi_device_init(0))
mt, md = LL.Model(tp), LL.Model(dp)
rng = np.random.default_rng(1)
prompt = [int(x) for x in rng.integers(3, mt.n_vocab, 32)]
ct, cd = LL.Context(mt, n_ctx=1024, n_batch=64), LL.Context(md, n_ctx=1024, n_batch=64)
cd.set_priority(1)
if seg:
    ct.set_segments(seg)
ct.prepare(); cd.prepare()
ct.decode(prompt, list(range(32)), [[0]] * 32); cd.decode(prompt, list(range(32)), [[0]] * 32)
for i in range(2):
    cd.decode([5], [32 + i], [[0]]); ct.decode([7], [32 + i], [[0]])
pos_t = 100
fl = [ct.start([7], [pos_t], [[0]])]; pos_t += 1
for i in range(8):
    t0 = time.time()
    cd.decode([5], [60 + i], [[0]])
    print(f"draft call {i}: {(time.time() - t0) * 1e3:.2f} ms host", flush=True)
    while ct.iprobe() and fl:
        ct.finish(fl.pop(0))
        fl.append(ct.start([7], [pos_t], [[0]])); pos_t += 1
for f in fl:
    ct.finish(f)
