#!/usr/bin/env python3
"""Where does the target's idle time between two forward passes go?  (round 6)

Runs the bench's default schedule (70B + TinyLlama, mode 0, segments + coalescing as bench.py sets them) for a few generations
with the engine's timeline on (PI_TIMELINE=1, include/pipeinfer.h: pipeinfer_timeline_dump) and prints, for the last one:
  * per forward pass: when the feeder composed it, what it carried, when it left the GPU, how long the GPU's target side sat
    idle before it, and why the reaction window ended (1 caller blocked, 2 caller reacted, 3 time-out);
  * totals: decode wall time = passes' GPU span + idle, idle by cause.

    PI_TIMELINE=1 python tools_dev/timeline.py [--workload 70b] [--gens 3] [--csv out.csv]
"""
import argparse
import collections
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PI_TIMELINE", "1")

import bench as B  # noqa: E402
from pipeinfer_amd import lib as L  # noqa: E402
from pipeinfer_amd import llama as LL  # noqa: E402
from pipeinfer_amd.tools import synth  # noqa: E402


def analyse(path, out=sys.stdout, verbose=True):
    ev = []
    for line in open(path):
        t, tag, a, b, c = line.rstrip("\n").split(",")
        ev.append((int(t), tag, int(a), int(b), int(c)))
    ev.sort(key=lambda e: e[0])
    if not ev:
        print("no events", file=out)
        return
    t0 = ev[0][0]
    passes = []          # dict(id, t_compose, runs, tokens, t_fed, t_done)
    cur = None
    react = collections.Counter()
    last_done = None
    drv = []             # caller-side events between passes
    gpu_us = {}
    for t, tag, a, b, c in ev:
        if tag == "pass":
            cur = dict(id=a, runs=b, tokens=c, t_compose=t, t_fed=None, t_done=None, idle_before=(t - last_done) if last_done else None, react=None, drv=drv)
            drv = []
            passes.append(cur)
        elif tag == "fed" and cur and cur["id"] == a:
            cur["t_fed"] = t
        elif tag == "pass_gpu_done":
            hit = [p for p in passes if p["id"] == a and p["t_done"] is None]
            if hit:
                hit[-1]["t_done"] = t
            last_done = t
        elif tag == "react_end":
            react[a] += 1
            if passes:
                passes[-1]["react_after"] = a
        elif tag == "gpu_us":
            gpu_us[b] = c
        if tag in ("start", "finish", "probe_back", "cancel", "sampled", "specloop", "finish_wait"):
            drv.append((t, tag, a, b, c))
    if verbose:
        print(f"{'t_ms':>9} {'pass':>6} {'runs':>4} {'tok':>3} {'idle_before_ms':>14} {'launch_ms':>9} {'span_ms':>8}  reaction-end  caller events while idle", file=out)
    tot_idle = tot_span = 0.0
    n = 0
    idle_hist = []
    for p in passes:
        if p["t_done"] is None:
            continue
        span = (p["t_done"] - p["t_compose"]) / 1e3
        idle = (p["idle_before"] or 0) / 1e3
        if p["idle_before"] is not None:
            tot_idle += idle; n += 1
            idle_hist.append(idle)
        tot_span += span
        if verbose:
            evs = " ".join(f"{tag}{'D' if a > 0 and tag in ('start', 'finish', 'probe_back') else ''}:{b}" + (f"/{c}" if tag == "start" else "") for (_, tag, a, b, c) in p["drv"][-14:])
            print(f"{(p['t_compose'] - t0) / 1e3:9.2f} {p['id']:6d} {p['runs']:4d} {p['tokens']:3d} {idle:14.2f} {((p['t_fed'] or p['t_compose']) - p['t_compose']) / 1e3:9.2f} {span:8.2f}  {p.get('react_after', '-')!s:>12}  {evs}", file=out)
    wall = (ev[-1][0] - t0) / 1e3
    print(f"events {len(ev)}, wall {wall:.1f} ms, passes {len(passes)}, sum of pass spans {tot_span:.1f} ms, sum of idle before a pass {tot_idle:.1f} ms "
          f"({100 * tot_idle / max(wall, 1e-9):.1f} % of wall), reaction-window ends {dict(react)}", file=out)
    if idle_hist:
        h = np.array(idle_hist)
        print("idle before a pass, ms: mean %.2f  p50 %.2f  p90 %.2f  max %.2f;  > 1 ms: %d of %d (%.1f ms in all)" % (h.mean(), np.median(h), np.percentile(h, 90), h.max(), int((h > 1).sum()), len(h), float(h[h > 1].sum())), file=out)
    # host time of the priority (draft) context's calls by phase: start_in -> start (cells, mask) -> start_out (launch) .. finish_wait -> finish (the GPU) -> finish_out (copies, accounting); caller's own time between two calls
    ph = collections.defaultdict(list); last = {}
    for t, tag, a, b, c in ev:
        if a <= 0 or tag not in ("start_in", "start", "start_out", "finish_wait", "finish", "finish_out"):
            continue
        prev = last.get("tag")
        if prev and (prev[0], tag) in (("start_in", "start"), ("start", "start_out"), ("start_out", "finish_wait"), ("finish_wait", "finish"), ("finish", "finish_out"), ("finish_out", "start_in")):
            ph[prev[0] + " -> " + tag].append(t - prev[1])
        last["tag"] = (tag, t)
    if ph:
        print("draft calls, host microseconds by phase (median / mean / n):", {k: (int(np.median(v)), int(np.mean([x for x in v if x < 5000] or [0])), len(v)) for k, v in ph.items()}, file=out)
    widths = collections.Counter((p["runs"], p["tokens"]) for p in passes)
    print("passes by (runs, tokens):", dict(sorted(widths.items())), file=out)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="70b")
    ap.add_argument("--gens", type=int, default=3)
    ap.add_argument("--csv", default="/tmp/pipeinfer_timeline.csv")
    ap.add_argument("--analyse", default=None, help="only read an existing csv")
    ap.add_argument("--quiet", action="store_true")
    ap.add_argument("--n-predict", type=int, default=128)
    ap.add_argument("--n-draft", type=int, default=3)
    args = ap.parse_args()
    if args.analyse:
        return analyse(args.analyse, verbose=not args.quiet)
    tspec, tft, dspec, dft = B.WORKLOADS[args.workload]
    cache = os.environ.get("PIPEINFER_CACHE", "/tmp/pipeinfer_models")
    tgt_path = B.ensure_model(cache, tspec, tft, 0x5EED, 1.0, 0.25)
    dft_path = B.ensure_model(cache, dspec, dft, 0xD4AF, 1.0, 0.25, flip=0.3)
    lib = L.load()
    L.check(lib.pi_device_init(0), "pi_device_init")
    mt, md = LL.Model(tgt_path), LL.Model(dft_path)
    ct, cd = LL.Context(mt, n_ctx=1024, n_batch=512), LL.Context(md, n_ctx=1024, n_batch=512)
    cd.set_priority(1); ct.set_priority(0)
    seg_layers = int(os.environ.get("PIPEINFER_SEG_LAYERS", str(max(1, synth.SPECS[tspec][1] // 4))))
    ct.set_segments(seg_layers)
    coalesce = int(os.environ.get("PIPEINFER_COALESCE", str(B.default_coalesce(args.n_draft))))
    ct.prepare(); cd.prepare()
    ct.set_coalescing(coalesce)
    rng = np.random.default_rng(0xBEEF)
    prompts = [[int(x) for x in rng.integers(3, mt.n_vocab, 128)] for _ in range(args.gens + 1)]
    spec = dict(n_predict=args.n_predict, n_draft=args.n_draft, n_parallel=3, p_accept=0.1, p_split=0.9, ignore_eos=1, mode=0)
    import ctypes as C
    lib.pipeinfer_timeline_dump.argtypes = [C.c_char_p]; lib.pipeinfer_timeline_dump.restype = C.c_int
    for i in range(args.gens):
        ct.kv_clear(); cd.kv_clear()
        lib.pipeinfer_timeline_dump(None)
        toks, r = LL.speculative(ct, cd, prompts[i], **spec)
        n = lib.pipeinfer_timeline_dump(args.csv.encode())
        print(f"generation {i}: {r['decode_tps']:.1f} tok/s, {n} timeline events", file=sys.stderr)
    analyse(args.csv, verbose=not args.quiet)
    ct.free(); cd.free()


if __name__ == "__main__":
    main()
