#!/usr/bin/env python3
"""bench.py — accepted tokens/s of the pipelined speculative-decoding hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one speculative generation: a 128-token synthetic prompt is encoded, then 128 tokens are decoded by the
PipeInfer driver (reference recipe examples/speculative/scripts/dolphin_tinyllama_8_nodes.sh: --draft 3 -pa 0.1 -ps 0.9
-np 3 -c 1024 -n 128, greedy, --ignore-eos).  `value` = accepted tokens per second of decode time over the K timed steps
(speculative.cpp:697 definition, summed over steps), with every weight already resident in HBM.

Workload at N=1 (BASELINE.json configs[1]): Llama-2-7B Q4_K_M target + TinyLlama-1.1B Q4_K_M draft, both on one MI355X,
draft and target contexts on separate HIP streams.  Weights are shape-exact synthetic GGUFs (no network): see
pipeinfer_amd/tools/synth.py — "bigram" structure so that draft and target agree on a realistic fraction of tokens.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pipeinfer_amd import lib as L  # noqa: E402
from pipeinfer_amd import llama as LL  # noqa: E402
from pipeinfer_amd.tools import synth  # noqa: E402

HBM_PEAK_GBS = 8000.0          # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 TB/s measured copy)
WORKLOADS = {
    # name: (target spec, target ftype, draft spec, draft ftype)
    "7b": ("llama-2-7b", "q4_k_m", "tinyllama-1.1b", "q4_k_m"),
    "70b": ("llama-2-70b", "q4_k_m", "tinyllama-1.1b", "q4_k_m"),
    "tiny": ("tiny-llama-b", "q4_k_m", "tiny-llama", "q4_k_m"),
    "1b": ("tinyllama-1.1b", "q4_k_m", "tinyllama-1.1b", "q4_k_m"),          # the draft model on its own (use with --mode 2)
    "70b-q3km": ("llama-2-70b", "q3_k_m", "tinyllama-1.1b", "q4_k_m"),       # the paper's Dolphin-70B Q3_K_M setup (README.md:165)
}


def model_path(cache, spec, ftype, seed, signal, noise):
    return os.path.join(cache, f"{spec}-{ftype}-bigram-s{seed}-g{signal}-n{noise}.gguf")


def ensure_model(cache, spec, ftype, seed, signal, noise):
    os.makedirs(cache, exist_ok=True)
    path = model_path(cache, spec, ftype, seed, signal, noise)
    if not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        t0 = time.time()
        synth.write_model(tmp, spec, ftype, seed=seed, structure="bigram", signal=signal, noise=noise)
        os.replace(tmp, path)
        print(f"[bench] generated {path} ({os.path.getsize(path) / 1e9:.2f} GB) in {time.time() - t0:.1f}s", file=sys.stderr)
    return path


def cpu_baseline(tgt_path, n_tokens, n_prompt):
    """Reference CPU kernels (oracle/_ref: the reference's ggml compiled from its own sources) decoding the SAME target model
    token by token on the host cores; falls back to our C port of them (oracle/liboracle.so).  Test infrastructure: only this
    leg of bench.py touches oracle/."""
    from tests import oracle_lib as O
    # ggml's spin-waiting thread pool stops scaling long before a 256-thread host is full: use at most 32 threads and say so
    cores = min(os.cpu_count() or 1, int(os.environ.get("PIPEINFER_CPU_THREADS", "32")))
    rng = np.random.default_rng(7)
    n_vocab = None
    if O.have_ref() and O.ref() is not None:
        m = O.RefModel(tgt_path, n_ctx=max(64, n_prompt + n_tokens + 8), max_seq=2, n_threads=cores)
        kind = "reference"
    else:
        O.oracle().o_set_threads(cores)
        m = O.OracleModel(tgt_path, n_ctx=max(64, n_prompt + n_tokens + 8), max_seq=2)
        kind = "port"
    n_vocab = m.n_vocab
    prompt = [int(x) for x in rng.integers(3, n_vocab, n_prompt)]
    lg = m.decode(prompt, list(range(n_prompt)), [[0]] * n_prompt)
    cur = int(np.argmax(lg[-1]))
    t0 = time.time()
    done = 0
    for i in range(n_tokens):
        lg = m.decode([cur], [n_prompt + i], [[0]])
        cur = int(np.argmax(lg[0]))
        done += 1
        if time.time() - t0 > 25.0:          # bounded sample
            break
    dt = time.time() - t0
    n_tokens = done
    m.close()
    return {"value": n_tokens / dt, "unit": "tokens/s", "cores": cores, "kind": kind,
            "sample": f"target-only greedy decode of {n_tokens} tokens after a {n_prompt}-token prompt, same synthetic GGUF, n_threads={cores}"}


def lossless_diagnostic(ct, prompt, ref_tokens, spec_tokens, n_predict):
    """Speculative and target-only output differ: report where, and how close the call was.  Every run of the engine equals the
    reference arithmetic bit for bit (tests/trace_compare_worker.py), but that arithmetic depends on the KV-cell layout (attention
    sums run in cell order), and f16 / Q8 re-quantisation amplifies the last-bit differences; on random-weight models a different
    layout can flip a close argmax.  The logits below are those of the common prefix evaluated as one batch."""
    bad = next(i for i in range(n_predict) if ref_tokens[i] != spec_tokens[i])
    ct.kv_clear()
    seq = list(prompt) + [int(t) for t in ref_tokens[:bad]]
    lg = ct.decode(seq, list(range(len(seq))), [[0]] * len(seq))[-1]
    a, b = int(ref_tokens[bad]), int(spec_tokens[bad])
    gap = float(abs(lg[a] - lg[b]))
    return {"token_index": bad, "target_only": a, "speculative": b, "logit_target_only": round(float(lg[a]), 6), "logit_speculative": round(float(lg[b]), 6),
            "gap": round(gap, 6), "near_tie": gap < 0.25,
            "note": "every run equals the reference arithmetic bit for bit on the cells it is given; that arithmetic sums attention in KV-cell order, and a "
                    "speculative generation places tokens in other cells than a target-only one: on random-weight models a near tie can flip (DESIGN.md section 7)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("PIPEINFER_WORKLOAD", "7b"), choices=list(WORKLOADS))
    ap.add_argument("--n-predict", type=int, default=128)
    ap.add_argument("--n-prompt", type=int, default=128)
    ap.add_argument("--n-draft", type=int, default=3)
    ap.add_argument("--n-parallel", type=int, default=3)
    ap.add_argument("--p-accept", type=float, default=0.1)
    ap.add_argument("--p-split", type=float, default=0.9)
    ap.add_argument("--n-ctx", type=int, default=1024)
    ap.add_argument("--signal", type=float, default=float(os.environ.get("PIPEINFER_SIGNAL", "1.0")))
    ap.add_argument("--noise", type=float, default=float(os.environ.get("PIPEINFER_NOISE", "0.8")))
    ap.add_argument("--cache", default=os.environ.get("PIPEINFER_CACHE", "/tmp/pipeinfer_models"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-tokens", type=int, default=8)
    ap.add_argument("--mode", type=int, default=0, help="0 PipeInfer, 2 target only")
    ap.add_argument("--min-spec-iters", type=int, default=0)
    ap.add_argument("--adaptive-window", type=int, default=4, help="accepted tokens without a speculative run before the draft is paused (0: never, the reference behaviour)")
    ap.add_argument("--adaptive-pause", type=int, default=64, help="tokens the draft sits out once paused (doubles up to 16x while no speculation happens)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        from pipeinfer_amd import pipeline
        return pipeline.bench_main(args, rank, world)

    tspec, tft, dspec, dft = WORKLOADS[args.workload]
    tgt_path = ensure_model(args.cache, tspec, tft, 0x5EED, args.signal, args.noise)
    dft_path = ensure_model(args.cache, dspec, dft, 0xD4AF, args.signal, args.noise)

    lib = L.load()
    L.check(lib.pi_device_init(0), "pi_device_init")
    t0 = time.time()
    mt, md = LL.Model(tgt_path), LL.Model(dft_path)
    ct, cd = LL.Context(mt, n_ctx=args.n_ctx, n_batch=512), LL.Context(md, n_ctx=args.n_ctx, n_batch=512)
    cd.set_priority(int(os.environ.get("PIPEINFER_DRAFT_PRIORITY", "0")))
    ct.set_priority(int(os.environ.get("PIPEINFER_TARGET_PRIORITY", "0")))
    n_draft_cus = int(os.environ.get("PIPEINFER_DRAFT_CUS", "0"))
    if n_draft_cus > 0 and args.mode == 0:
        cd.set_cu_range(0, n_draft_cus); ct.set_cu_range(n_draft_cus, 256)
    ct.prepare(); cd.prepare()
    print(f"[bench] models resident in HBM after {time.time() - t0:.1f}s", file=sys.stderr)

    spec = dict(n_predict=args.n_predict, n_draft=args.n_draft, n_parallel=args.n_parallel, p_accept=args.p_accept, p_split=args.p_split,
                ignore_eos=1, mode=args.mode, min_spec_iters=args.min_spec_iters, adaptive_window=args.adaptive_window, adaptive_pause=args.adaptive_pause)
    rng = np.random.default_rng(0xBEEF)
    prompts = [[int(x) for x in rng.integers(3, mt.n_vocab, args.n_prompt)] for _ in range(args.warmup + args.steps + 1)]

    def generation(i, **over):
        ct.kv_clear(); cd.kv_clear()
        kw = dict(spec); kw.update(over)
        return LL.speculative(ct, cd, prompts[i], **kw)

    # lossless check (also the first warm-up): target-only greedy == speculative output on the same prompt
    ref_tokens, _ = generation(0, mode=2)
    spec_tokens, _ = generation(0)
    lossless = ref_tokens[:args.n_predict] == spec_tokens[:args.n_predict]
    first_diff = None if lossless else lossless_diagnostic(ct, prompts[0], ref_tokens, spec_tokens, args.n_predict)
    for i in range(args.warmup):
        generation(1 + i)

    ct.reset_stats(); cd.reset_stats()
    L.check(lib.pi_device_sync())
    t_start = time.time()
    results = []
    for i in range(args.steps):
        _, r = generation(1 + args.warmup + i)
        results.append(r)
    L.check(lib.pi_device_sync())
    t_total = time.time() - t_start
    st_t, st_d = ct.stats(), cd.stats()                 # per-call GPU time and bytes of the timed steps only

    # on one GPU the driver (faithfully) finds the target back before the draft is ahead and launches no speculative run, so the timed
    # steps do not exercise draft trees, verification, acceptance and cancellation at the real model sizes: one extra, untimed
    # generation forces a speculation iteration per token (min_spec_iters, never part of `value`) and is checked for losslessness
    forced = None
    if args.mode == 0:
        f_tokens, f_res = generation(0, min_spec_iters=1, adaptive_window=0, adaptive_pause=0)
        forced = {"tokens_per_s": round(f_res["n_predict"] / f_res["t_dec_s"], 2), "spec_runs": f_res["n_runs_spec"], "canceled_runs": f_res["n_runs_canceled"],
                  "drafted_tokens": f_res["n_drafted_tokens"], "accepted_draft_tokens": f_res["n_accept_clean"],
                  "acceptance_rate": round(f_res["n_accept_clean"] / max(1, f_res["n_drafted_tokens"]), 4),
                  "lossless_vs_target_only": bool(ref_tokens[:args.n_predict] == f_tokens[:args.n_predict])}
        if not forced["lossless_vs_target_only"]:
            forced["first_difference"] = lossless_diagnostic(ct, prompts[0], ref_tokens, f_tokens, args.n_predict)

    n_tok = sum(r["n_predict"] for r in results)
    t_dec = sum(r["t_dec_s"] for r in results)
    value = n_tok / t_dec
    drafted = sum(r["n_drafted_tokens"] for r in results)
    acc_clean = sum(r["n_accept_clean"] for r in results)
    acc_ref = 100.0 * sum(r["n_accept"] for r in results) / max(1, sum(r["n_drafted"] for r in results))

    # roofline of the dominant kernel: the Q4_K single-token mat-vec (k_mmvq_kq<Q4_K, NT=1>), every 4th launch bracketed by HIP
    # events on its own stream.  Events recorded inside a captured graph cannot be read back on ROCm 7.2, so the sampling
    # pass replays the same decode step kernel by kernel (graphs off) right after the timed region, on the same context.
    ct.set_graphs(0)
    lib.pi_kernel_timing(1, 5)          # 5 is coprime to the 3- and 4-launch periods of a layer: every launch shape gets sampled
    generation(1 + args.warmup, mode=2, n_predict=min(32, args.n_predict))
    L.check(lib.pi_device_sync())
    ct.set_graphs(1)
    avg_us, avg_bytes, n_s, n_l = C.c_double(), C.c_double(), C.c_uint64(), C.c_uint64()
    lib.pi_kernel_timing_report(12, 1, C.byref(avg_us), C.byref(avg_bytes), C.byref(n_s), C.byref(n_l))
    achieved = (avg_bytes.value / (avg_us.value * 1e-6) / 1e9) if avg_us.value > 0 else 0.0
    # HBM traffic of the same kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction): bench.py cannot
    # run the profiler on itself, so the number comes from profiles/ (same workload, same launches) and says so
    traffic, traffic_src = None, None
    try:
        pmc = json.load(open(os.path.join(ROOT, "profiles", "r01_pmc_fetch_size.json")))
        if args.workload == "7b":
            traffic, traffic_src = pmc["dominant_kernel"]["avg_fetch_bytes_per_launch"], "profiles/r01_pmc_fetch_size.json"
    except Exception:
        pass
    roofline = {"bound": "hbm", "kernel": "k_mmvq_kq<Q4_K,NT=1>", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
                "avg_launch_us": round(avg_us.value, 3), "alg_bytes_per_launch": round(avg_bytes.value), "launches": int(n_l.value), "sampled": int(n_s.value),
                # single-token decode calls only (prompt batches excluded): GPU time between the call's first and last event
                "target_decode_call": {"calls": st_t["n_calls_1"], "gpu_ms_per_call": round(st_t["gpu_ms_1"] / max(1, st_t["n_calls_1"]), 4),
                                       "alg_GB_per_call": round(st_t["alg_bytes_1"] / max(1, st_t["n_calls_1"]) / 1e9, 4),
                                       "GBps": round(st_t["alg_bytes_1"] / max(1e-9, st_t["gpu_ms_1"] * 1e-3) / 1e9, 1)},
                "draft_decode_call": {"calls": st_d["n_calls_1"], "gpu_ms_per_call": round(st_d["gpu_ms_1"] / max(1, st_d["n_calls_1"]), 4),
                                      "GBps": round(st_d["alg_bytes_1"] / max(1e-9, st_d["gpu_ms_1"] * 1e-3) / 1e9, 1)}}
    lib.pi_kernel_timing(0, 8)

    out = {"metric": "accepted tokens/sec + acceptance rate, 70B Q4_K target / 1.1B draft", "value": round(value, 2), "unit": "tokens/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * t_total / args.steps, 2), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "int8 x int4/6 dot, fp32 accumulate (f16 KV)", "data": "synthetic",
           "config": {"workload": f"{tspec} {tft} target + {dspec} {dft} draft on 1xMI355X, two HIP streams (BASELINE configs[1])",
                      "n_prompt": args.n_prompt, "n_predict": args.n_predict, "n_draft": args.n_draft, "n_parallel": args.n_parallel,
                      "p_accept": args.p_accept, "p_split": args.p_split, "n_ctx": args.n_ctx, "sampling": "greedy", "synthetic_structure": f"bigram signal={args.signal} noise={args.noise}",
                      "adaptive_draft": {"window": args.adaptive_window, "pause": args.adaptive_pause}},
           "acceptance": {"accepted_draft_tokens": acc_clean, "drafted_tokens": drafted, "rate": round(acc_clean / max(1, drafted), 4),
                          "reference_formula_pct": round(acc_ref, 2), "spec_runs": sum(r["n_runs_spec"] for r in results),
                          "canceled_runs": sum(r["n_runs_canceled"] for r in results),
                          "draft_paused_tokens": sum(r["n_draft_paused"] for r in results), "draft_catchups": sum(r["n_draft_catchups"] for r in results)},
           "lossless_vs_target_only": bool(lossless), "first_difference": first_diff, "forced_speculation_check": forced, "ttft_s": round(float(np.mean([r["ttft_s"] for r in results])), 5),
           "itl_s": round(float(np.mean([r["itl_s"] for r in results])), 6), "encode_tps": round(float(np.mean([r["encode_tps"] for r in results])), 1),
           "roofline": roofline}
    if not args.no_cpu_baseline:
        for c in (ct, cd):
            c.free()
        try:
            out["cpu_baseline"] = cpu_baseline(tgt_path, args.cpu_tokens, 16)
        except Exception as e:  # the GPU result stands on its own; say why the CPU leg is missing
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": repr(e)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
