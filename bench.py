#!/usr/bin/env python3
"""bench.py — accepted tokens/s of the pipelined speculative-decoding hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

A "step" is one speculative generation: a 128-token synthetic prompt is encoded, then 128 tokens are decoded by the
PipeInfer driver (reference recipe examples/speculative/scripts/dolphin_tinyllama_8_nodes.sh: --draft 3 -pa 0.1 -ps 0.9
-np 3 -c 1024 -n 128, greedy, --ignore-eos).  `value` = accepted tokens per second of decode time over the K timed steps
(speculative.cpp:697 definition, summed over steps), with every weight already resident in HBM.

Workload at N=1: the configuration BASELINE.json's metric is quoted on — Llama-2-70B Q4_K_M target + TinyLlama-1.1B Q4_K_M
draft; both fit one MI355X (41 GB + 0.7 GB), draft and target contexts on separate HIP streams.  `value` is the
reference-faithful PipeInfer driver (mode 0: asynchronous continuous speculation, no additions).  The same prompts are also
run target-only (mode 2) and with the reference's synchronous tree speculation (mode 1, examples/speculative_orig): those rates
are reported under `other_schedules`, never as `value`.  Weights are shape-exact synthetic GGUFs (no network): see
pipeinfer_amd/tools/synth.py — "bigram" structure, the draft is wrong about a fixed fraction of the vocabulary so that the
acceptance rate is a controlled quantity with large logit margins in both models.
"""
import argparse
import ctypes as C
import hashlib
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
METRIC = "accepted tokens/sec + acceptance rate, 70B Q4_K target / 1.1B draft"
WORKLOADS = {
    # name: (target spec, target ftype, draft spec, draft ftype)
    "70b": ("llama-2-70b", "q4_k_m", "tinyllama-1.1b", "q4_k_m"),             # the configuration the metric is quoted on
    "7b": ("llama-2-7b", "q4_k_m", "tinyllama-1.1b", "q4_k_m"),               # BASELINE configs[1]
    "tiny": ("tiny-llama-b", "q4_k_m", "tiny-llama", "q4_k_m"),
    "1b": ("tinyllama-1.1b", "q4_k_m", "tinyllama-1.1b", "q4_k_m"),          # the draft model on its own (use with --mode 2)
    "70b-q3km": ("llama-2-70b", "q3_k_m", "tinyllama-1.1b", "q4_k_m"),       # the paper's Dolphin-70B Q3_K_M setup (README.md:165)
    "falcon40b": ("falcon-40b", "q5_k_m", "falcon-7b", "q4_0"),               # BASELINE configs[3] (two-norm Falcon block, GQA 128/8, 65024-token BPE vocabulary)
    "goliath": ("goliath-120b", "q4_k_m", "llama-2-7b", "q4_k_m"),            # BASELINE configs[4] (137 layers of 70B shape; Xwin-7B = Llama-2-7B shape; run with --n-draft 8)
}
TYPE_NAMES = {10: "Q2_K", 11: "Q3_K", 12: "Q4_K", 13: "Q5_K", 14: "Q6_K", 2: "Q4_0", 8: "Q8_0"}


def model_path(cache, spec, ftype, seed, signal, noise, flip=0.0, layers=None):
    tag = f"-f{flip}" if flip else ""
    tag += f"-l{layers[0]}+{layers[1]}" if layers else ""
    return os.path.join(cache, f"{spec}-{ftype}-bigram-s{seed}-g{signal}-n{noise}{tag}.gguf")


def ensure_model(cache, spec, ftype, seed, signal, noise, flip=0.0, layers=None):
    os.makedirs(cache, exist_ok=True)
    path = model_path(cache, spec, ftype, seed, signal, noise, flip, layers)
    if not os.path.exists(path):
        tmp = path + f".tmp{os.getpid()}"
        t0 = time.time()
        synth.write_model(tmp, spec, ftype, seed=seed, structure="bigram", signal=signal, noise=noise, pi_flip=flip, layers=layers)
        os.replace(tmp, path)
        print(f"[bench] generated {path} ({os.path.getsize(path) / 1e9:.2f} GB) in {time.time() - t0:.1f}s", file=sys.stderr)
    return path


def kernel_source_sha():
    """sha of the kernel sources: profiles/ files carry the sha of the code they were measured on, and are only quoted when it matches"""
    h = hashlib.sha256()
    d = os.path.join(ROOT, "pipeinfer_amd", "csrc", "kernels")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            h.update(f.encode()); h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


def committed_traffic(workload, kernel_key):
    """HBM traffic of the dominant kernel from the committed PMC pass (rocprofv3 --pmc FETCH_SIZE, x2 gfx950 correction): bench.py cannot
    run the profiler on itself.  The file is only quoted if it was measured on THIS kernel source (sha) and workload."""
    import glob
    sha = kernel_source_sha()
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r*_pmc_fetch_size_{workload}.json")), reverse=True)      # newest round first
    if not files:
        return None, f"no committed PMC pass for workload {workload} (profiles/r*_pmc_fetch_size_{workload}.json)"
    pmc, name, seen = None, None, []
    for f in files:
        try:
            cand = json.load(open(f))
        except Exception:
            continue
        seen.append(f"{os.path.basename(f)}: {cand.get('kernel_src_sha16')}")
        if cand.get("kernel_src_sha16") == sha:
            pmc, name = cand, os.path.basename(f)
            break
    if pmc is None:
        return None, f"the committed PMC passes were measured on other kernel sources ({'; '.join(seen)}), these are {sha}: not quoted"
    k = pmc.get("families", {}).get(kernel_key)
    if not k:
        return None, f"profiles/{name} has no entry for {kernel_key}"
    return k, f"profiles/{name}"


def _time_cpu_model(path, n_tokens, n_prompt, cores, budget_s):
    from tests import oracle_lib as O
    if O.have_ref() and O.ref_kv() is not None:
        # the reference's own llama.cpp — loader, KV cache, graph builder, ggml thread pool — compiled from its sources (oracle/ref_kv_harness.cpp)
        m = O.RefLlama(path, n_ctx=max(64, n_prompt + 3 * n_tokens + 8), max_seq=2, n_threads=cores)
        kind = "reference"
    elif O.have_ref() and O.ref() is not None:
        m = O.RefModel(path, n_ctx=max(64, n_prompt + 3 * n_tokens + 8), max_seq=2, n_threads=cores)
        kind = "reference"
    else:
        O.oracle().o_set_threads(cores)
        m = O.OracleModel(path, n_ctx=max(64, n_prompt + 3 * n_tokens + 8), max_seq=2)
        kind = "port"
    rng = np.random.default_rng(7)
    prompt = [int(x) for x in rng.integers(3, m.n_vocab, n_prompt)]
    lg = m.decode(prompt, list(range(n_prompt)), [[0]] * n_prompt)
    cur = int(np.argmax(lg[-1]))
    # three passes of n_tokens; the fastest counts and all three are reported: ggml's spin-waiting thread pool is sensitive to whatever else the host is doing at that moment
    reps, pos = [], n_prompt
    for rep in range(3):
        t0 = time.time()
        done = 0
        for i in range(n_tokens):
            lg = m.decode([cur], [pos], [[0]])
            cur = int(np.argmax(lg[0]))
            pos += 1
            done += 1
            if time.time() - t0 > budget_s / 3:  # bounded sample
                break
        reps.append(((time.time() - t0) / done, done))
    m.close()
    best, done_best = min(reps)
    return best, done_best, kind, [round(1e3 * r[0], 2) for r in reps]


def cpu_baseline(tspec, tft, tgt_path, cache, signal, noise, n_tokens=8, n_prompt=4):
    """The reference's CPU path (oracle/_ref: its llama.cpp graph builder on its ggml kernels, compiled from its own sources) decoding the target model token by
    token on the host cores; falls back to our C port of them (oracle/liboracle.so).  Test infrastructure: only this leg of bench.py
    touches oracle/.  Models that fit the time budget are decoded whole; the 70B target is timed on two depth-truncated copies (8 and 16
    layers with the tensor types of layers 8.. of the 80 — half of them "more bits" layers, like the whole model) and extrapolated
    linearly in the layer count (decode cost is additive per layer: BASELINE.md section 2, method B1)."""
    # ggml's spin-waiting thread pool stops scaling long before a 256-thread host is full: use at most 32 threads and say so
    cores = min(os.cpu_count() or 1, int(os.environ.get("PIPEINFER_CPU_THREADS", "32")))
    n_layer = synth.SPECS[tspec][1]
    if n_layer <= 32:
        spt, done, kind, reps = _time_cpu_model(tgt_path, n_tokens, 16, cores, 25.0)
        return {"value": 1.0 / spt, "unit": "tokens/s", "cores": cores, "threads": cores, "host_cores": os.cpu_count(), "kind": kind, "ms_per_token_repetitions": reps,
                "sample": f"target-only greedy decode of {done} tokens after a 16-token prompt, same synthetic GGUF, n_threads={cores}; fastest of three repetitions"}
    # two copies that are both far beyond any last-level cache (4.5 GB and 8.6 GB of weights per token): round 4's 4-layer copy (2.4 GB) decoded at half the per-layer cost
    # of the 8-layer one, and the clamp `head = max(0, ...)` hid it
    p8 = ensure_model(cache, tspec, tft, 0x5EED, signal, noise, layers=(8, 8))
    p16 = ensure_model(cache, tspec, tft, 0x5EED, signal, noise, layers=(8, 16))
    t8, d8, kind, reps8 = _time_cpu_model(p8, n_tokens, n_prompt, cores, 10.0)
    t16, d16, _, reps16 = _time_cpu_model(p16, n_tokens, n_prompt, cores, 18.0)
    return cpu_extrapolation(n_layer, t8, t16, reps8, reps16, d8, d16, kind, cores)


def cpu_extrapolation(n_layer, t8, t16, reps8, reps16, d8, d16, kind, cores):
    """seconds per token of the full model from the 8- and the 16-layer copy, two ways, both reported and nothing clamped:
      difference  per layer = (t16 - t8) / 8, head (embedding + lm_head + fixed cost) = t8 - 8 per layer          -> head + n_layer per layer
      ratio       per layer = t16 / 16 (the head's share charged to the layers: an upper bound)                    -> n_layer t16 / 16
    `value` is the difference estimate (the head is counted once, as in the full model); `value_range` spans both estimates over the repetitions."""
    per_layer = (t16 - t8) / 8.0
    head = t8 - 8.0 * per_layer                                             # may come out negative when the host's timings are noisy: shown as it is
    spt_diff = head + n_layer * per_layer
    spt_ratio = n_layer * t16 / 16.0
    spt = spt_diff if spt_diff > 0 else spt_ratio
    cands = [spt_diff, spt_ratio, (max(reps16) - min(reps8)) / 8e3 * n_layer + head, (min(reps16) - max(reps8)) / 8e3 * n_layer + head, n_layer * max(reps16) / 16e3, n_layer * min(reps16) / 16e3]
    cands = [c for c in cands if c > 0]
    return {"value": 1.0 / spt, "unit": "tokens/s", "cores": cores, "threads": cores, "host_cores": os.cpu_count(), "kind": kind,
            "ms_per_token_8_layers": reps8, "ms_per_token_16_layers": reps16,
            "ms_per_layer": {"difference": round(per_layer * 1e3, 3), "t16_over_16": round(t16 / 16.0 * 1e3, 3), "t8_over_8": round(t8 / 8.0 * 1e3, 3)}, "ms_head_difference_style": round(head * 1e3, 3),
            "ms_per_token_full_model": {"difference": round(spt_diff * 1e3, 1), "t16_over_16": round(spt_ratio * 1e3, 1)},
            "value_range": [round(1.0 / max(cands), 3), round(1.0 / min(cands), 3)],
            "sample_short": f"reference kernels (oracle/_ref), {cores} of {os.cpu_count()} host cores: target-only greedy decode of {d8}+{d16} tokens on 8- and 16-layer copies of the {n_layer}-layer "
                            f"model, per layer (t16-t8)/8 = {per_layer * 1e3:.2f} ms, head {head * 1e3:.2f} ms, extrapolated to {spt * 1e3:.0f} ms per token",
            "sample": f"target-only greedy decode, reference kernels on {cores} of the host's {os.cpu_count()} cores: {d8}+{d16} tokens (fastest of three repetitions each) on 8- and 16-layer copies of the "
                      f"{n_layer}-layer model (tensor types of layers 8..15 / 8..23, embedding + lm_head included; 4.5 and 8.6 GB of weights per token, both beyond any cache) -> difference style "
                      f"{per_layer * 1e3:.2f} ms per layer + {head * 1e3:.2f} ms head = {spt_diff * 1e3:.0f} ms per token; ratio style t16/16 = {t16 / 16.0 * 1e3:.2f} ms per layer = {spt_ratio * 1e3:.0f} ms per token"}


def lossless_diagnostic(ct, prompt, ref_tokens, spec_tokens, n_predict):
    """Speculative and target-only output differ: report where, and how close the call was.  Every run of the engine equals the
    reference arithmetic bit for bit (tests/test_gpu_engine.py: per-run oracle comparison at these shapes), but that arithmetic depends
    on the KV-cell layout (attention sums run in cell order), and f16 / Q8 re-quantisation amplifies the last-bit differences; on
    random-weight models a different layout can flip a close argmax.  The logits below are those of the common prefix evaluated as one batch."""
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


def summarize(results):
    n_tok = sum(r["n_predict"] for r in results)
    t_dec = sum(r["t_dec_s"] for r in results)
    drafted = sum(r["n_drafted_tokens"] for r in results)
    acc_clean = sum(r["n_accept_clean"] for r in results)
    acc_ref = 100.0 * sum(r["n_accept"] for r in results) / max(1, sum(r["n_drafted"] for r in results))
    spec_runs = sum(r["n_runs_spec"] for r in results)
    return {"tokens_per_s": round(n_tok / t_dec, 2),
            "accepted_draft_tokens": acc_clean, "drafted_tokens": drafted,
            "rate": round(acc_clean / drafted, 4) if spec_runs > 0 and drafted > 0 else None,          # undefined without speculation
            "reference_formula_pct": round(acc_ref, 2) if spec_runs > 0 else None,
            "spec_runs": spec_runs, "nonspec_runs": sum(r["n_runs_nonspec"] for r in results), "canceled_runs": sum(r["n_runs_canceled"] for r in results),
            "draft_paused_tokens": sum(r["n_draft_paused"] for r in results), "draft_catchups": sum(r["n_draft_catchups"] for r in results)}


FAMILIES = {0: "matvec", 1: "matrix_cores", 2: "norm_quant", 3: "quantize", 4: "act_quant", 5: "attention", 6: "other"}
FAMILY_KERNELS = {"matvec_nt1": "k_mmvq_kq / k_mmvq_kq2 / k_mmvq_q0r (VALU mat-vec, one token)", "matvec_multi": "k_mmvq_* (VALU mat-vec, 2..8 tokens)",
                  "matrix_cores": "k_mmq_kq / k_mmq_g2 (matrix-core mat-mat, 3..32 tokens per pass)", "norm_quant": "k_norm_quant", "quantize": "k_quantize_q8_K / _q8_0",
                  "act_quant": "k_act_quant", "attention": "k_attention", "other": "k_get_rows, k_rope_kv, k_topk1/2, k_norm, k_add, ..."}


LINE_BUDGET = 4000             # the ONE stdout line stays below this many characters: the driver keeps an ~8.5 KB tail of the output and must be able to parse it


def _short(text, n=240):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(out, detail_path=None):
    """The ONE JSON line of the bench contract, kept small (< LINE_BUDGET characters) whatever the size of the tables behind it: headline keys, `roofline` of the dominant
    kernel family with `by_family` reduced to {share, frac}, the whole-call figures, `cpu_baseline` with a one-sentence sample.  Everything else — `per_instance`, the full
    `by_family`, cancel / segment statistics, per-stage tables of the N > 1 job — goes to the detail file (and to stderr)."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "acceptance",
           "lossless_vs_target_only", "ttft_s", "itl_s", "encode_tps", "transport", "hop", "devices_shared")
    line = {k: out[k] for k in top if k in out}
    if isinstance(line.get("config"), dict):
        cfg = dict(line["config"])
        if not any((cfg.get("driver_additions") or {}).values()):           # all zero (the reference-faithful driver): said once, in the detail file
            cfg.pop("driver_additions", None)
        for k, v in list(cfg.items()):
            if isinstance(v, str):
                cfg[k] = _short(v, 160)
        line["config"] = cfg
    if isinstance(line.get("acceptance"), dict):
        keep = ("accepted_draft_tokens", "drafted_tokens", "rate", "reference_formula_pct", "spec_runs", "nonspec_runs", "canceled_runs", "runs_skipped_by_stages")
        line["acceptance"] = {k: v for k, v in line["acceptance"].items() if k in keep}
    if out.get("first_difference"):
        fd = out["first_difference"]
        line["first_difference"] = {k: fd.get(k) for k in ("token_index", "gap", "near_tie")}
    if "other_schedules" in out:
        line["other_schedules"] = {name: {"tokens_per_s": s_.get("tokens_per_s"), "lossless_vs_target_only": s_.get("lossless_vs_target_only")} for name, s_ in out["other_schedules"].items()}
    r = out.get("roofline") or {}
    keys = ("bound", "kernel", "family", "achieved", "peak", "unit", "frac", "traffic", "traffic_fetch_over_alg", "avg_launch_us", "alg_bytes_per_launch", "launches",
            "aborted_launches", "share_of_gpu_time", "kernel_src_sha16", "target_decode_call", "target_all_calls", "draft_decode_call", "target_passes")
    rl = {k: r[k] for k in keys if k in r}
    if isinstance(rl.get("kernel"), str):
        rl["kernel"] = _short(rl["kernel"], 80)
    if "by_family" in r:
        rl["by_family"] = {name: {"share": f.get("share_of_gpu_time"), "frac": f.get("frac")} for name, f in r["by_family"].items()}
    if "stages" in r:                                                        # N > 1: one short row per stage
        rl["stages"] = [{"rank": s_.get("rank"), "layers": s_.get("layers"), "gpu_ms_per_call": s_.get("gpu_ms_per_call"), "frac": round(s_.get("GBps", 0.0) / HBM_PEAK_GBS, 4),
                         "hop_store": s_.get("hop_store"), "ring": s_.get("ring_alloc")} for s_ in r["stages"]]
    line["roofline"] = rl
    if "cpu_baseline" in out:
        c = out["cpu_baseline"]
        line["cpu_baseline"] = {k: c[k] for k in ("value", "unit", "cores", "kind", "host_cores", "value_range") if k in c}
        if isinstance(line["cpu_baseline"].get("value"), float):
            line["cpu_baseline"]["value"] = round(line["cpu_baseline"]["value"], 4)
        line["cpu_baseline"]["sample"] = _short(c.get("sample_short") or c.get("sample", ""), 260)
    if detail_path:
        line["detail"] = detail_path
    text = json.dumps(line)
    if len(text) > LINE_BUDGET:                                              # never grow past the budget: shed the optional parts, least important first
        for victim in ("first_difference", "other_schedules", "acceptance"):
            line.pop(victim, None)
            text = json.dumps(line)
            if len(text) <= LINE_BUDGET:
                break
    if len(text) > LINE_BUDGET:
        for victim in ("by_family", "stages", "target_passes", "target_all_calls", "draft_decode_call"):
            line["roofline"].pop(victim, None)
            text = json.dumps(line)
            if len(text) <= LINE_BUDGET:
                break
    return text


def emit(out, detail_path):
    """full result -> detail file (+ stderr, BEFORE the line: whoever keeps a tail of the combined output still ends on the line); compact line -> stdout, last"""
    full = json.dumps(out)
    written = None
    if detail_path:
        try:
            os.makedirs(os.path.dirname(os.path.abspath(detail_path)), exist_ok=True)
            with open(detail_path, "w") as f:
                f.write(full + "\n")
            written = detail_path
        except OSError as e:
            print(f"[bench] could not write the detail file {detail_path}: {e}", file=sys.stderr)
    print("[bench detail] " + full, file=sys.stderr)
    sys.stderr.flush()
    print(compact_line(out, written))
    sys.stdout.flush()


class KtRow(C.Structure):          # struct pi_kt_row (include/pipeinfer_ops.h)
    _fields_ = [("family", C.c_int32), ("type", C.c_int32), ("n_tok", C.c_int32), ("sampled_dropped", C.c_int32), ("shape_bytes", C.c_uint64), ("launches", C.c_uint64),
                ("bytes_total", C.c_double), ("sampled", C.c_uint64), ("avg_us", C.c_double), ("avg_sampled_bytes", C.c_double)]


def _rows(buf, n):
    return [{"family": r.family, "type": r.type, "n_tok": r.n_tok, "shape_bytes": int(r.shape_bytes), "launches": int(r.launches), "bytes_total": float(r.bytes_total),
             "sampled": int(r.sampled), "sampled_dropped": int(r.sampled_dropped), "avg_us": float(r.avg_us), "avg_sampled_bytes": float(r.avg_sampled_bytes)} for r in buf[:n]]


def timing_rows(lib):
    """eager launches of the open timing window: counts, algorithmic bytes and the sampled durations (call after a device sync)"""
    buf = (KtRow * 4096)()
    return _rows(buf, lib.pi_kernel_timing_rows(buf, 4096))


def lifetime_rows(lib):
    """every eager launch of this process so far (counts and algorithmic bytes): under PI_NO_GRAPHS=1 that is every launch — the population a rocprofv3 --pmc pass of the process sees"""
    buf = (KtRow * 4096)()
    return _rows(buf, lib.pi_kernel_lifetime_rows(buf, 4096))


def launch_rows(lib, ctx):
    """launches of a context's captured graphs since its reset_stats(): graph contents x replays (counts and algorithmic bytes; no durations)"""
    lib.pipeinfer_get_launch_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.pipeinfer_get_launch_rows.restype = C.c_int
    buf = (KtRow * 4096)()
    return _rows(buf, lib.pipeinfer_get_launch_rows(ctx.h, buf, 4096))


def aborted_launches(lib, ctx):
    """launches of the context since its reset_stats() that found their run cancelled and returned on entry (device-side counters beside the abort word), per family index"""
    lib.pipeinfer_get_aborted_launches.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    lib.pipeinfer_get_aborted_launches.restype = C.c_int
    buf = (C.c_uint64 * 8)()
    n = lib.pipeinfer_get_aborted_launches(ctx.h, buf, 8)
    return {i: int(buf[i]) for i in range(max(0, n))}


def family_of(r):
    f = FAMILIES.get(r["family"], "other")
    return ("matvec_nt1" if r["n_tok"] == 1 else "matvec_multi") if f == "matvec" else f


def roofline_tables(counts, samples, aborted=None):
    """counts: the launches of the TIMED region (graph contents x replays + eager launches); samples: per-instance durations of the graphs-off replay of the same schedule
    (launches that did the work: the sampler leaves out the ones that found their run aborted, `sampled_dropped`).
    An instance = (family, weight type, tokens of the launch, launch shape).  Estimated GPU time of an instance in the timed region = its launches there x the replay's
    average duration of the same instance (the nearest width of the same launch shape when that exact width did not occur in the replay).
    aborted: {family index: launches of the timed region that returned on entry because their run had been cancelled} — kept apart: they are taken out of the family's
    launches (pro rata over its instances: the device counts them per family) instead of being priced with full bytes and a full duration."""
    aborted = aborted or {}
    fam_launches = {}
    for c in counts:
        fam_launches[c["family"]] = fam_launches.get(c["family"], 0) + c["launches"]
    live = {f: max(0.0, 1.0 - aborted.get(f, 0) / n) if n else 1.0 for f, n in fam_launches.items()}      # fraction of a family's launches that did the work
    by_key = {}
    for r in samples:
        if r["sampled"]:
            by_key[(r["family"], r["type"], r["n_tok"], r["shape_bytes"])] = r
    inst, fam = [], {}
    for c in counts:
        if not c["launches"]:
            continue
        k = (c["family"], c["type"], c["n_tok"], c["shape_bytes"])
        smp = by_key.get(k)
        how = "same instance"
        if smp is None:                                  # nearest sampled width of the same kernel family / type / launch shape
            near = [v for kk, v in by_key.items() if kk[0] == k[0] and kk[1] == k[1] and kk[3] == k[3]]
            if not near and k[0] >= 2:                  # families without a launch shape: nearest width of the family
                near = [v for kk, v in by_key.items() if kk[0] == k[0]]
            if near:
                smp = min(near, key=lambda v: abs(v["n_tok"] - c["n_tok"])); how = f"nearest sampled width ({smp['n_tok']} tokens)"
        avg_us = smp["avg_us"] if smp else None
        # attention: a captured graph is launched with n_kv = n_ctx and reads the real n_kv on the device: price it with the bytes of the sampled (eager) launches
        bytes_per = (smp["avg_sampled_bytes"] if (smp and c["family"] == 5) else c["bytes_total"] / c["launches"])
        n_live = c["launches"] * live.get(c["family"], 1.0)                  # launches that did the work
        row = {"family": family_of(c), "type": TYPE_NAMES.get(c["type"], "-") if c["family"] <= 1 else "-", "n_tok": c["n_tok"], "launch_MB": round(c["shape_bytes"] / 1e6, 2) if c["shape_bytes"] else None,
               "launches": c["launches"], "alg_bytes_per_launch": round(bytes_per), "avg_us": round(avg_us, 3) if avg_us else None, "duration_from": how if smp else "not sampled",
               "samples": smp["sampled"] if smp else 0, "samples_left_out_as_aborted": smp.get("sampled_dropped", 0) if smp else 0,
               "GBps": round(bytes_per / (avg_us * 1e-6) / 1e9, 1) if avg_us else None, "est_gpu_ms": round(n_live * avg_us * 1e-3, 3) if avg_us else None}
        inst.append(row)
        f = fam.setdefault(row["family"], {"launches": 0, "launches_priced": 0, "alg_bytes": 0.0, "est_gpu_ms": 0.0, "aborted": 0, "live": 0.0})
        f["launches"] += c["launches"]
        f["aborted"] = min(aborted.get(c["family"], 0), fam_launches.get(c["family"], 0))
        if avg_us:
            f["launches_priced"] += c["launches"]; f["live"] += n_live; f["alg_bytes"] += bytes_per * n_live; f["est_gpu_ms"] += n_live * avg_us * 1e-3
    total_ms = sum(f["est_gpu_ms"] for f in fam.values()) or 1.0
    out = {}
    for name, f in sorted(fam.items(), key=lambda kv: -kv[1]["est_gpu_ms"]):
        gbps = f["alg_bytes"] / max(1e-12, f["est_gpu_ms"] * 1e-3) / 1e9
        # (the one-token and the multi-token VALU mat-vecs share a device counter: `aborted_launches` of either is the kernel family's)
        out[name] = {"kernels": FAMILY_KERNELS.get(name, name), "launches": f["launches"], "launches_priced": f["launches_priced"], "aborted_launches": f["aborted"], "est_gpu_ms": round(f["est_gpu_ms"], 2), "alg_GB": round(f["alg_bytes"] / 1e9, 3),
                     "share_of_gpu_time": round(f["est_gpu_ms"] / total_ms, 4), "alg_bytes_per_launch": round(f["alg_bytes"] / max(1e-9, f["live"])),
                     "avg_launch_us": round(1e3 * f["est_gpu_ms"] / max(1e-9, f["live"]), 3), "achieved_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBS, 4)}
    inst.sort(key=lambda r: -(r["est_gpu_ms"] or 0.0))
    return out, inst


def merge_family_tables(tables):
    """by_family tables of several stages (N > 1: one per stage, each priced with its own GPU's durations) -> one table: bytes and GPU time add up, the rate is their ratio"""
    acc = {}
    for t in tables:
        for name, f in t.items():
            a = acc.setdefault(name, {"kernels": f["kernels"], "launches": 0, "launches_priced": 0, "aborted_launches": 0, "est_gpu_ms": 0.0, "alg_GB": 0.0, "live": 0.0})
            a["launches"] += f["launches"]; a["launches_priced"] += f["launches_priced"]; a["aborted_launches"] += f["aborted_launches"]
            a["est_gpu_ms"] += f["est_gpu_ms"]; a["alg_GB"] += f["alg_GB"]
            a["live"] += 1e3 * f["est_gpu_ms"] / f["avg_launch_us"] if f["avg_launch_us"] else 0.0
    total = sum(a["est_gpu_ms"] for a in acc.values()) or 1.0
    out = {}
    for name, a in sorted(acc.items(), key=lambda kv: -kv[1]["est_gpu_ms"]):
        gbps = a["alg_GB"] / max(1e-12, a["est_gpu_ms"] * 1e-3)
        out[name] = {"kernels": a["kernels"], "launches": a["launches"], "launches_priced": a["launches_priced"], "aborted_launches": a["aborted_launches"], "est_gpu_ms": round(a["est_gpu_ms"], 2),
                     "alg_GB": round(a["alg_GB"], 3), "share_of_gpu_time": round(a["est_gpu_ms"] / total, 4), "alg_bytes_per_launch": round(1e9 * a["alg_GB"] / max(1e-9, a["live"])),
                     "avg_launch_us": round(1e3 * a["est_gpu_ms"] / max(1e-9, a["live"]), 3), "achieved_GBps": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBS, 4)}
    return out


def default_coalesce(n_draft):
    """widest pass the target's feeder composes: 8 tokens - two speculative runs at the headline's n_draft 3 (or one run of 8 drafted tokens), and the widest pass of the
    two-row-groups-per-wave matrix-core kernel.  Round 4, results written by a kernel (tok/s at 8 | 16): 70B 102-105 | 92.8, 7B 257 | 257, Falcon-40B 75.8 | 75.6,
    Goliath (n_draft 8) 55.7 | 46.5 - a wider pass mostly carries runs that continuous speculation cancels while they compute."""
    return 8


def launch_ranks(n):
    """`python bench.py --gpus N` without a launcher around it: start N ranks (one per GPU) as CHILD processes through torch.distributed.run,
    wait for them, print rank 0's single JSON line and return non-zero if any rank failed.  Nothing here initialises the GPU."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")            # dmabuf IPC: the peer rings between the stage processes need it
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    line = None
    for ln in proc.stdout:                                       # rank 0's JSON line is the only thing that belongs on stdout
        t = ln.strip()
        if t.startswith("{") and '"metric"' in t:
            line = t
        elif t:
            print(t, file=sys.stderr)
    rc = proc.wait()
    if line is not None:
        print(line)
    if rc != 0 or line is None:
        print(f"[bench] {n}-rank job failed: exit code {rc}" + ("" if line is not None else ", no result line"), file=sys.stderr)
        sys.exit(rc if rc != 0 else 1)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--workload", default=os.environ.get("PIPEINFER_WORKLOAD", "70b"), choices=list(WORKLOADS))
    ap.add_argument("--n-predict", type=int, default=128)
    ap.add_argument("--n-prompt", type=int, default=128)
    ap.add_argument("--n-draft", type=int, default=3)
    ap.add_argument("--n-parallel", type=int, default=3)
    ap.add_argument("--p-accept", type=float, default=0.1)
    ap.add_argument("--p-split", type=float, default=0.9)
    ap.add_argument("--n-ctx", type=int, default=1024)
    ap.add_argument("--signal", type=float, default=float(os.environ.get("PIPEINFER_SIGNAL", "1.0")))
    ap.add_argument("--noise", type=float, default=float(os.environ.get("PIPEINFER_NOISE", "0.25")))
    ap.add_argument("--draft-miss", type=float, default=float(os.environ.get("PIPEINFER_DRAFT_MISS", "0.3")),
                    help="fraction of the vocabulary the synthetic draft model predicts wrongly (acceptance ~ 1 - this)")
    ap.add_argument("--cache", default=os.environ.get("PIPEINFER_CACHE", "/tmp/pipeinfer_models"))
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-schedules", action="store_true", help="skip the untimed-for-`value` runs of modes 1 and 2")
    ap.add_argument("--cpu-tokens", type=int, default=8)
    ap.add_argument("--mode", type=int, default=0, help="schedule of the timed steps: 0 PipeInfer (reference-faithful, the default), 1 synchronous speculation, 2 target only")
    ap.add_argument("--min-spec-iters", type=int, default=0)
    ap.add_argument("--max-inflight", type=int, default=0)
    ap.add_argument("--adaptive-window", type=int, default=0, help="our adaptive draft pause (0: off = the reference behaviour)")
    ap.add_argument("--adaptive-pause", type=int, default=0)
    ap.add_argument("--detail", default=os.environ.get("PIPEINFER_BENCH_DETAIL", "/tmp/pipeinfer_bench_detail.json"),
                    help="where the full result (per-instance table, every family, cancel / segment statistics) is written; the stdout line stays small")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started as ONE command (the reference starts every rank from one mpirun: ggml-mpi.c:58-61 MPI_Init_thread, speculative.cpp:116-139 comm split):
        # this process only spawns the ranks and forwards rank 0's line — it never touches the GPU itself
        return launch_ranks(args.gpus)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        from pipeinfer_amd import pipeline
        return pipeline.bench_main(args, rank, world)

    tspec, tft, dspec, dft = WORKLOADS[args.workload]
    tgt_path = ensure_model(args.cache, tspec, tft, 0x5EED, args.signal, args.noise)
    dft_path = ensure_model(args.cache, dspec, dft, 0xD4AF, args.signal, args.noise, flip=args.draft_miss)

    lib = L.load()
    L.check(lib.pi_device_init(0), "pi_device_init")
    t0 = time.time()
    mt, md = LL.Model(tgt_path), LL.Model(dft_path)
    ct, cd = LL.Context(mt, n_ctx=args.n_ctx, n_batch=512), LL.Context(md, n_ctx=args.n_ctx, n_batch=512)
    cd.set_priority(int(os.environ.get("PIPEINFER_DRAFT_PRIORITY", "1")))
    ct.set_priority(int(os.environ.get("PIPEINFER_TARGET_PRIORITY", "0")))
    draft_cus = int(os.environ.get("PIPEINFER_DRAFT_CUS", "0"))
    if draft_cus > 0:                                  # disjoint CU masks: the draft's stream on the first `draft_cus` compute units, the target's on the rest
        cd.set_cu_range(0, draft_cus); ct.set_cu_range(draft_cus, 256)
    target_cu_lo = int(os.environ.get("PIPEINFER_TARGET_CU_LO", "0"))
    if target_cu_lo > 0 and not draft_cus:             # only the TARGET is confined (to CUs [lo, 256)): the draft's kernels may start at once on the CUs it leaves free and use the others as they free up
        ct.set_cu_range(target_cu_lo, 256)
    # one GPU: the target's stage graphs are launched segment by segment by a feeder thread, so that a cancelled speculative run stops at the
    # next segment boundary and the (high-priority) draft is served between segments instead of behind whole target runs
    # Segment size = how often the draft gets the GPU (between segments the feeder lets the priority context go first) and how much of a run is enqueued ahead of the GPU.
    # Round 3 (run coalescing + kernel-granular cancel, 70B workload, accepted tok/s of the PipeInfer schedule / ms per draft step): 40 layers per segment 79.9 / 4.9,
    # 20: 98.0 / 2.5, 16: 93.2 / 2.2, 10: 85.1 / 2.0 (96.8 with passes capped at 8 tokens), 5: 89.1 / 1.8.  A speculative run of three drafted tokens has to be queued
    # before the non-speculative run in flight finishes (12 ms): that needs a draft step under 4 ms; serving the draft faster than that only feeds more speculative runs
    # that continuous speculation cancels again.  Default: a quarter of the layers.  (Round 2, one forward pass per run: 2 layers 28, 8: 66, 40: 73, one graph without a feeder 82.)
    seg_layers = int(os.environ.get("PIPEINFER_SEG_LAYERS", str(max(1, synth.SPECS[tspec][1] // 4))))
    if seg_layers > 0:
        ct.set_segments(seg_layers)
    # Engine feature, not a driver change: the target's feeder takes every live run waiting in its queue into one forward pass (include/pipeinfer.h: run coalescing).
    # On ONE GPU the PipeInfer driver's non-speculative run for a freshly sampled token and the speculative run(s) drafted meanwhile wait behind the run in flight; carried
    # together they stream the 41 GB of weights once instead of once each.  Every token's bits are those it gets alone (tests/test_gpu_engine.py).
    # Width of a pass (default_coalesce).  Round 4, 70B, with the results written by a kernel (a draft call 2.8 -> 1.8 ms beside the target): 16 tokens 92.8 tok/s
    # (the faster draft queues more speculative runs than the target can verify before they are cancelled), 12: 98.9, 10: 98.9, 8: 102-105, 7: 101.6, 6: 86.
    coalesce = int(os.environ.get("PIPEINFER_COALESCE", str(default_coalesce(args.n_draft)))) if seg_layers > 0 else 0
    ct.prepare(); cd.prepare()
    ct.set_coalescing(coalesce)
    print(f"[bench] models resident in HBM after {time.time() - t0:.1f}s", file=sys.stderr)

    spec = dict(n_predict=args.n_predict, n_draft=args.n_draft, n_parallel=args.n_parallel, p_accept=args.p_accept, p_split=args.p_split,
                ignore_eos=1, mode=args.mode, min_spec_iters=args.min_spec_iters, max_inflight=args.max_inflight,
                adaptive_window=args.adaptive_window, adaptive_pause=args.adaptive_pause)
    rng = np.random.default_rng(0xBEEF)
    prompts = [[int(x) for x in rng.integers(3, mt.n_vocab, args.n_prompt)] for _ in range(args.warmup + args.steps + 1)]

    def generation(i, **over):
        ct.kv_clear(); cd.kv_clear()
        kw = dict(spec); kw.update(over)
        return LL.speculative(ct, cd, prompts[i], **kw)

    # lossless check (also the first warm-up): target-only greedy == the timed schedule's output on the same prompt
    ref_tokens, _ = generation(0, mode=2)
    spec_tokens, _ = generation(0)
    lossless = ref_tokens[:args.n_predict] == spec_tokens[:args.n_predict]
    first_diff = None if lossless else lossless_diagnostic(ct, prompts[0], ref_tokens, spec_tokens, args.n_predict)
    for i in range(args.warmup):
        generation(1 + i)

    ct.reset_stats(); cd.reset_stats()
    L.check(lib.pi_device_sync())
    # launch accounting of the timed region: graphs count themselves (contents x replays, read back below); eager launches (PI_NO_GRAPHS=1 runs under the profiler) are
    # counted by an open window that times next to nothing (every 2^30th launch of an instance)
    lib.pi_kernel_timing(1, 1 << 30)
    t_start = time.time()
    results = []
    for i in range(args.steps):
        _, r = generation(1 + args.warmup + i)
        results.append(r)
    L.check(lib.pi_device_sync())
    t_total = time.time() - t_start
    timed_counts = launch_rows(lib, ct) + [r for r in timing_rows(lib)]            # target graphs + whatever ran eagerly (both contexts when graphs are off)
    draft_counts = launch_rows(lib, cd)
    timed_aborted = aborted_launches(lib, ct)                                      # launches of cancelled runs that returned on entry: not priced as work
    lib.pi_kernel_timing(0, 8)
    st_t, st_d = ct.stats(), cd.stats()                 # per-call GPU time and bytes of the timed steps only
    main_sum = summarize(results)

    # the other two schedules of the reference's comparison (plot.py:6-7) on the same prompts; never `value`
    other = {}
    if not args.no_other_schedules:
        for mode, name in ((2, "target_only"), (1, "speculative_orig")):
            if mode == args.mode:
                continue
            outs = [generation(1 + args.warmup + i, mode=mode) for i in range(min(args.steps, 2))]
            s = summarize([r for _, r in outs])
            s["lossless_vs_target_only"] = bool(generation(0, mode=mode)[0][:args.n_predict] == ref_tokens[:args.n_predict]) if mode != 2 else True
            other[name] = s
        if coalesce > 1 and args.mode == 0:            # the same schedule with one forward pass per run, as the reference's stages compute them
            ct.set_coalescing(0)
            s = summarize([generation(1 + args.warmup + i)[1] for i in range(min(args.steps, 2))])
            s["lossless_vs_target_only"] = bool(generation(0)[0][:args.n_predict] == ref_tokens[:args.n_predict])
            other["pipeinfer_one_pass_per_run"] = s
            ct.set_coalescing(coalesce)

    # roofline of the dominant kernel.  Events recorded inside a captured graph cannot be read back on ROCm 7.2, so the sampling pass
    # replays one generation of the timed schedule kernel by kernel (graphs off) right after the timed region, on the same contexts;
    # every 5th launch of each (type, width) instance goes through hipExtLaunchKernelGGL whose events carry the kernel's own timestamps.
    ct.set_graphs(0)
    lib.pi_kernel_timing(1, 5)          # 5 is coprime to the 3- and 4-launch periods of a layer: every launch shape gets sampled
    generation(1 + args.warmup, n_predict=min(48, args.n_predict))
    L.check(lib.pi_device_sync())
    ct.set_graphs(1)
    samples = timing_rows(lib)
    lib.pi_kernel_timing(0, 8)
    # the kernel FAMILY the timed region spent most GPU time in: its launches there (counts) x the replay's per-instance durations
    by_family, table = roofline_tables(timed_counts, samples, timed_aborted)
    dom_name = next(iter(by_family), None)
    if dom_name:
        dom = by_family[dom_name]
        tr, traffic_src = committed_traffic(args.workload, dom_name)
        # traffic per launch: the committed PMC pass measured FETCH_SIZE per launch of this family AND (same run, same accounting as here) its algorithmic bytes per
        # launch; their ratio (bytes fetched per algorithmic byte) applied to this run's algorithmic bytes per launch
        traffic = round(tr["fetch_over_alg"] * dom["alg_bytes_per_launch"]) if tr else None
        roofline = {"bound": "hbm", "kernel": dom["kernels"], "family": dom_name, "achieved": dom["achieved_GBps"], "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                    "traffic": traffic, "traffic_fetch_over_alg": tr["fetch_over_alg"] if tr else None, "traffic_source": traffic_src, "kernel_src_sha16": kernel_source_sha(),
                    "avg_launch_us": dom["avg_launch_us"], "alg_bytes_per_launch": dom["alg_bytes_per_launch"], "launches": dom["launches"], "aborted_launches": dom["aborted_launches"],
                    "share_of_gpu_time": dom["share_of_gpu_time"],
                    "how": "launch counts and algorithmic bytes: the timed region's own (captured graphs x replays) minus the launches that returned on entry because their run had been "
                           "cancelled (device-side counters per family); durations: HIP-event samples of a graphs-off replay of the same schedule, per instance, launches that did the work only "
                           "(samples below a third of the instance's median are left out and counted)"}
    else:
        roofline = {"bound": "hbm", "kernel": None, "achieved": 0.0, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": 0.0, "traffic": None}
    roofline["by_family"] = by_family
    for name, f in by_family.items():                                                # PMC traffic of every family the committed pass has
        tr, _ = committed_traffic(args.workload, name)
        f["traffic_fetch_over_alg"] = tr["fetch_over_alg"] if tr else None
    roofline["draft_launches"] = sum(r["launches"] for r in draft_counts)
    if os.environ.get("PI_NO_GRAPHS"):
        # profiled runs launch everything eagerly: the launch accounting of the WHOLE process (both contexts, every generation of this run), per family — what
        # tools_dev/pmc_to_json.py divides the counters' bytes by (same launches, same population)
        life = {}
        for r in lifetime_rows(lib):
            f = life.setdefault(family_of(r), {"launches": 0, "alg_bytes": 0.0})
            f["launches"] += r["launches"]; f["alg_bytes"] += r["bytes_total"]
        roofline["process_eager_launches"] = {k: {"launches": v["launches"], "alg_bytes_per_launch": round(v["alg_bytes"] / max(1, v["launches"]))} for k, v in life.items()}
    roofline["per_instance"] = table
    # whole decode calls of the timed steps (prompt batches excluded): GPU time between the call's first and last event
    roofline["target_decode_call"] = {"calls": st_t["n_calls_1"], "gpu_ms_per_call": round(st_t["gpu_ms_1"] / max(1, st_t["n_calls_1"]), 4),
                                      "alg_GB_per_call": round(st_t["alg_bytes_1"] / max(1, st_t["n_calls_1"]) / 1e9, 4),
                                      "GBps": round(st_t["alg_bytes_1"] / max(1e-9, st_t["gpu_ms_1"] * 1e-3) / 1e9, 1)}
    roofline["target_all_calls"] = {"calls": st_t["n_calls"], "gpu_ms_per_call": round(st_t["gpu_ms"] / max(1, st_t["n_calls"]), 4),
                                    "GBps": round(st_t["alg_bytes"] / max(1e-9, st_t["gpu_ms"] * 1e-3) / 1e9, 1)}
    roofline["target_segments"] = {"layers_per_segment": seg_layers, "launched": st_t["n_segments"], "skipped_by_cancel": st_t["n_segments_skipped"]}
    # early cancellation: runs that llama_cancel_run reached AFTER their start return from every kernel on entry from then on (abort word) and the feeder drops their
    # unlaunched segments; `skipped_frac` = 1 - (GPU time those runs still took) / (what the same number of complete calls takes)
    avg_call_ms = st_t["gpu_ms"] / max(1, st_t["n_calls"])
    roofline["target_cancelled_runs"] = {"finished_as_cancelled": st_t["n_cancel_runs"], "cancelled_after_start": st_t["n_cancel_runs_started"],
                                         "gpu_ms_still_spent": round(st_t["cancel_gpu_ms"], 2), "gpu_ms_of_as_many_complete_calls": round(st_t["n_cancel_runs_started"] * avg_call_ms, 2),
                                         "skipped_frac": round(1.0 - st_t["cancel_gpu_ms"] / max(1e-9, st_t["n_cancel_runs_started"] * avg_call_ms), 4) if st_t["n_cancel_runs_started"] else None,
                                         "kernel_abort_word": os.environ.get("PI_NO_ABORT_WORD") is None}
    roofline["target_passes"] = {"passes": st_t["n_passes"], "runs": st_t["n_pass_runs"], "tokens": st_t["n_pass_tokens"], "passes_of_several_runs": st_t["n_coalesced_passes"], "reaction_windows": st_t["n_reaction_windows"]}
    roofline["draft_decode_call"] = {"calls": st_d["n_calls_1"], "gpu_ms_per_call": round(st_d["gpu_ms_1"] / max(1, st_d["n_calls_1"]), 4),
                                     "GBps": round(st_d["alg_bytes_1"] / max(1e-9, st_d["gpu_ms_1"] * 1e-3) / 1e9, 1)}

    sched = {0: "PipeInfer (asynchronous continuous speculation, examples/speculative)", 1: "synchronous tree speculation (examples/speculative_orig)", 2: "target only"}[args.mode]
    out = {"metric": METRIC if args.workload == "70b" else f"accepted tokens/sec + acceptance rate, {tspec} {tft} target / {dspec} draft", "value": main_sum["tokens_per_s"], "unit": "tokens/s",
           "n_gpus": 1, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(1e3 * t_total / args.steps, 2), "higher_is_better": True,
           "scaling": "strong", "vs_baseline": None, "dtype": "int8 x int4/6 dot, fp32 accumulate (f16 KV)", "data": "synthetic",
           "config": {"workload": f"{tspec} {tft} target + {dspec} {dft} draft on 1xMI355X, two HIP streams", "schedule": sched,
                      "n_prompt": args.n_prompt, "n_predict": args.n_predict, "n_draft": args.n_draft, "n_parallel": args.n_parallel,
                      "p_accept": args.p_accept, "p_split": args.p_split, "n_ctx": args.n_ctx, "sampling": "greedy",
                      "synthetic_structure": f"bigram signal={args.signal} noise={args.noise}, draft wrong on {args.draft_miss:.0%} of the vocabulary",
                      "engine": {"run_coalescing_max_tokens": coalesce, "kernel_abort_word": os.environ.get("PI_NO_ABORT_WORD") is None, "target_layers_per_segment": seg_layers},
                      "driver_additions": {"adaptive_window": args.adaptive_window, "adaptive_pause": args.adaptive_pause, "min_spec_iters": args.min_spec_iters, "max_inflight": args.max_inflight}},
           "acceptance": {k: v for k, v in main_sum.items() if k != "tokens_per_s"},
           "lossless_vs_target_only": bool(lossless), "first_difference": first_diff, "other_schedules": other,
           "ttft_s": round(float(np.mean([r["ttft_s"] for r in results])), 5),
           "itl_s": round(float(np.mean([r["itl_s"] for r in results])), 6), "encode_tps": round(float(np.mean([r["encode_tps"] for r in results])), 1),
           "roofline": roofline}
    if not args.no_cpu_baseline:
        for c in (ct, cd):
            c.free()
        try:
            out["cpu_baseline"] = cpu_baseline(tspec, tft, tgt_path, args.cache, args.signal, args.noise, args.cpu_tokens)
        except Exception as e:  # the GPU result stands on its own; say why the CPU leg is missing
            out["cpu_baseline"] = {"value": None, "unit": "tokens/s", "cores": os.cpu_count(), "kind": "unavailable", "sample": repr(e)}
    emit(out, args.detail)


if __name__ == "__main__":
    main()
