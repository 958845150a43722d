"""CPU (-m "not gpu"): bench.py's roofline accounting — launches of the timed region (graph contents x replays) x per-instance durations of the graphs-off replay -> kernel
families, the dominant family and its achieved rate — on a hand-made table, and the oracle replay helper on a hand-made run log."""
import numpy as np

import bench as B
from tests.oracle_follow import RunLog


def row(family, type_, n_tok, shape, launches, bytes_total, sampled=0, avg_us=0.0, avg_sampled_bytes=0.0):
    return {"family": family, "type": type_, "n_tok": n_tok, "shape_bytes": shape, "launches": launches, "bytes_total": bytes_total, "sampled": sampled, "avg_us": avg_us, "avg_sampled_bytes": avg_sampled_bytes}


def test_roofline_tables_price_the_timed_regions_launches_with_the_replays_durations():
    MB = 1_000_000
    counts = [row(1, 12, 7, 264 * MB, 100, 100 * 264.0 * MB),          # matrix cores, up + gate at 7 tokens: 100 launches in the timed region
              row(1, 12, 5, 264 * MB, 10, 10 * 264.0 * MB),            # ... at 5 tokens: not sampled in the replay -> nearest sampled width of the same launch shape
              row(0, 12, 1, 120 * MB, 50, 50 * 120.0 * MB),            # one-token mat-vec
              row(2, 0, 7, 0, 200, 200 * 0.3 * MB),                    # norm + quantise
              row(5, 0, 7, 0, 100, 100 * 99.0 * MB),                   # attention: a graph's launch count carries n_kv = n_ctx; priced with the sampled (real) bytes
              row(3, 0, 9, 0, 4, 4 * 0.2 * MB)]                        # quantise at a width the replay never saw and no shape: nearest width of the family
    samples = [row(1, 12, 7, 264 * MB, 20, 20 * 264.0 * MB, sampled=4, avg_us=55.0, avg_sampled_bytes=264.0 * MB),
               row(0, 12, 1, 120 * MB, 30, 30 * 120.0 * MB, sampled=6, avg_us=25.0, avg_sampled_bytes=120.0 * MB),
               row(2, 0, 7, 0, 40, 40 * 0.3 * MB, sampled=8, avg_us=7.0, avg_sampled_bytes=0.3 * MB),
               row(5, 0, 7, 0, 20, 20 * 4.5 * MB, sampled=4, avg_us=14.0, avg_sampled_bytes=4.5 * MB),
               row(3, 0, 7, 0, 20, 20 * 0.2 * MB, sampled=4, avg_us=4.0, avg_sampled_bytes=0.2 * MB)]
    fam, inst = B.roofline_tables(counts, samples)
    assert next(iter(fam)) == "matrix_cores"                                             # ordered by estimated GPU time: 110 launches x 55 us
    mc = fam["matrix_cores"]
    assert mc["launches"] == 110 and mc["launches_priced"] == 110
    assert abs(mc["est_gpu_ms"] - 110 * 55.0e-3) < 1e-6
    assert abs(mc["achieved_GBps"] - 264.0 / 55.0e-6 / 1e3) < 1.0 and abs(mc["frac"] - mc["achieved_GBps"] / B.HBM_PEAK_GBS) < 1e-3
    assert fam["matvec_nt1"]["avg_launch_us"] == 25.0 and fam["norm_quant"]["launches"] == 200
    assert fam["attention"]["alg_bytes_per_launch"] == round(4.5 * MB)                   # the sampled bytes, not the graph's upper bound
    assert fam["quantize"]["launches_priced"] == 4                                       # priced with the family's nearest sampled width
    five = [r for r in inst if r["family"] == "matrix_cores" and r["n_tok"] == 5][0]
    assert five["avg_us"] == 55.0 and "nearest" in five["duration_from"]
    assert abs(sum(f["share_of_gpu_time"] for f in fam.values()) - 1.0) < 1e-3


def test_family_names():
    assert B.family_of({"family": 0, "n_tok": 1}) == "matvec_nt1" and B.family_of({"family": 0, "n_tok": 4}) == "matvec_multi"
    assert B.family_of({"family": 1, "n_tok": 7}) == "matrix_cores" and B.family_of({"family": 5, "n_tok": 1}) == "attention"


def test_run_log_orders_runs_by_their_start_and_keeps_the_rows_of_their_finish():
    log = RunLog()
    meta = np.zeros(16, np.int32)
    log.tap(7, 0, 1, meta, None)                       # start of run 7
    log.tap(8, 11, 3, meta, None)                      # start of run 8 (queued behind it)
    log.tap(7, 0, 1, meta, np.ones((1, 4), np.float32))   # finish of run 7 with rows
    log.tap(8, 11, 3, meta, None)                      # finish of run 8: cancelled, no rows
    log.tap(7, 0, 1, meta, None)                       # the id comes round again (a new run): a new entry
    assert log.order == [7, 8, 7]
    assert log.runs[8]["finished"] and log.runs[8]["rows"] is None


def test_the_stdout_line_stays_small_whatever_the_tables_hold(tmp_path):
    """BENCH_r04.parsed was null because the line carried a 123-row per_instance table (32.8 KB against the driver's 8.5 KB tail).  The line is built by
    compact_line() from the full result; the tables go to the detail file."""
    import json
    MB = 1_000_000
    counts, samples = [], []
    for i in range(150):                                                     # 150 instances: more than any schedule has produced
        fam, typ, nt, shape = i % 7, 12 + i % 3, 1 + i % 8, (30 + i) * MB if i % 7 <= 1 else 0
        counts.append(row(fam, typ, nt, shape, 1000 + i, (1000 + i) * 50.0 * MB))
        samples.append(row(fam, typ, nt, shape, 200, 200 * 50.0 * MB, sampled=40, avg_us=20.0 + i, avg_sampled_bytes=50.0 * MB))
    by_family, table = B.roofline_tables(counts, samples)
    assert len(table) == 150
    dom = by_family[next(iter(by_family))]
    call = {"calls": 1074, "gpu_ms_per_call": 18.7012, "alg_GB_per_call": 41.3211, "GBps": 3644.1}
    sched = {"tokens_per_s": 132.11, "accepted_draft_tokens": 164, "drafted_tokens": 288, "rate": 0.5694, "reference_formula_pct": 56.94, "spec_runs": 96, "nonspec_runs": 0,
             "canceled_runs": 0, "draft_paused_tokens": 0, "draft_catchups": 0, "lossless_vs_target_only": True}
    out = {"metric": B.METRIC, "value": 102.03, "unit": "tokens/s", "n_gpus": 1, "steps": 20, "warmup": 1, "ms_per_step": 1423.55, "higher_is_better": True, "scaling": "strong",
           "vs_baseline": None, "dtype": "int8 x int4/6 dot, fp32 accumulate (f16 KV)", "data": "synthetic",
           "config": {"workload": "llama-2-70b q4_k_m target + tinyllama-1.1b q4_k_m draft on 1xMI355X, two HIP streams", "schedule": "PipeInfer (asynchronous continuous speculation, examples/speculative)",
                      "n_prompt": 128, "n_predict": 128, "n_draft": 3, "n_parallel": 3, "p_accept": 0.1, "p_split": 0.9, "n_ctx": 1024, "sampling": "greedy",
                      "synthetic_structure": "bigram signal=1.0 noise=0.25, draft wrong on 30% of the vocabulary",
                      "engine": {"run_coalescing_max_tokens": 8, "kernel_abort_word": True, "target_layers_per_segment": 20},
                      "driver_additions": {"adaptive_window": 0, "adaptive_pause": 0, "min_spec_iters": 0, "max_inflight": 0}},
           "acceptance": dict(sched), "lossless_vs_target_only": True, "first_difference": None,
           "other_schedules": {"target_only": dict(sched), "speculative_orig": dict(sched), "pipeinfer_one_pass_per_run": dict(sched)},
           "ttft_s": 0.12711, "itl_s": 0.009803, "encode_tps": 1009.2,
           "roofline": {"bound": "hbm", "kernel": dom["kernels"], "family": next(iter(by_family)), "achieved": dom["achieved_GBps"], "peak": B.HBM_PEAK_GBS, "unit": "GB/s", "frac": dom["frac"],
                        "traffic": 124329000, "traffic_fetch_over_alg": 0.9957, "traffic_source": "profiles/r05_pmc_fetch_size_70b.json", "kernel_src_sha16": "a67b17026aa4e8c1",
                        "avg_launch_us": dom["avg_launch_us"], "alg_bytes_per_launch": dom["alg_bytes_per_launch"], "launches": dom["launches"], "aborted_launches": 1234,
                        "share_of_gpu_time": dom["share_of_gpu_time"], "how": "x" * 300, "by_family": by_family, "per_instance": table,
                        "target_decode_call": call, "target_all_calls": call, "draft_decode_call": call, "target_segments": {"layers_per_segment": 20, "launched": 4000, "skipped_by_cancel": 900},
                        "target_cancelled_runs": {"finished_as_cancelled": 1885, "note": "y" * 200}, "target_passes": {"passes": 1105, "runs": 3178, "tokens": 8742, "passes_of_several_runs": 700}},
           "cpu_baseline": {"value": 1.1894837294446785, "unit": "tokens/s", "cores": 32, "threads": 32, "host_cores": 256, "kind": "reference", "value_range": [0.991, 1.367],
                            "ms_per_token_8_layers": [65.72, 62.63, 71.06], "ms_per_token_16_layers": [165.72, 162.63, 171.06], "sample": "z" * 900}}
    assert len(json.dumps(out)) > 30000                                      # the full result is what broke the driver's parser
    detail = str(tmp_path / "detail.json")
    text = B.compact_line(out, detail)
    assert len(text) < B.LINE_BUDGET < 6000, len(text)
    line = json.loads(text)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["config"]["workload"].startswith("llama-2-70b") and line["value"] == 102.03
    rl = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_us", "alg_bytes_per_launch", "launches", "aborted_launches", "target_decode_call", "draft_decode_call"):
        assert k in rl, k
    assert "per_instance" not in rl and set(rl["by_family"][rl["family"]]) == {"share", "frac"}
    assert line["cpu_baseline"]["kind"] == "reference" and len(line["cpu_baseline"]["sample"]) <= 260
    assert set(line["other_schedules"]["speculative_orig"]) == {"tokens_per_s", "lossless_vs_target_only"}
    assert line["detail"] == detail
    # emit(): the detail file holds everything, stdout ends with the small line
    import contextlib
    import io
    so, se = io.StringIO(), io.StringIO()
    with contextlib.redirect_stdout(so), contextlib.redirect_stderr(se):
        B.emit(out, detail)
    assert json.load(open(detail))["roofline"]["per_instance"] == table
    assert so.getvalue().count("\n") == 1 and json.loads(so.getvalue())["value"] == 102.03 and len(so.getvalue()) < B.LINE_BUDGET + 1


def test_aborted_launches_are_kept_apart_from_the_launches_that_did_the_work():
    """launches of cancelled runs return on entry; they are neither averaged into the durations (the sampler leaves them out) nor priced with full bytes
    (the device counts them per family; bench takes them out of the family's launches).  frac = bytes of the launches that did the work / their time."""
    MB = 1_000_000
    counts = [row(1, 12, 6, 264 * MB, 100, 100 * 264.0 * MB), row(1, 12, 6, 38 * MB, 100, 100 * 38.0 * MB), row(5, 0, 6, 0, 100, 100 * 9.0 * MB)]
    samples = [dict(row(1, 12, 6, 264 * MB, 20, 20 * 264.0 * MB, sampled=12, avg_us=55.0, avg_sampled_bytes=264.0 * MB), sampled_dropped=8),
               dict(row(1, 12, 6, 38 * MB, 20, 20 * 38.0 * MB, sampled=12, avg_us=18.0, avg_sampled_bytes=38.0 * MB), sampled_dropped=8),
               row(5, 0, 6, 0, 20, 20 * 2.0 * MB, sampled=4, avg_us=14.0, avg_sampled_bytes=2.0 * MB)]
    fam0, _ = B.roofline_tables(counts, samples)
    fam, inst = B.roofline_tables(counts, samples, {1: 50, 5: 0})            # a quarter of the 200 matrix-core launches found their run aborted
    mc = fam["matrix_cores"]
    assert mc["launches"] == 200 and mc["aborted_launches"] == 50
    assert abs(mc["est_gpu_ms"] - 0.75 * fam0["matrix_cores"]["est_gpu_ms"]) < 0.01
    assert mc["frac"] == fam0["matrix_cores"]["frac"] and mc["avg_launch_us"] == fam0["matrix_cores"]["avg_launch_us"]      # the rate is that of the launches that did the work
    assert fam["attention"]["aborted_launches"] == 0 and fam["attention"]["est_gpu_ms"] == fam0["attention"]["est_gpu_ms"]
    assert mc["share_of_gpu_time"] < fam0["matrix_cores"]["share_of_gpu_time"]
    assert [r for r in inst if r["launch_MB"] == 264.0][0]["samples_left_out_as_aborted"] == 8


def test_cpu_baseline_extrapolation_reports_both_estimates_and_clamps_nothing():
    """(t8 - t4) / 4 = 10.5 ms per layer with t4 = 20.6 ms < 4 x 10.5 was clamped to a 0.00 ms head.  The leg now times 8- and 16-layer copies and reports
    the difference-style and the ratio-style estimate side by side."""
    c = B.cpu_extrapolation(80, 0.090, 0.170, [90.0, 95.0, 99.0], [170.0, 176.0, 181.0], 8, 8, "reference", 32)
    assert c["ms_per_layer"]["difference"] == 10.0 and c["ms_per_layer"]["t16_over_16"] == 10.625 and c["ms_head_difference_style"] == 10.0
    assert c["ms_per_token_full_model"] == {"difference": 810.0, "t16_over_16": 850.0}
    assert abs(c["value"] - 1.0 / 0.810) < 1e-9 and c["value_range"][0] <= c["value"] <= c["value_range"][1]
    # an inconsistent pair (a negative head) is shown, not hidden
    c = B.cpu_extrapolation(80, 0.060, 0.170, [60.0], [170.0], 8, 8, "reference", 32)
    assert c["ms_head_difference_style"] < 0 and c["ms_per_token_full_model"]["difference"] == round((0.060 - 8 * 0.01375 + 80 * 0.01375) * 1e3, 1)
    assert len(c["sample_short"]) < 400
