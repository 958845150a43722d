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
