"""GPU: the configuration bench.py's `value` runs in — PipeInfer driver (mode 0), the target context segmented (a quarter of its layers per segment) behind its feeder thread
with run coalescing at bench.py's width (8 tokens at n_draft 3), the draft a priority context on the same GPU — with EVERY run compared against the oracle: the run tap records each run's metadata block and the rows it
got back (most of them out of a pass shared with other runs: per-token n_kv and cell table in the attention launches, compose_pass, pass_conflict's same-computation exemption,
members cancelled inside a pass), and the oracle replays the runs one by one on the same cells.  usage: headline_trace_worker.py <workload[-lN]> [n_predict] [n_draft] [n_parallel] [p_accept]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pipeinfer_amd import llama as LL, lib as L
from tests import oracle_lib as O
from tests.oracle_follow import RunLog
import bench as B

wl = sys.argv[1] if len(sys.argv) > 1 else "70b-l4"
n_predict = int(sys.argv[2]) if len(sys.argv) > 2 else 48
n_draft = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n_parallel = int(sys.argv[4]) if len(sys.argv) > 4 else 3
p_accept = float(sys.argv[5]) if len(sys.argv) > 5 else 0.1
lib = L.load(); L.check(lib.pi_device_init(0))
cut = None
if "-l" in wl:
    wl, nl = wl.split("-l"); cut = (8, int(nl))
tspec, tft, dspec, dft = B.WORKLOADS[wl]
cache = os.environ.get("PIPEINFER_CACHE", "/tmp/pipeinfer_models")
tgt_path = B.ensure_model(cache, tspec, tft, 0x5EED, 1.0, 0.25, layers=cut)
dft_path = B.ensure_model(cache, dspec, dft, 0xD4AF, 1.0, 0.25, flip=0.3)
N_CTX, N_PROMPT, N_BATCH = 1024, 32, 512
mt, md = LL.Model(tgt_path), LL.Model(dft_path)
api = LL.api(); api.pipeinfer_model_n_layer.argtypes = [C.c_void_p]; api.pipeinfer_model_n_layer.restype = C.c_int32
n_layer = api.pipeinfer_model_n_layer(mt.h)
n_vocab = mt.n_vocab
rng = np.random.default_rng(0xBEEF)
prompt = [int(x) for x in rng.integers(3, mt.n_vocab, N_PROMPT)]

outs, log, stats, res0 = {}, RunLog(), None, None
for mode in (2, 0):
    ct, cd = LL.Context(mt, n_ctx=N_CTX, n_batch=N_BATCH), LL.Context(md, n_ctx=N_CTX, n_batch=N_BATCH)
    if mode == 0:                                              # bench.py's engine settings for the one-GPU line
        ct.set_segments(max(1, n_layer // 4)); cd.set_priority(1)
        ct.prepare(); cd.prepare()
        ct.set_coalescing(B.default_coalesce(n_draft))
        ct.set_run_tap(log.tap)
    out, res = LL.speculative(ct, cd, prompt, n_predict=n_predict, n_draft=n_draft, n_parallel=n_parallel, p_accept=p_accept, p_split=0.9, ignore_eos=1, mode=mode)
    outs[mode] = [int(x) for x in out[:n_predict]]
    if mode == 0:
        stats, res0 = ct.stats(), res
    ct.free(); cd.free()
print("runs", len(log.order), "passes", {k: stats[k] for k in ("n_passes", "n_pass_runs", "n_pass_tokens", "n_coalesced_passes", "n_cancel_runs", "n_cancel_runs_started")},
      {k: res0[k] for k in ("n_runs_spec", "n_runs_nonspec", "n_runs_canceled", "n_accept_clean")})
md.free(); mt.free()
o = O.oracle()
o.o_set_threads(int(os.environ.get("ORACLE_THREADS", "64")))
oh = o.o_model_load(tgt_path.encode(), N_CTX, 8)
assert oh
compared, bad = log.replay(o, oh, N_CTX, n_vocab)
print("compared", compared, "of", len(log.order), "runs; oracle mismatches", len(bad), bad[:5])
same = outs[0] == outs[2]
print("speculative == target-only:", same)
ok = same and not bad and compared > 0 and stats["n_coalesced_passes"] > 0
print("HEADLINE-TRACE", "OK" if ok else "FAIL", "coalesced_passes", stats["n_coalesced_passes"], "compared", compared, "cancelled_after_start", stats["n_cancel_runs_started"])
sys.exit(0 if ok else 2)
