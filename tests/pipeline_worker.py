"""world_size-2 worker for tests/test_pipeline_gloo.py (launched by torch.distributed.run, gloo, no GPU).

rank 0  the C++ PipeInfer driver (pipeinfer_speculative) with the TARGET context as the proxy of a remote pipeline
        (pipeinfer_transport_attach_head: the library's shared-memory transport) and the DRAFT context served in-process
rank 1  pipeinfer_transport_serve — the whole target model as one stage (world 3: a relay stage + the model)

The layers themselves are computed by the ORACLE (oracle/model.c o_model_forward) on both sides: this is test infrastructure
standing in for pipeinfer_stage_run, so that the transport, the metadata block, early cancellation and the driver are
exercised without a GPU.  Writes its verdict as JSON to argv[1] (rank 0 only)."""
import collections
import ctypes as C
import json
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipeinfer_amd import llama as LL, pipeline as P          # noqa: E402
from tests import oracle_lib as O, scenarios as S             # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
N_CTX = 256


class OracleCompute:
    """metadata block -> (top-8 ids, logits of those, sum exp, full logits) through the oracle's forward pass"""

    def __init__(self, stem):
        self.o = O.oracle()
        self.h = self.o.o_model_load(os.path.join(GOLD, stem + ".gguf").encode(), N_CTX, 8)
        assert self.h
        self.n_vocab = self.o.o_model_info(self.h, 0)

    def run(self, n, meta, x_in=None, x_out=None):
        meta = np.ascontiguousarray(meta, np.int32)
        words = (N_CTX + 31) // 32
        tok, pos, dyn = meta[:n], meta[n:2 * n], meta[2 * n:2 * n + 4]
        kv_head, n_kv = int(dyn[0]), int(dyn[1])
        bits = meta[2 * n + 4:2 * n + 4 + n * words].view(np.uint32).reshape(n, words)
        vis = np.unpackbits(bits.view(np.uint8).reshape(n, -1), axis=1, bitorder="little")[:, :n_kv]
        mask = np.where(vis == 1, 0.0, -np.inf).astype(np.float32)
        logits = np.zeros((n, self.n_vocab), np.float32)
        rc = self.o.o_model_forward(self.h, np.ascontiguousarray(tok).ctypes.data, np.ascontiguousarray(pos).ctypes.data, n, kv_head, n_kv,
                                    np.ascontiguousarray(mask).ctypes.data, logits.ctypes.data)
        assert rc == 0
        ids = np.zeros((n, 8), np.int32); vals = np.zeros((n, 8), np.float32); sums = np.zeros(n, np.float32)
        for i in range(n):
            order = np.lexsort((np.arange(self.n_vocab), -logits[i]))[:8]           # logit descending, lower id first on ties
            ids[i], vals[i] = order, logits[i][order]
            sums[i] = np.exp(logits[i] - logits[i].max(), dtype=np.float32).sum(dtype=np.float32)
        return ids, vals, sums, logits


class Relay:
    """first of two stages in the world_size-3 test: the oracle has no partial forward, so this stage computes nothing and hands the next
    one an activation block derived from the run's tokens — the last stage checks it arrived intact (header hop, data hop, ordering)"""

    @staticmethod
    def pattern(n, meta, n_embd):
        tok = np.asarray(meta[:n], np.float32)
        return (tok[:, None] * 0.5 + np.arange(n_embd, dtype=np.float32)[None, :] * 0.25).astype(np.float32)

    def __init__(self, n_embd):
        self.n_embd = n_embd

    def run(self, n, meta, x_in=None, x_out=None):
        want = self.pattern(n, meta, self.n_embd)
        if x_in is not None:                                # a relay behind a relay (world_size >= 4): what the stage before it stored must have arrived intact
            assert np.array_equal(np.asarray(x_in), want), "activation block changed on its way between two relays"
        x_out[:] = want
        return None


class CheckedOracle(OracleCompute):
    """last of two stages: verifies the relayed block, then runs the whole oracle model on the metadata"""

    def run(self, n, meta, x_in=None, x_out=None):
        want = Relay.pattern(n, np.ascontiguousarray(meta, np.int32), x_in.shape[1])
        assert np.array_equal(np.asarray(x_in), want), "activation block changed on its way between the stages"
        return super().run(n, meta)


class LocalOps:
    """struct pipeinfer_remote_ops served in-process (the draft context of the head)"""

    def __init__(self, compute):
        self.compute, self.done = compute, collections.OrderedDict()
        self.ops = P.RemoteOps(None, P.START_FN(self._start), P.POLL_FN(self._poll), P.WAIT_FN(self._wait), P.CANCEL_FN(self._cancel))

    def attach(self, ctx):
        ctx.lib.pipeinfer_context_set_remote.argtypes = [C.c_void_p, C.c_void_p]
        ctx.lib.pipeinfer_context_set_remote.restype = None
        ctx.lib.pipeinfer_context_set_remote(ctx.h, C.byref(self.ops))

    def _start(self, user, run_seq, batch_id, n, meta, nbytes):
        block = np.frombuffer(C.string_at(meta, (nbytes + 3) // 4 * 4), dtype=np.int32).copy()
        self.done[run_seq] = (n, self.compute.run(n, block))
        return 0

    def _poll(self, user):
        return 1 if self.done else 0

    def _wait(self, user, run_seq, logits, ids, vals, sums):
        n, (i8, v8, s, lg) = self.done.pop(run_seq)
        C.memmove(ids, i8.ctypes.data, i8.nbytes); C.memmove(vals, v8.ctypes.data, v8.nbytes)
        C.memmove(sums, s.ctypes.data, s.nbytes); C.memmove(logits, lg.ctypes.data, lg.nbytes)
        return 1

    def _cancel(self, user, batch_id):
        pass


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("PI_TEST_WATCHDOG", "100")), exit=True)
    out_path, tgt, dft = sys.argv[1], sys.argv[2], sys.argv[3]
    dist.init_process_group("gloo")
    rank = dist.get_rank()
    world = dist.get_world_size()
    assert 2 <= world <= 5
    probe = OracleCompute(tgt)
    n_embd, n_vocab = probe.o.o_model_info(probe.h, 1), probe.n_vocab
    tp = P.Transport(P.job_name(), rank, world, n_embd, n_vocab, 64, N_CTX, device=-1, timeout_ms=60000)
    assert not tp.device_mode
    if rank >= 1:
        if world == 2:
            comp = probe
        else:
            comp = CheckedOracle(tgt) if rank == world - 1 else Relay(n_embd)     # every stage but the last relays (the oracle runs whole models): headers, activations and skips still cross every hop
        run_fn = comp.run
        lat = float(os.environ.get("PI_TEST_STAGE_LATENCY_MS", "0"))
        if lat > 0:                                         # a slow stage: the head's runs queue up in the ring behind the one being computed (coalescing test)
            import time

            def run_fn(n, meta, x_in=None, x_out=None, _f=comp.run):
                time.sleep(lat / 1e3)
                return _f(n, meta, x_in, x_out)
        st = tp.serve(backend=run_fn)
        dist.gather_object(st, None, dst=0)                 # the stage's counters go into the verdict: conservation is asserted by the test
        dist.barrier()
        tp.close()
        print("stage", rank, "served", st, file=sys.stderr)
        return
    g = np.load(os.path.join(GOLD, tgt + ".npz"))
    prompt, _ = S.scenario_greedy(512, seed=int(sys.argv[4]), n_prompt=8)
    want = [int(x) for x in g["greedy_tokens"][:24]]
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    verdict = {"trials": []}
    for trial, (n_draft, n_parallel, min_spec) in enumerate(((3, 3, 1), (4, 2, 2), (2, 1, 0))):
        ct, cd = LL.Context(mt, n_ctx=N_CTX, n_batch=64), LL.Context(md, n_ctx=N_CTX, n_batch=64)
        tp.attach_head(ct)                                  # one transport, several generations
        local = LocalOps(OracleCompute(dft))
        local.attach(cd)
        out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=n_draft, n_parallel=n_parallel, p_accept=0.05, p_split=0.9, ignore_eos=0, min_spec_iters=min_spec, verbose=int(os.environ.get("V", "0")))
        verdict["trials"].append({"out": [int(x) for x in out[:24]], "want": want, "n_runs_spec": res["n_runs_spec"], "n_runs_canceled": res["n_runs_canceled"],
                                  "n_accept_clean": res["n_accept_clean"], **{k: tp.stats()[k] for k in ("runs", "skipped", "cancel_notes", "received")}})
        ct.free(); cd.free()
    tp.shutdown()
    stages = [None] * world
    dist.gather_object(None, stages, dst=0)
    verdict["stages"] = stages[1:]
    dist.barrier()
    tp.close()
    with open(out_path, "w") as f:
        json.dump(verdict, f)


if __name__ == "__main__":
    main()
