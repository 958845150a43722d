"""Per-run oracle comparison of a context's whole schedule (test infrastructure).

A context's run tap (pipeinfer_context_set_run_tap) hands over every run's metadata block at its start and its logits rows at its finish — whatever carried the run on the
GPU: one graph, segments behind the feeder thread, or a pass that several runs shared (run coalescing).  `replay` then executes the runs one after the other, in the order they
were started, through the oracle's forward pass on the same cells, the same n_kv and the same visibility rows (reference: llama_decode_internal, llama.cpp:5461-5827, one run at a
time), and compares bit for bit.  Runs the engine never computed (cancelled) are still executed by the oracle: nothing may depend on what they would have written."""
import numpy as np


class RunLog:
    def __init__(self):
        self.entries, self.order, self.runs = [], [], {}            # entries in start order; order: their run ids; runs: id -> the LATEST entry with that id

    def tap(self, run_id, batch_id, n, meta, rows):
        r = self.runs.get(run_id)
        if r is None or r["finished"]:                                      # a start (ids are per context and only repeat after 2^30 runs)
            r = dict(batch_id=batch_id, n=n, meta=meta, rows=None, finished=False)
            self.entries.append(r); self.order.append(run_id); self.runs[run_id] = r
        else:                                                               # the finish of the run that is open under this id
            r["rows"], r["finished"] = rows, True

    def replay(self, o, oh, n_ctx, n_vocab):
        """o: the oracle library (tests.oracle_lib.oracle()), oh: a model handle of o_model_load with the same n_ctx.  Returns (runs compared, mismatches)."""
        words = (n_ctx + 31) // 32
        compared, bad = 0, []
        for k, r in enumerate(self.entries):
            n, block = r["n"], r["meta"]
            tok, pos, dyn = block[:n], block[n:2 * n], block[2 * n:2 * n + 4]
            assert int(dyn[3]) == 0, "a run's own block never carries a cell table"
            kv_head, n_kv = int(dyn[0]), int(dyn[1])
            bits = block[2 * n + 4:2 * n + 4 + n * words].view(np.uint32).reshape(n, words)
            vis = np.unpackbits(bits.view(np.uint8).reshape(n, -1), axis=1, bitorder="little")[:, :n_kv]
            mask = np.where(vis == 1, 0.0, -np.inf).astype(np.float32)
            want = np.zeros((n, n_vocab), np.float32)
            rc = o.o_model_forward(oh, np.ascontiguousarray(tok).ctypes.data, np.ascontiguousarray(pos).ctypes.data, n, kv_head, n_kv, np.ascontiguousarray(mask).ctypes.data, want.ctypes.data)
            assert rc == 0, rc
            if r["rows"] is not None:
                compared += 1
                if not np.array_equal(want.view(np.uint32), r["rows"].view(np.uint32)):
                    bad.append((k, r["batch_id"], n, kv_head, n_kv, float(np.abs(want - r["rows"]).max())))
        return compared, bad
