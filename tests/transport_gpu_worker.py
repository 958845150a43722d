"""3-rank dry run of the library's transport on ONE GPU (launched by torch.distributed.run from tests/test_gpu_engine.py): rank 0 = the C++
PipeInfer driver + draft context + proxy target context, ranks 1 and 2 = HIP stage contexts over layer ranges of the target, all on
device 0.  The activations travel as peer stores into the next stage's HBM ring (hipIpc between processes), the metadata, results and
cancel notes through the shared-memory segment; torch is only used for the rendezvous (gloo) and never touches the GPU here.
argv: out.json target_stem draft_stem seed ranges("0-1,1-3")"""
import ctypes as C
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pipeinfer_amd import llama as LL, pipeline as P          # noqa: E402
from tests import scenarios as S                              # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
N_CTX, N_BATCH = 256, 64


def main():
    import faulthandler
    faulthandler.dump_traceback_later(int(os.environ.get("PI_TEST_WATCHDOG", "240")), exit=True)
    out_path, tgt, dft, seed = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4])
    ranges = [tuple(int(v) for v in r.split("-")) for r in sys.argv[5].split(",")]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == len(ranges) + 1
    path = os.path.join(GOLD, tgt + ".gguf")
    stage = None
    if rank == 0:
        mt, md = LL.Model(path), LL.Model(os.path.join(GOLD, dft + ".gguf"))
        mt.lib.llama_n_embd.argtypes = [C.c_void_p]
        n_embd, n_vocab = mt.lib.llama_n_embd(mt.h), mt.n_vocab
    else:
        l0, l1 = ranges[rank - 1]
        stage = P.HipStage(path, l0, l1, rank == 1, rank == world - 1, N_CTX, N_BATCH, device_index=0, seg_layers=1)
        if os.environ.get("PIPEINFER_TEST_STAGE_COALESCE") and world == 2:         # the stage is the whole target: it carries the runs waiting in its ring in one pass
            stage.ctx.set_coalescing(int(os.environ["PIPEINFER_TEST_STAGE_COALESCE"]))
        n_embd, n_vocab = stage.n_embd, stage.n_vocab
    dist.barrier()
    tp = P.Transport(P.job_name(), rank, world, n_embd, n_vocab, N_BATCH, N_CTX, device=0, timeout_ms=120000)
    if rank >= 1:
        st = tp.serve(ctx=stage.ctx)
        box = [None] * world
        dist.gather_object({"rank": rank, "device_mode": tp.device_mode, **st, **{k: stage.stats()[k] for k in ("n_calls", "n_segments", "n_segments_skipped", "n_passes", "n_pass_runs", "n_coalesced_passes")}}, None, dst=0)
        dist.barrier()
        tp.close()
        stage.close()
        return
    g = np.load(os.path.join(GOLD, tgt + ".npz"))
    prompt, _ = S.scenario_greedy(512, seed=seed, n_prompt=8)
    want = [int(x) for x in g["greedy_tokens"][:24]]
    verdict = {"device_mode": tp.device_mode, "trials": []}
    for n_draft, n_parallel, min_spec in ((3, 3, 1), (4, 2, 3), (2, 1, 0)):
        ct, cd = LL.Context(mt, n_ctx=N_CTX, n_batch=N_BATCH), LL.Context(md, n_ctx=N_CTX, n_batch=N_BATCH)
        tp.attach_head(ct)
        out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=n_draft, n_parallel=n_parallel, p_accept=0.05, p_split=0.9, ignore_eos=0, min_spec_iters=min_spec)
        verdict["trials"].append({"out": [int(x) for x in out[:24]], "want": want, "n_runs_spec": res["n_runs_spec"], "n_runs_canceled": res["n_runs_canceled"],
                                  "n_accept_clean": res["n_accept_clean"], **{k: tp.stats()[k] for k in ("runs", "skipped", "cancel_notes", "received")}})
        ct.free(); cd.free()
    tp.shutdown()
    box = [None] * world
    dist.gather_object(None, box, dst=0)
    verdict["stages"] = box[1:]
    dist.barrier()
    tp.close()
    with open(out_path, "w") as f:
        json.dump(verdict, f)


if __name__ == "__main__":
    main()
