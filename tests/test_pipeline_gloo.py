"""N > 1 path on CPU: world_size 2 and 3 (torchrun + gloo for the rendezvous; the runs travel through the library's shared-memory transport
on host rings, the oracle stands in for the stage kernels).  The peer-store flavour of the same transport runs in tests/test_gpu_engine.py."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def test_split_layers_matches_reference_rule():
    from pipeinfer_amd import pipeline as P
    assert P.split_layers(80, 7) == [(0, 11), (11, 22), (22, 33), (33, 44), (44, 55), (55, 66), (66, 80)]       # last stage takes the remainder
    assert P.split_layers(32, 1) == [(0, 32)]
    assert P.split_layers(60, 3, [0.5, 0.25, 0.25]) == [(0, 30), (30, 45), (45, 60)]
    assert P.split_layers(4, 3, [0.9, 0.9, 0.9]) == [(0, 3), (3, 4), (4, 4)]                                   # clamped at the end of the model


@pytest.mark.parametrize("tgt,dft,seed,world", [("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, 2), ("tiny_llama_q4_k_m", "tiny_llama_q4_k_m", 11, 2),
                                                ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, 3)])      # 3: head + two stages (header and activation hop between stages)
def test_head_and_stage_over_gloo_are_lossless(tmp_path, tgt, dft, seed, world):
    """rank 0: C++ driver + proxy target context; rank 1 (and 2): pipeinfer_transport_serve.  The accepted tokens must equal the reference
    greedy continuation, every started run must come back (computed or skipped after a cancel) and cancels must reach the stages."""
    out = tmp_path / "verdict.json"
    env = dict(os.environ, OMP_NUM_THREADS="2", PYTHONPATH=ROOT)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "pipeline_worker.py"), str(out), tgt, dft, str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    v = json.loads(out.read_text())
    assert len(v["trials"]) == 3
    for t in v["trials"]:
        assert t["out"] == t["want"], t
        assert t["runs"] > 0 and t["skipped"] <= t["runs"]                   # counters are cumulative over the three generations
    last = v["trials"][-1]
    assert last["cancel_notes"] <= sum(t["n_runs_canceled"] for t in v["trials"])   # one shared note per cancelled run; only speculative runs are cancelled on the wire (speculative.cpp:1211)
    if tgt != dft:
        assert last["cancel_notes"] > 0
    assert last["skipped"] <= last["cancel_notes"]                           # a stage only ever skips what the head cancelled
    assert any(t["n_runs_spec"] > 0 for t in v["trials"])
    if tgt == dft:
        assert any(t["n_accept_clean"] > 0 for t in v["trials"])
