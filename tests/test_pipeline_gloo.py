"""N > 1 path on CPU: world_size 2, 3 and 4 (torchrun + gloo for the rendezvous; the runs travel through the library's shared-memory transport
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
                                                ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, 3),       # 3: head + two stages (header and activation hop between stages)
                                                ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, 4),       # 4: one layer per stage — a run crosses two stage-to-stage hops; skips propagate through two forwards
                                                ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, -2)])     # -2: world 2 with run coalescing in the stage's serve loop (the stage is the whole target)
def test_head_and_stage_over_gloo_are_lossless(tmp_path, tgt, dft, seed, world):
    """rank 0: C++ driver + proxy target context; rank 1 (and 2): pipeinfer_transport_serve.  The accepted tokens must equal the reference
    greedy continuation, every started run must come back (computed or skipped after a cancel) and cancels must reach the stages."""
    out = tmp_path / "verdict.json"
    env = dict(os.environ, OMP_NUM_THREADS="2", PYTHONPATH=ROOT)
    coalesce = world < 0
    world = abs(world)
    if coalesce:                                                                 # the serve loop gathers the runs waiting in its ring, applies the conflict rule, answers in start order
        env.update(PIPEINFER_STAGE_COALESCE="16", PI_TEST_STAGE_LATENCY_MS="3")  # (a slow stage: runs do queue up behind the one being computed)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(ROOT, "tests", "pipeline_worker.py"), str(out), tgt, dft, str(seed)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-3000:]
    v = json.loads(out.read_text())
    assert len(v["trials"]) == 3
    for t in v["trials"]:
        assert t["out"] == t["want"], t
        assert t["runs"] > 0 and t["skipped"] <= t["runs"]                   # counters are cumulative over the three generations
        assert t["received"] == t["runs"], t                                   # conservation at the head: every started run came back (computed or as a skipped header)
    last = v["trials"][-1]
    assert len(v["stages"]) == world - 1
    if coalesce:
        print("stage coalescing over gloo:", {k: v["stages"][0][k] for k in ("runs", "skipped", "passes_of_several_runs")})
        assert v["stages"][0]["passes_of_several_runs"] > 0, v["stages"][0]
    for st in v["stages"]:
        # conservation at every stage: each run the head started passed through it exactly once, as a computed run or as a skipped header (a run cut short inside the
        # stage counts as skipped); nothing lost, nothing duplicated
        assert st["received"] == last["runs"] and st["runs"] + st["skipped"] == st["received"] and st["cut_short"] <= st["skipped"], (st, last)
    # what the head saw come back as skipped is what the LAST stage forwarded as skipped (a skip propagates down the pipeline, never up)
    assert last["skipped"] == v["stages"][-1]["skipped"], (last, v["stages"])
    for a, b in zip(v["stages"], v["stages"][1:]):
        assert a["skipped"] <= b["skipped"]
    assert last["cancel_notes"] <= sum(t["n_runs_canceled"] for t in v["trials"])   # one shared note per cancelled run; only speculative runs are cancelled on the wire (speculative.cpp:1211)
    if tgt != dft:
        assert last["cancel_notes"] > 0
    assert last["skipped"] <= last["cancel_notes"]                           # a stage only ever skips what the head cancelled
    assert any(t["n_runs_spec"] > 0 for t in v["trials"])
    if tgt == dft:
        assert any(t["n_accept_clean"] > 0 for t in v["trials"])


def test_bench_spawns_its_ranks_and_refuses_to_share_a_device():
    """`python bench.py --gpus 2` with no launcher around it and no RANK / WORLD_SIZE in the environment (how the round driver calls it): the parent process starts the
    ranks itself as children (torch.distributed.run) — here, without a GPU, both ranks then refuse loudly to double up on one device, the parent forwards the failure as
    a non-zero exit code and prints no result line.  (The same command on a GPU box, three ranks sharing the one GPU by explicit override: tests/test_gpu_engine.py.)"""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "PIPEINFER_ALLOW_SHARED_GPU")}
    env.update(PYTHONPATH=ROOT, OMP_NUM_THREADS="2")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload", "tiny", "--steps", "1", "--warmup", "0", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert "refusing to share devices" in r.stderr, r.stderr[-2000:]
    assert "[bench rank 0]" in r.stderr and "[bench rank 1]" in r.stderr           # two ranks were really started
    assert not r.stdout.strip()
