"""-m gpu: the engine behind the llama.h boundary against golden logits produced by the REAL reference kernels
(tests/golden/*.npz, made by tests/golden/make_golden.py), on the tiny quantised Llama / Falcon fixtures:
  * every decode of a PipeInfer-style KV script (multi-seq prompt, partition fork, tree batch, accept/prune, cell reuse)
  * greedy continuation == the reference's (the lossless-speculation ground truth)
  * the PipeInfer driver's accepted tokens == that same greedy continuation, for several draft models and settings
Tolerance on logits: 1e-3 absolute (BASELINE.json north_star); measured spread is printed."""
import ctypes as C
import os

import numpy as np
import pytest

from tests import scenarios as S

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SEEDS = {"tiny_llama_q4_k_m": 11, "tiny_llama_q4_0": 12, "tiny_llama_b_q4_k_m": 13, "tiny_falcon_q4_0": 14, "tiny_falcon40_q4_k_m": 15, "tiny_llama_q3_k_m": 16, "tiny_llama_q2_k": 17, "tiny_llama_q5_1": 18, "tiny_llama_q5_0": 19, "tiny_llama_q4_1": 20, "tiny_llama_hd80_q4_0": 29,
         # the reference's other graph builders (llama.cpp:3985-4939): goldens from its OWN loader + build_* (tests/golden/make_golden.py archs)
         "tiny_baichuan7_q4_0": 21, "tiny_baichuan13_q4_0": 22, "tiny_starcoder_q4_0": 23, "tiny_refact_q4_0": 24, "tiny_bloom_q4_0": 25, "tiny_mpt_q4_k_m": 26, "tiny_stablelm_q4_k_m": 27, "tiny_persimmon_q4_k_m": 28}
LOGIT_TOL = 1e-3


@pytest.fixture(scope="module")
def LL():
    from pipeinfer_amd import llama as LL
    from tests import gpu_util as U
    U.init()
    return LL


class EngineBackend:
    def __init__(self, LL, stem, n_ctx=128):
        self.model = LL.Model(os.path.join(GOLD, stem + ".gguf"))
        self.ctx = LL.Context(self.model, n_ctx=n_ctx, n_batch=64)

    def decode(self, tokens, pos, seq_ids):
        return self.ctx.decode(tokens, pos, seq_ids)

    def seq_rm(self, *a):
        self.ctx.seq_rm(*a)

    def seq_cp(self, *a):
        self.ctx.seq_cp(*a)

    def seq_keep(self, *a):
        self.ctx.seq_keep(*a)

    def close(self):
        self.ctx.free(); self.model.free()


@pytest.mark.parametrize("stem", list(SEEDS))
def test_scenario_logits_match_reference(LL, stem):
    g = np.load(os.path.join(GOLD, stem + ".npz"))
    b = EngineBackend(LL, stem)
    outs = S.run_scenario(S.scenario_spec_like(b.model.n_vocab, seed=SEEDS[stem]), b)
    worst = 0.0
    for i, lg in enumerate(outs):
        ref = g[f"spec_like_{i}"]
        err = float(np.abs(lg - ref).max())
        worst = max(worst, err)
        assert err <= LOGIT_TOL, (stem, i, err)
        assert np.array_equal(lg.argmax(1), ref.argmax(1)), (stem, i)
    print(f"{stem}: max |logit - reference| over the KV script = {worst:.2e}")
    b.close()


@pytest.mark.parametrize("stem", list(SEEDS))
def test_greedy_continuation_matches_reference(LL, stem):
    g = np.load(os.path.join(GOLD, stem + ".npz"))
    b = EngineBackend(LL, stem)
    prompt, n_gen = S.scenario_greedy(b.model.n_vocab, seed=SEEDS[stem])
    lg = b.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
    assert np.abs(lg - g["greedy_prompt_logits"]).max() <= LOGIT_TOL
    toks, cur = [], int(np.argmax(lg[-1]))
    ids, vals, _ = b.ctx.topk(len(prompt) - 1)
    assert ids[0] == cur                                   # device top-k agrees with the host argmax
    for i in range(n_gen):
        toks.append(cur)
        lg = b.decode([cur], [len(prompt) + i], [[0]])
        cur = int(np.argmax(lg[0]))
    assert toks == list(g["greedy_tokens"]), stem
    assert np.abs(lg - g["greedy_last_logits"]).max() <= LOGIT_TOL
    b.close()


def test_async_decode_and_iprobe(LL):
    """start/finish split: several runs in flight on the stream, results identical to synchronous decodes"""
    import time
    b = EngineBackend(LL, "tiny_llama_q4_k_m")
    rng = np.random.default_rng(0)
    toks = [int(x) for x in rng.integers(3, 512, 12)]
    sync = [b.decode([t], [i], [[0]]) for i, t in enumerate(toks)]
    b.ctx.kv_clear()
    tickets = [b.ctx.start([t], [i], [[0]], batch_id=i + 1) for i, t in enumerate(toks)]
    t0 = time.time()
    while not b.ctx.iprobe():
        assert time.time() - t0 < 10
    outs = [b.ctx.finish(t) for t in tickets]
    assert not b.ctx.iprobe()                               # nothing in flight any more
    for a, c in zip(sync, outs):
        assert np.array_equal(a, c)
    b.close()


def greedy_reference(stem, n_prompt, n_gen):
    g = np.load(os.path.join(GOLD, stem + ".npz"))
    prompt, _ = S.scenario_greedy(256 if "baichuan" in stem else 512, seed=SEEDS[stem], n_prompt=n_prompt)
    return prompt, list(g["greedy_tokens"][:n_gen])


@pytest.mark.parametrize("tgt,dft", [("tiny_llama_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_llama_q4_k_m", "tiny_llama_q4_0"),
                                      ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_falcon40_q4_k_m", "tiny_falcon_q4_0"),
                                      ("tiny_llama_q3_k_m", "tiny_llama_q4_0"), ("tiny_llama_q2_k", "tiny_llama_q3_k_m"),
                                      # the generic-block architectures (512-token vocabularies; Baichuan's is 256).  Targets WITHOUT ALiBi only: the reference adds i * m_k with i the
                                      # CELL index (ggml.c:10789, "K-shift is likely not working" llama.cpp:3737), so under PipeInfer's scattered cells an ALiBi target's logits depend on
                                      # where its cells lie and "speculative == plain greedy" is not a property of the reference either; those targets are pinned run by run against the
                                      # oracle on the same cells instead (test_every_run_of_a_speculative_generation_equals_the_oracle).  As DRAFTS they only propose.
                                      ("tiny_starcoder_q4_0", "tiny_mpt_q4_k_m"), ("tiny_stablelm_q4_k_m", "tiny_bloom_q4_0"), ("tiny_starcoder_q4_0", "tiny_stablelm_q4_k_m"),
                                      ("tiny_stablelm_q4_k_m", "tiny_refact_q4_0"), ("tiny_baichuan7_q4_0", "tiny_baichuan13_q4_0"), ("tiny_persimmon_q4_k_m", "tiny_starcoder_q4_0")])
@pytest.mark.parametrize("n_draft,n_parallel,p_split,min_spec", [(3, 3, 0.9, 0), (3, 3, 0.9, 1), (8, 1, 0.9, 2), (4, 3, 0.05, 3)])
def test_pipeinfer_driver_is_lossless(LL, tgt, dft, n_draft, n_parallel, p_split, min_spec):
    """accepted-token sequence of the asynchronous speculative driver == the reference target's greedy continuation.
    min_spec > 0 forces speculation iterations even when the (tiny, fast) target is already back, so that the draft tree,
    accept/prune, continuous speculation and cancel paths all run."""
    prompt, want = greedy_reference(tgt, 8, 24)
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
    out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=n_draft, n_parallel=n_parallel, p_accept=0.05, p_split=p_split, ignore_eos=0,
                              min_spec_iters=min_spec)
    if len(out) < 24:                                       # (ignore_eos = 0: a generation whose greedy continuation reaches the end-of-sequence token stops there, speculative.cpp:485-488)
        assert out and out[-1] == mt.lib.llama_token_eos(mt.h) and out == want[:len(out)], (out, want, res)
    else:
        assert out[:24] == want, (out, want, res)
        assert res["n_predict"] >= 24
    if tgt == dft and min_spec > 0:
        assert res["n_accept_clean"] > 0                    # identical draft: speculation is actually being accepted
    print(tgt, dft, n_draft, n_parallel, {k: res[k] for k in ("n_drafted", "n_accept", "n_accept_clean", "n_drafted_tokens", "n_runs_spec", "n_runs_canceled")})
    for c in (ct, cd):
        c.free()
    for m in (mt, md):
        m.free()


@pytest.mark.parametrize("window,pause,min_spec", [(2, 3, 0), (-3, 2, 1), (-2, 5, 2), (1, 1, 0)])
def test_adaptive_draft_pause_is_lossless_and_the_draft_catches_up(LL, window, pause, min_spec):
    """pipeinfer.h adaptive_window / adaptive_pause: the draft sits out tokens while speculation is not happening and is caught up with
    one batched decode.  The output must stay the target's greedy continuation; with an identical draft model and forced pauses
    (window < 0) in the middle of continuous speculation, drafted tokens must still be accepted after a catch-up — a draft KV cache
    left out of step would mispredict."""
    tgt = dft = "tiny_llama_q4_k_m"
    prompt, want = greedy_reference(tgt, 8, 24)
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    res = {}
    for name, kw in (("plain", {}), ("paused", dict(adaptive_window=window, adaptive_pause=pause))):
        ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
        out, res[name] = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=3, n_parallel=3, p_accept=0.05, p_split=0.9, ignore_eos=0, min_spec_iters=min_spec, **kw)
        assert out[:24] == want, (name, out, want, res[name])
        ct.free(); cd.free()
    assert res["plain"]["n_draft_paused"] == 0 and res["plain"]["n_draft_catchups"] == 0
    assert res["paused"]["n_draft_paused"] > 0
    if window < 0:
        assert res["paused"]["n_draft_catchups"] > 0
        rate = lambda r: r["n_accept_clean"] / max(1, r["n_drafted_tokens"])
        assert res["paused"]["n_accept_clean"] > 0 and rate(res["paused"]) >= 0.5 * rate(res["plain"]), res
    mt.free(); md.free()


def test_cpp_llama_h_caller_reproduces_the_reference(tmp_path):
    """examples/speculative_min.cpp — a C++11 program that sees nothing but include/llama.h (+ pipeinfer.h) and the shared library, in
    the style of the reference's examples/speculative: llama_batch_init / llama_decode / llama_get_logits_ith / llama_sample_token_greedy,
    then the asynchronous speculative driver.  Both token lines must be the reference's greedy continuation of the fixture."""
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "speculative_min")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "speculative_min.cpp"),
                    "-L" + os.path.join(root, "pipeinfer_amd"), "-lpipeinfer_amd", "-Wl,-rpath," + os.path.join(root, "pipeinfer_amd"), "-o", exe], check=True)
    tgt, dft = "tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"
    prompt, want = greedy_reference(tgt, 8, 24)
    r = subprocess.run([exe, os.path.join(GOLD, tgt + ".gguf"), os.path.join(GOLD, dft + ".gguf"), "24"] + [str(t) for t in prompt],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = {l.split(":")[0]: l.split(":")[1].split() for l in r.stdout.splitlines() if ":" in l}
    assert [int(t) for t in lines["plain"]] == want, (lines, want)
    assert [int(t) for t in lines["spec"]] == want, (lines, want)
    stats = [l for l in r.stdout.splitlines() if l.startswith("spec_runs")][0].split()
    assert int(stats[1]) > 0                                      # speculation really ran


def test_cli_driver_with_the_reference_flags(tmp_path):
    """examples/speculative_amd.cpp: the reference's command line (-m -md -n --draft -np -pa -ps --ignore-eos ...) and closing report on top of
    llama.h; the generated ids must be the golden continuation, results.csv gets the reference's four columns"""
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe, csv = str(tmp_path / "speculative_amd"), str(tmp_path / "results.csv")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "speculative_amd.cpp"),
                    "-L" + os.path.join(root, "pipeinfer_amd"), "-lpipeinfer_amd", "-Wl,-rpath," + os.path.join(root, "pipeinfer_amd"), "-o", exe], check=True)
    tgt, dft = "tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"
    prompt, want = greedy_reference(tgt, 8, 24)
    r = subprocess.run([exe, "-m", os.path.join(GOLD, tgt + ".gguf"), "-md", os.path.join(GOLD, dft + ".gguf"), "--prompt-tokens", ",".join(str(t) for t in prompt),
                        "-n", "23", "--draft", "4", "-np", "2", "-pa", "0.05", "-ps", "0.9", "-c", "256", "-b", "64", "--results", csv],
                       capture_output=True, text=True, errors="replace", timeout=300)        # (the fixture's token pieces are not valid UTF-8)
    assert r.returncode == 0, r.stderr[-2000:]
    toks = [l for l in r.stdout.splitlines() if l.startswith("tokens:")][0].split()[1:]
    assert [int(t) for t in toks][:24] == want, (toks, want)
    for key in ("encoded", "decoded", "n_draft   = 4", "n_predict =", "n_drafted =", "n_accept  =", "accept    ="):
        assert any(l.startswith(key) for l in r.stdout.splitlines()), (key, r.stdout[-1500:])
    cols = open(csv).read().strip().split(",")
    assert len(cols) == 4 and all(float(c) > 0 for c in cols)
    # a text prompt goes through llama_tokenize (SPM vocabulary of the fixture); only that it runs and reports is checked here
    r2 = subprocess.run([exe, "-m", os.path.join(GOLD, tgt + ".gguf"), "-md", os.path.join(GOLD, dft + ".gguf"), "-p", "hello world again", "-n", "8", "-c", "256", "-b", "64",
                         "--ignore-eos", "--target-only"], capture_output=True, text=True, errors="replace", timeout=300)
    assert r2.returncode == 0 and "decoded" in r2.stdout, r2.stderr[-2000:]


@pytest.mark.parametrize("split,host_rings", [("0.34,0.66/1.0", False), ("1.0", False), ("0.34,0.33,0.33", True)])
def test_cli_driver_layer_split_across_processes(tmp_path, split, host_rings):
    """--mpi-layer-split: the same command line started once per rank (the reference: once per MPI rank, speculative.cpp:116-139); rank 0 drives and
    drafts, the other ranks serve the target's layer ranges through pipeinfer_transport_*.  All ranks share this box's one GPU.  The ids must be the
    golden continuation, speculation must have run, and every stage must exit cleanly after the head's shutdown."""
    import shutil, subprocess
    if shutil.which("g++") is None:
        pytest.skip("no host C++ compiler on this box")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "speculative_amd")
    subprocess.run(["g++", "-std=c++11", "-O2", "-Wall", "-Werror", "-I" + os.path.join(root, "include"), os.path.join(root, "examples", "speculative_amd.cpp"),
                    "-L" + os.path.join(root, "pipeinfer_amd"), "-lpipeinfer_amd", "-Wl,-rpath," + os.path.join(root, "pipeinfer_amd"), "-o", exe], check=True)
    tgt, dft = "tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"
    prompt, want = greedy_reference(tgt, 8, 24)
    world = len(split.split("/")[0].split(",")) + 1
    base = [exe, "-m", os.path.join(GOLD, tgt + ".gguf"), "-md", os.path.join(GOLD, dft + ".gguf"), "--prompt-tokens", ",".join(str(t) for t in prompt),
            "-n", "23", "--draft", "4", "-np", "2", "-pa", "0.05", "-ps", "0.9", "-c", "256", "-b", "64", "--mpi-layer-split", split, "--device", "0"]
    env = dict(os.environ, PIPEINFER_JOB=f"pi-cli-{os.getpid()}-{world}", PIPEINFER_WORLD=str(world))
    if host_rings:
        env["PIPEINFER_TRANSPORT_HOST"] = "1"
    procs = [subprocess.Popen(base, env=dict(env, PIPEINFER_RANK=str(r)), stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, errors="replace") for r in range(world)]
    outs = []
    try:
        for p in procs:
            outs.append(p.communicate(timeout=300))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][1][-2000:])
    toks = [l for l in outs[0][0].splitlines() if l.startswith("tokens:")][0].split()[1:]
    assert [int(t) for t in toks][:24] == want, (toks, want)
    assert "host rings" in outs[0][1] if host_rings else "peer stores" in outs[0][1], outs[0][1][-500:]
    assert int([l for l in outs[0][0].splitlines() if l.startswith("n_drafted =")][0].split()[-1]) > 0
    for r in range(1, world):
        assert f"rank {r} serves layers" in outs[r][1] and " runs, " in outs[r][1], outs[r][1][-500:]
    # the wrong number of processes for the split is refused before anything is opened
    bad = subprocess.run(base, env=dict(env, PIPEINFER_RANK="0", PIPEINFER_WORLD=str(world + 1)), capture_output=True, text=True, timeout=60)
    assert bad.returncode == 2 and "start" in bad.stderr


@pytest.mark.parametrize("spec,ftype,ftype_id", [("tiny-llama-b", "q4_k_m", 15), ("tiny-llama", "q3_k_m", 12), ("tiny-falcon40", "q5_k_m", 17), ("tiny-llama", "q5_0", 8)])
def test_engine_runs_the_files_its_quantiser_writes(LL, tmp_path, spec, ftype, ftype_id):
    """llama_model_quantize (csrc/engine/quantize.cpp) turns an fp32 model into the mixture the reference would write; the engine then decodes that
    file and gives the oracle's logits bit for bit (prompt batch and single tokens)."""
    import ctypes as C
    from pipeinfer_amd.tools import synth
    from tests import oracle_lib as O
    from tests.test_quantize import QuantizeParams
    src, dst = str(tmp_path / "f32.gguf"), str(tmp_path / "q.gguf")
    synth.write_model(src, spec, "f32", seed=9, structure="random", std=0.05)
    lib = LL.api()
    lib.llama_model_quantize.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(QuantizeParams)]
    lib.llama_model_quantize_default_params.restype = QuantizeParams
    p = lib.llama_model_quantize_default_params()
    p.ftype, p.nthread = ftype_id, 2
    assert lib.llama_model_quantize(src.encode(), dst.encode(), C.byref(p)) == 0
    m = LL.Model(dst)
    c = LL.Context(m, n_ctx=128, n_batch=32)
    o = O.OracleModel(dst, n_ctx=128, max_seq=2)
    toks = [3, 17, 400, 250, 99, 12, 7]
    got = c.decode(toks, list(range(len(toks))), [[0]] * len(toks))
    want = o.decode(toks, list(range(len(toks))), [[0]] * len(toks))
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
    for i in range(6):
        t = int(np.argmax(got[-1]))
        got = c.decode([t], [len(toks) + i], [[0]])
        want = o.decode([t], [len(toks) + i], [[0]])
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), i
    o.close(); c.free(); m.free()


def test_pipeinfer_driver_lossless_under_timing_jitter(LL):
    """the driver's control flow depends on which run finishes first, so repeat one configuration with fresh contexts
    (this caught target and draft contexts racing on a shared top-k scratch buffer)."""
    tgt, dft = "tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"
    prompt, want = greedy_reference(tgt, 8, 24)
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    for trial in range(25):
        ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
        out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=3, n_parallel=3, p_accept=0.05, p_split=0.9, ignore_eos=0, min_spec_iters=trial % 3)
        assert out[:24] == want, (trial, out, want, res)
        ct.free(); cd.free()
    mt.free(); md.free()


def test_target_only_mode_matches(LL):
    prompt, want = greedy_reference("tiny_llama_q4_k_m", 8, 24)
    mt = LL.Model(os.path.join(GOLD, "tiny_llama_q4_k_m.gguf"))
    ct = LL.Context(mt, n_ctx=256, n_batch=64)
    out, res = LL.speculative(ct, None, prompt, n_predict=23, mode=2, ignore_eos=0)
    assert out[:24] == want
    ct.free(); mt.free()



@pytest.mark.parametrize("stem,ranges", [("tiny_llama_b_q4_k_m", "0-1,1-3"), ("tiny_llama_q4_k_m", "0-2"), ("tiny_falcon40_q4_k_m", "0-1,1-2"),
                                         # the generic block: learned positions / the embeddings' LayerNorm belong to the FIRST stage only, the ALiBi slopes to every stage
                                         ("tiny_starcoder_q4_0", "0-1,1-2"), ("tiny_bloom_q4_0", "0-1,1-2"), ("tiny_mpt_q4_k_m", "0-1,1-2"), ("tiny_stablelm_q4_k_m", "0-1,1-2"), ("tiny_persimmon_q4_k_m", "0-1,1-2")])
def test_pipeline_stages_reproduce_the_single_process_path(stem, ranges):
    """head proxy context + pipeinfer_stage_run on layer ranges == the whole model in one context (tests/pipeline_gpu_worker.py).
    Runs in its own process (several contexts of several models; keeps the parent's device state clean)."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "pipeline_gpu_worker.py"), stem, ranges], capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "PIPELINE-OK" in r.stdout


@pytest.mark.parametrize("tgt,dft,seed,ranges,host_rings", [("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-1,1-3", False), ("tiny_llama_q4_k_m", "tiny_llama_q4_k_m", 11, "0-1,1-2", False),
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-1,1-3", True),
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-1,1-3", "host-headers"),   # peer stores, headers published by the host (round 3's hop)
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-1,1-3", "probe-fail"),     # the real-pattern hop probe fails on a rank: every rank falls back to host-published headers
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-1,1-3", "coarse-ring"),    # PIPEINFER_RING_ALLOC=coarse: the ring as plain hipMalloc memory (rounds 1-4)
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-1,1-3", "uncached-ring"),
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-1,1-2,2-3", False),       # three stages: a middle stage waits AND publishes
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-3", False),      # --gpus 2: one stage that is first and last
                                                            ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 13, "0-3", "coalesce")]) # ... and carries the runs waiting in its ring in ONE pass (run coalescing on the stage)
def test_three_processes_on_one_gpu_through_the_library_transport(tmp_path, tgt, dft, seed, ranges, host_rings):
    """head + two HIP stages as three processes on this one GPU (tests/transport_gpu_worker.py): run headers, metadata, results and cancel
    notes through the shared segment, activations as peer stores into the next stage's HBM ring (hipIpc).  The accepted tokens must be
    the reference greedy continuation; every started run comes back; stages only skip what the head cancelled.
    host_rings: the fall-back every rank takes together when the peer mapping is not usable (PIPEINFER_TRANSPORT_HOST forces it): the HIP stages then copy
    their activations through rings in the shared segment."""
    import json, socket, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0)); port = sk.getsockname()[1]
    out = tmp_path / "verdict.json"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ranges.count(',') + 2}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.join(root, "tests", "transport_gpu_worker.py"), str(out), tgt, dft, str(seed), ranges]
    env = dict(os.environ, PYTHONPATH=root, OMP_NUM_THREADS="2", HSA_ENABLE_IPC_MODE_LEGACY="0")
    mode = host_rings
    host_headers = mode in ("host-headers", "probe-fail")
    host_rings = mode is True
    if host_rings:
        env["PIPEINFER_TRANSPORT_HOST"] = "1"
    if mode == "host-headers":
        env["PIPEINFER_TRANSPORT_FLAGS"] = "0"
    if mode == "probe-fail":
        env["PIPEINFER_TEST_HOP_PROBE_FAIL"] = "1"
    if mode in ("coarse-ring", "uncached-ring"):
        env["PIPEINFER_RING_ALLOC"] = mode.split("-")[0]
    if mode == "coalesce":
        env["PIPEINFER_TEST_STAGE_COALESCE"] = "16"
        env["PIPEINFER_TEST_STAGE_LATENCY_US"] = "2000"      # the stage picks its input up late: the head's runs queue behind one another whatever the GPU's speed (without it 1 start in ~10 saw no two headers waiting)
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-4000:]
    v = json.loads(out.read_text())
    print("transport:", {k: v[k] for k in ("device_mode", "stages")})
    assert v["device_mode"] == (not host_rings), "the peer-store path was not usable: " + r.stderr[-2000:]
    for t in v["trials"]:
        assert t["out"] == t["want"], t
    last = v["trials"][-1]
    assert last["skipped"] <= last["cancel_notes"] <= sum(t["n_runs_canceled"] for t in v["trials"])
    assert any(t["n_runs_spec"] > 0 for t in v["trials"])
    assert all(s["device_mode"] == (not host_rings) for s in v["stages"]) and all(s["runs"] > 0 for s in v["stages"])
    if mode == "coalesce":                                                                 # runs did travel together, and the tokens above are still the reference's
        s0 = v["stages"][0]
        print("stage coalescing:", {k: s0[k] for k in ("n_passes", "n_pass_runs", "n_coalesced_passes", "runs", "skipped")})
        assert s0["n_coalesced_passes"] > 0 and s0["n_pass_runs"] > s0["n_coalesced_passes"], s0
    # conservation: every run the head started came back, and passed every stage exactly once (computed, or as a skipped header)
    assert last["received"] == last["runs"]
    for s in v["stages"]:
        assert s["received"] == last["runs"] and s["runs"] + s["skipped"] == s["received"], (s, last)
    # the hop: one in-stream store into the consumer's ring slot (or the logits into the result slot) + one header; GPU time of the store per run, host time from the end
    # of a run to its header being visible downstream per run — both far below the 50 us a host-mediated hop costs
    for s in v["stages"]:
        hop_gpu_us, hop_host_us = 1e3 * s["hop_gpu_ms"] / s["runs"], 1e3 * s["send_ms"] / s["received"]
        print(f"stage {s['rank']}: store of the activations / logits {hop_gpu_us:.1f} us of GPU time per run, header published {hop_host_us:.1f} us after the run; waited {s['wait_out_ms']:.2f} ms in all for a free slot downstream")
        if not host_rings:
            assert hop_gpu_us < 50.0 and hop_host_us < 50.0, s
    # device-side arrival flags (the default in device mode): every run is announced downstream before it is computed and the consuming stage's run is enqueued behind a
    # device-side wait; the flag written by the producing stage's STREAM (behind its peer store) is seen by the waiting kernel within microseconds — no host thread on the
    # critical path of a stage boundary (the reference: blocking MPI_Recv, ggml-mpi.c:451-487)
    if mode == "probe-fail":
        assert "hop probe FAILED" in r.stderr and "Headers are published by the host" in r.stderr
    for s in v["stages"]:
        # decided and logged at the transport's set-up, per rank: how the hop's store is done and what memory the stage's own ring is (round 5: fine-grained by default)
        assert s["hop_store_kernel"] == (0 if host_rings else 1), s
        if s["rank"] >= 2 and not host_rings:
            assert s["ring_alloc"] == {"coarse-ring": 1, "uncached-ring": 3}.get(mode, 2), s
    if not host_rings and not host_headers and len(v["stages"]) >= 2:                      # (a stage boundary exists: the real-pattern probe ran on it)
        assert "hop probe ok" in r.stderr, r.stderr[-1500:]
    for s in v["stages"]:
        assert bool(s["device_flags"]) == (not host_rings and not host_headers), s
        if s["device_flags"] and s["rank"] >= 2:
            assert s["flag_runs"] > 0, s
            seen_us = s["flag_seen_us"] / s["flag_runs"]
            print(f"stage {s['rank']}: arrival flag seen {seen_us:.2f} us after the producing stream stamped it (mean of {s['flag_runs']} runs); {s['flag_wait_gpu_ms']:.1f} ms of GPU time waiting for input")
            assert seen_us < 100.0, s       # (three processes time-share ONE device here: the number is an upper bound for stages that own their GPU)


@pytest.mark.parametrize("tgt,dft,n_layer", [("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m", 3), ("tiny_llama_q4_k_m", "tiny_llama_q4_k_m", 2), ("tiny_falcon40_q4_k_m", "tiny_falcon_q4_0", 2),
                                             ("tiny_mpt_q4_k_m", "tiny_starcoder_q4_0", 2), ("tiny_bloom_q4_0", "tiny_stablelm_q4_k_m", 2), ("tiny_starcoder_q4_0", "tiny_refact_q4_0", 2),
                                             ("tiny_stablelm_q4_k_m", "tiny_mpt_q4_k_m", 2), ("tiny_baichuan13_q4_0", "tiny_baichuan7_q4_0", 40),
                                             ("tiny_persimmon_q4_k_m", "tiny_stablelm_q4_k_m", 2), ("tiny_starcoder_q4_0", "tiny_persimmon_q4_k_m", 2)])
def test_every_run_of_a_speculative_generation_equals_the_oracle(tgt, dft, n_layer):
    """the PipeInfer driver runs with the target behind a proxy context; each run's metadata block (tokens, positions, cells,
    visibility mask) is executed by the HIP stage AND by the oracle's forward pass on the same cells: logits must agree bit for
    bit for every run of the generation — accepted, rejected or cancelled (tests/trace_compare_worker.py).  Runs with an
    artificial run latency so that several speculative runs are in flight and get adopted."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = ":".join([os.path.join(GOLD, tgt + ".gguf"), os.path.join(GOLD, dft + ".gguf"), str(n_layer)])
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "trace_compare_worker.py"), spec, "0.003", "40"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, PYTHONPATH=root, ORACLE_THREADS="4"))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "TRACE-PARITY OK" in r.stdout, r.stdout[-2000:]


# (the second set: round 4's generation lengths — every accept / cancel / adopt path many times over; behind PIPEINFER_SLOW_TESTS=1, tests/conftest.py)
@pytest.mark.parametrize("workload,n_predict,n_draft,n_parallel,p_accept", [("7b", 12, 3, 3, 0.1), ("70b-l4", 32, 3, 3, 0.1), ("falcon40b-l4", 24, 3, 3, 0.002), ("goliath-l4", 16, 8, 2, 0.002)] + [
    pytest.param(*a, marks=pytest.mark.slow) for a in (("7b", 48, 3, 3, 0.1), ("70b-l4", 48, 3, 3, 0.1), ("falcon40b-l4", 40, 3, 3, 0.002), ("goliath-l4", 40, 8, 2, 0.002))])
def test_every_run_equals_the_oracle_at_full_size_shapes(workload, n_predict, n_draft, n_parallel, p_accept):
    """the same per-run comparison at the shapes of BASELINE's configurations: the whole Llama-2-7B Q4_K_M (K = 4096 / 11008, 32 layers), and
    Llama-2-70B shapes (GQA 64/8, K = 8192 / 28672: the tile-per-wave decode instances, the matrix-core kernel on verify batches and prompt
    chunks) cut to four layers so that the CPU oracle keeps up; BASELINE configs[3]: Falcon-40B Q5_K_M shapes (two-norm block, fused QKV of 9216 rows,
    GQA 128/8 with 64-wide heads, K = 8192 / 32768, Q8_0 lm_head of 65024 rows) with the whole Falcon-7B Q4_0 as the draft (K = 4544 / 18176, MQA);
    configs[4]: Goliath-120B Q4_K_M shapes with a Llama-2-7B-shaped draft and depth-8 drafts (8-token verify batches).  (The synthetic Falcon and 7B-shaped drafts spread their
    probability widely, hence the low p_accept there.)  Every run the driver launches — prompt batches, single tokens, speculative
    batches, accepted or cancelled — must give the oracle's logits bit for bit on the cells it was given."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "trace_compare_worker.py"), workload, "0.02", str(n_predict), str(n_draft), str(n_parallel), str(p_accept)], capture_output=True, text=True, timeout=1500,
                       env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    assert "TRACE-PARITY OK" in r.stdout, r.stdout[-2000:]
    spec_runs = int(r.stdout.split("TRACE-PARITY OK spec_runs")[1].split()[0])
    accepted = int(r.stdout.split("TRACE-PARITY OK spec_runs")[1].split("accepted")[1].split()[0])
    # the short generations must still walk the paths the long ones were there for: speculative batches among the compared runs, and drafted tokens that were adopted
    assert spec_runs >= 2 and accepted >= 1, r.stdout[-1500:]


def test_full_size_properties_llama_7b(LL, tmp_path_factory):
    """BASELINE's single-GPU target at full size (Llama-2-7B shapes, Q4_K_M mixture, 4.1 GB of synthetic weights): properties that
    need no oracle run — a k-token batch gives the same bits as the same tokens decoded one by one (cells contiguous in both), a
    repeated call gives the same bits (HIP-graph replay included), and removing a speculative branch (seq_rm) restores the logits of
    the surviving sequence."""
    import bench as B
    cache = os.environ.get("PIPEINFER_CACHE", "/tmp/pipeinfer_models")
    path = B.ensure_model(cache, "llama-2-7b", "q4_k_m", 0x5EED, 1.0, 0.8)
    m = LL.Model(path)
    rng = np.random.default_rng(5)
    prompt = [int(x) for x in rng.integers(3, m.n_vocab, 48)]
    extra = [int(x) for x in rng.integers(3, m.n_vocab, 8)]
    a, b = LL.Context(m, n_ctx=1024, n_batch=512), LL.Context(m, n_ctx=1024, n_batch=512)
    la, lb = a.decode(prompt, list(range(48)), [[0]] * 48), b.decode(prompt, list(range(48)), [[0]] * 48)
    assert np.array_equal(la.view(np.uint32), lb.view(np.uint32))                      # determinism, two contexts
    for k in (1, 3, 8):
        singles = [a.decode([extra[i]], [48 + i], [[0]])[0] for i in range(k)]
        batch = b.decode(extra[:k], [48 + i for i in range(k)], [[0]] * k)
        for i in range(k):
            assert np.array_equal(singles[i].view(np.uint32), batch[i].view(np.uint32)), (k, i)
        a.seq_rm(0, 48, -1); b.seq_rm(0, 48, -1)                                        # roll both back to the prompt
    base = a.decode([extra[0]], [48], [[0]])[0]                                         # cell 48, sequence 0
    a.seq_rm(0, 48, -1)
    a.seq_cp(0, 1, 0, 48)                                                               # fork: sequence 1 shares the prompt
    a.decode([extra[1], extra[2]], [48, 49], [[1], [1]])                               # a speculative branch on sequence 1
    a.seq_rm(1, 48, -1)                                                                 # rejected: prune it
    again = a.decode([extra[0]], [48], [[0]])[0]
    assert np.array_equal(base.view(np.uint32), again.view(np.uint32))
    for c in (a, b):
        c.free()
    m.free()


@pytest.mark.parametrize("tgt,dft", [("tiny_llama_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_falcon40_q4_k_m", "tiny_falcon_q4_0"),
                                      ("tiny_llama_q3_k_m", "tiny_llama_q4_0")])
@pytest.mark.parametrize("n_draft,n_parallel,p_split", [(3, 1, 0.9), (5, 3, 0.05), (8, 2, 0.3)])
def test_synchronous_speculation_mode_is_lossless(LL, tgt, dft, n_draft, n_parallel, p_split):
    """mode 1 = examples/speculative_orig restated (draft a tree, verify it in ONE target batch): accepted tokens == the reference target's
    greedy continuation; with an identical draft every drafted token is accepted (n_accept is a deterministic function of the two models)."""
    prompt, want = greedy_reference(tgt, 8, 24)
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
    out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=n_draft, n_parallel=n_parallel, p_accept=0.05, p_split=p_split, ignore_eos=0, mode=1)
    assert out[:24] == want, (out, want, res)
    assert res["n_predict"] >= 24 and res["n_drafted"] > 0
    if tgt == dft and n_parallel == 1:
        assert res["n_accept"] == res["n_drafted_tokens"], res        # the draft IS the target: nothing is ever rejected
    out2, res2 = None, None
    ct.kv_clear(); cd.kv_clear()
    out2, res2 = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=n_draft, n_parallel=n_parallel, p_accept=0.05, p_split=p_split, ignore_eos=0, mode=1)
    assert out2 == out and (res2["n_drafted"], res2["n_accept"]) == (res["n_drafted"], res["n_accept"])      # deterministic counters
    for c in (ct, cd):
        c.free()
    mt.free(); md.free()


def test_reference_driver_binary_runs_on_the_engine(LL, tmp_path):
    """oracle/_ref/speculative_ref = the reference's UNMODIFIED examples/speculative/speculative.cpp + common/*.cpp, compiled in the build
    container against include/llama.h and linked to libpipeinfer_amd.so (make -C oracle refdriver).  It must generate, on this GPU, exactly
    the text of the reference CPU kernels' greedy continuation of the same prompt."""
    import subprocess
    from tests import oracle_lib as O
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "oracle", "_ref", "speculative_ref")
    if not os.path.exists(exe):
        pytest.skip("oracle/_ref/speculative_ref was not built (needs the reference sources)")
    tgt, dft = "tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"
    text, n_gen = "hello world again and again", 24
    m = LL.Model(os.path.join(GOLD, tgt + ".gguf"))
    prompt = m.tokenize(text, add_bos=True, special=True)
    ref = O.RefModel(os.path.join(GOLD, tgt + ".gguf"), n_ctx=256, max_seq=2, n_threads=2) if O.have_ref() and O.ref() is not None else O.OracleModel(os.path.join(GOLD, tgt + ".gguf"), n_ctx=256, max_seq=2)
    lg = ref.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
    want = [int(np.argmax(lg[-1]))]
    for i in range(n_gen):
        want.append(int(np.argmax(ref.decode([want[-1]], [len(prompt) + i], [[0]])[0])))
    ref.close()
    want_text = b"".join(m.token_to_piece(t) for t in want[:n_gen + 1])
    m.free()
    r = subprocess.run([exe, "-m", os.path.join(GOLD, tgt + ".gguf"), "-md", os.path.join(GOLD, dft + ".gguf"), "-p", text, "-n", str(n_gen), "-c", "256", "-b", "64",
                        "--draft", "3", "-np", "3", "-pa", "0.05", "-ps", "0.9", "-pr", "0.0", "--temp", "0", "--repeat-last-n", "0", "--mpi-layer-split", "1.0/1.0",
                        "-t", "1", "-tb", "1"], capture_output=True, timeout=600, cwd=str(tmp_path))     # its log file goes to cwd
    assert r.returncode == 0, r.stderr[-3000:]
    assert r.stdout.startswith(want_text[:len(r.stdout)]) and len(r.stdout) >= len(want_text) - 8, (r.stdout, want_text)
    assert b"n_drafted" in r.stderr and b"n_accept" in r.stderr, r.stderr[-1500:]
    assert os.path.exists(os.path.join(str(tmp_path), "results.csv"))


def test_segmented_context_equals_the_single_graph_context(LL):
    """pipeinfer_context_set_segments: the stage captured as per-layer segments and launched by the feeder thread computes the same bits as
    the one-graph context (decode widths 1..17, so the VALU and the matrix-core mat-mul paths are both crossed)"""
    stem = "tiny_llama_b_q4_k_m"
    m = LL.Model(os.path.join(GOLD, stem + ".gguf"))
    a, b = LL.Context(m, n_ctx=256, n_batch=64), LL.Context(m, n_ctx=256, n_batch=64)
    b.set_segments(1)
    rng = np.random.default_rng(5)
    pos = 0
    for n in (9, 1, 1, 4, 17, 2, 1, 5):
        toks = [int(x) for x in rng.integers(3, m.n_vocab, n)]
        la = a.decode(toks, list(range(pos, pos + n)), [[0]] * n)
        lb = b.decode(toks, list(range(pos, pos + n)), [[0]] * n)
        assert np.array_equal(la.view(np.uint32), lb.view(np.uint32)), n
        pos += n
    assert b.stats()["n_segments"] == 8 * 3                  # three layers = three segments per call
    a.free(); b.free(); m.free()


@pytest.mark.parametrize("tgt,dft", [("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_falcon40_q4_k_m", "tiny_falcon_q4_0")])
@pytest.mark.parametrize("min_spec", [0, 1, 3])
def test_driver_lossless_with_a_segmented_target_and_a_priority_draft(LL, tgt, dft, min_spec):
    """the single-GPU arrangement of bench.py: the target is launched segment by segment by its feeder thread (cancelled speculative runs stop at
    the next segment boundary), the draft context has stream priority and is served between the target's segments"""
    prompt, want = greedy_reference(tgt, 8, 24)
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    for trial in range(4):
        ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
        ct.set_segments(1); cd.set_priority(1)
        out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=3, n_parallel=3, p_accept=0.05, p_split=0.9, ignore_eos=0, min_spec_iters=min_spec, verbose=int(os.environ.get("PIPEINFER_VERBOSE", "0")))
        assert out[:24] == want, (trial, out, want, res)
        ct.free(); cd.free()
    mt.free(); md.free()


@pytest.mark.timeout(120)
def test_a_priority_context_freed_with_unfinished_runs_releases_the_device_arbiter(LL):
    """the single-GPU arbiter lets a segmented (target) feeder launch its next segment only when no priority (draft) context has a run pending, or on a credit
    given when such a run finishes.  A priority context that is freed with runs it never finished must give its claim back: otherwise every segmented context of
    the process waits forever (seen as a rare stall of later tests)."""
    m = LL.Model(os.path.join(GOLD, "tiny_llama_b_q4_k_m.gguf"))
    d = LL.Context(m, n_ctx=256, n_batch=64)
    d.set_priority(1)
    d.start([5, 9, 13], [0, 1, 2], [[0]] * 3)                    # started, never finished
    d.start([7], [3], [[0]])
    d.free()
    t, ref = LL.Context(m, n_ctx=256, n_batch=64), LL.Context(m, n_ctx=256, n_batch=64)
    t.set_segments(1)                                            # its feeder thread asks the arbiter before every segment
    prompt = [5, 9, 13, 200, 77, 31, 8, 19]
    a = t.decode(prompt, list(range(8)), [[0]] * 8)
    b = ref.decode(prompt, list(range(8)), [[0]] * 8)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    t.free(); ref.free(); m.free()


def test_cancel_of_a_run_in_flight_stops_it_at_a_segment_boundary(LL):
    """llama_cancel_run after llama_start_async_decode (llama.cpp:9981-9993 + the abort callback :5665-5687): the run completes "empty", the
    context stays usable and later runs compute the same bits as on a fresh context"""
    stem = "tiny_llama_b_q4_k_m"
    m = LL.Model(os.path.join(GOLD, stem + ".gguf"))
    c, ref = LL.Context(m, n_ctx=256, n_batch=64), LL.Context(m, n_ctx=256, n_batch=64)
    c.set_segments(1)
    prompt = [5, 9, 13, 200, 77, 31, 8, 19]
    l0 = c.decode(prompt, list(range(8)), [[0]] * 8); r0 = ref.decode(prompt, list(range(8)), [[0]] * 8)
    assert np.array_equal(l0.view(np.uint32), r0.view(np.uint32))
    skipped = 0
    for trial in range(20):
        t = c.start([42, 43, 44], [8, 9, 10], [[1], [1], [1]], batch_id=100 + trial)      # a speculative batch in partition 1
        c.cancel([100 + trial])
        assert c.lib.llama_finish_async_decode(c.h, t[1].c, t[0]) == 1
        c.seq_rm(1, 8, -1)                                                                 # the driver frees the partition's cells
    skipped = c.stats()["n_segments_skipped"]
    l1 = c.decode([42], [8], [[0]]); r1 = ref.decode([42], [8], [[0]])
    assert np.array_equal(l1.view(np.uint32), r1.view(np.uint32))
    print("segments skipped by cancels:", skipped, "of", 20 * 3)
    c.free(); ref.free(); m.free()


def test_bench_starts_its_own_ranks(tmp_path):
    """`python3 bench.py --gpus 3 ...` exactly as the round driver invokes it — no torchrun around it, no RANK / WORLD_SIZE in the environment
    (the reference starts every rank from one command: ggml-mpi.c:58-61, speculative.cpp:116-139).  The parent spawns the ranks as children before
    any GPU call and forwards rank 0's single JSON line.  Three ranks share this one GPU, which needs the explicit override; without it the job refuses."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(PYTHONPATH=root, OMP_NUM_THREADS="2", PIPEINFER_CACHE=str(tmp_path / "models"))
    detail = str(tmp_path / "detail.json")
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--workload", "tiny", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--detail", detail]
    r = subprocess.run(cmd, env=dict(env, PIPEINFER_ALLOW_SHARED_GPU="1"), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1 and len(lines[0]) < 4000, r.stdout                    # ONE small line on stdout (the driver keeps a tail of the output) ...
    line = json.loads(lines[0])
    assert line["n_gpus"] == 3 and line["value"] > 0 and line["lossless_vs_target_only"] is True and line["detail"] == detail
    assert line["transport"] in ("peer", "host") and line["devices_shared"] is True
    rl = line["roofline"]
    assert rl["bound"] == "hbm" and rl["frac"] > 0 and rl["family"] in rl["by_family"] and [s_["rank"] for s_ in rl["stages"]] == [1, 2], rl      # the by-family roofline of N = 1, summed over the stages
    out = json.load(open(detail))                                                 # ... everything else in the detail file
    assert set(out["hop_ms"]) == {"1", "2"} and set(out["wait_for_input_ms"]) == {"1", "2"}
    assert set(out["roofline"]["by_family_per_stage"]) == {"1", "2"}
    st = {s_["rank"]: s_ for s_ in out["roofline"]["stages"]}
    assert st[1]["hop_store"] == ("kernel" if out["transport"] == "peer" else "memcpy") and st[2]["ring_alloc"] in ("fine-grained", "uncached", "coarse-grained (hipMalloc)", "none"), st
    print("bench --gpus 3 (one shared GPU):", {k: out[k] for k in ("value", "transport", "hop_ms", "hop_us_per_run", "wait_for_input_ms")}, {r_: (s_["hop_store"], s_["ring_alloc"]) for r_, s_ in st.items()},
          {k: (v["share_of_gpu_time"], v["frac"]) for k, v in out["roofline"]["by_family"].items()})
    # more ranks than devices without the override: refused loudly, non-zero exit, no result line
    r2 = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
    assert r2.returncode != 0 and "refusing to share devices" in r2.stderr and not r2.stdout.strip()


@pytest.mark.parametrize("segments", [0, 1])
def test_cancel_after_the_start_stops_the_run_at_kernel_granularity(LL, segments):
    """llama_cancel_run after llama_start_async_decode with the device abort word (llama.cpp:9981-9993; the reference's workers poll between graph nodes, ggml.c:15748-15811):
    the cancelled run's kernels return on entry from then on, whether the context launches whole runs itself (segments = 0) or through its feeder; the run finishes "empty" (no
    top-k rows may be sampled from it), the counters see it, a cancel for a run that has already finished changes nothing, and later runs equal a fresh context bit for bit."""
    stem = "tiny_llama_b_q4_k_m"
    m = LL.Model(os.path.join(GOLD, stem + ".gguf"))
    c, ref = LL.Context(m, n_ctx=256, n_batch=64), LL.Context(m, n_ctx=256, n_batch=64)
    if segments:
        c.set_segments(segments)
    prompt = [5, 9, 13, 200, 77, 31, 8, 19]
    l0 = c.decode(prompt, list(range(8)), [[0]] * 8); r0 = ref.decode(prompt, list(range(8)), [[0]] * 8)
    assert np.array_equal(l0.view(np.uint32), r0.view(np.uint32))
    c.reset_stats()
    ids, lg, sm = (C.c_int32 * 8)(), (C.c_float * 8)(), C.c_float()
    for trial in range(12):
        t = c.start([42, 43, 44, 45], [8, 9, 10, 11], [[1]] * 4, batch_id=100 + trial)      # a speculative batch in partition 1
        c.cancel([100 + trial])
        assert c.lib.llama_finish_async_decode(c.h, t[1].c, t[0]) == 1
        assert c.lib.pipeinfer_get_topk(c.h, 0, ids, lg, C.byref(sm)) != 0, "a cancelled run has no rows to sample"
        c.seq_rm(1, 8, -1)                                                                 # the driver frees the partition's cells
        # an uncancelled run right behind it computes the reference's bits (the abort word was re-armed by its own pi_op_abort_begin) ...
        a = c.decode([42 + trial], [8], [[2]]); b = ref.decode([42 + trial], [8], [[2]])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), trial
        c.cancel([100 + trial])                                                            # ... and a late, repeated cancel of the finished run touches nothing
        a = c.decode([7 + trial], [9], [[2]]); b = ref.decode([7 + trial], [9], [[2]])
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), trial
        c.seq_rm(2, 8, -1); ref.seq_rm(2, 8, -1)
    st = c.stats()
    assert st["n_cancel_runs"] == 12 and st["n_cancel_runs_started"] == 12, st
    # what the cancelled runs' launches did instead of working is on the device's books, per kernel family (bench.py keeps them out of the roofline); reset_stats clears them
    import bench as B
    ab = B.aborted_launches(c.lib, c)
    print(f"segments={segments}: 12 runs cancelled after their start took {st['cancel_gpu_ms']:.3f} ms of GPU time in all; segments dropped by the feeder: {st['n_segments_skipped']}; launches aborted on entry per family: {ab}")
    if segments == 0:                                                                      # (the caller enqueued every kernel of a run before the cancel: they were all met on the device; a feeder
        assert sum(ab.values()) > 0, (ab, st)                                              #  usually still has the run in its queue and drops it whole)
    c.reset_stats()
    assert sum(B.aborted_launches(c.lib, c).values()) == 0
    c.free(); ref.free(); m.free()



def _coalescing_scenario(LL, model_path, n_ctx, n_batch, seg_layers, rng):
    """runs started back to back on a coalescing context == the same runs decoded one by one on a plain context, bit for bit.  The batch mixes what the PipeInfer driver
    keeps queued on one GPU: a 3-token speculative run in its own sequence, the non-speculative run of the canonical sequence, a chained speculative run that ATTENDS to the
    cells of the first one (an in-pass dependency), and a run that is cancelled before the feeder gets to it."""
    m = LL.Model(model_path)
    c, ref = LL.Context(m, n_ctx=n_ctx, n_batch=n_batch), LL.Context(m, n_ctx=n_ctx, n_batch=n_batch)
    c.set_segments(seg_layers)
    c.prepare(); ref.prepare()
    c.set_coalescing(16)
    from tests.oracle_follow import RunLog
    log = RunLog()
    c.set_run_tap(log.tap)                                                                 # ... and every run of the coalescing context against the ORACLE (below)
    nv = m.n_vocab
    prompt = [int(x) for x in rng.integers(3, nv, 12)]
    a = c.decode(prompt, list(range(12)), [[0, 1, 2, 3]] * 12); b = ref.decode(prompt, list(range(12)), [[0, 1, 2, 3]] * 12)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    coalesced = 0
    for rnd in range(6):
        t = [int(x) for x in rng.integers(3, nv, 9)]
        p0 = 12 + 4 * rnd
        runs = [([t[0]], [p0], [[0, 1, 2, 3]], 0),                                       # opens the round: the feeder takes it alone, the others queue up behind it
                ([t[1], t[2], t[3]], [p0 + 1, p0 + 2, p0 + 3], [[1]] * 3, 11),            # speculative run, sequence 1
                ([t[4]], [p0 + 1], [[0]], 0),                                              # non-speculative run, canonical sequence
                ([t[5], t[6]], [p0 + 4, p0 + 5], [[1]] * 2, 12),                          # chained: attends to the cells of the sequence-1 run above
                ([t[7], t[8]], [p0 + 1, p0 + 2], [[2]] * 2, 13)]                          # cancelled before it starts computing
        # A cell that changes hands while its first owner is still queued — what the PipeInfer driver does to the first token of a speculative run when it starts the
        # non-speculative run of the same position (speculative.cpp:1252-1255): rounds 2, 3 free the first cell of the speculative run before the canonical run is started,
        # which then takes that very cell.  Round 2 gives it a DIFFERENT token (the runs must not share a pass), round 3 the SAME token (same id, position, cell, n_kv and
        # visibility: the same computation — they may).  Either way every run's bits are those of the runs decoded one after the other.
        handover = rnd in (2, 3)
        if rnd == 3:
            runs[2] = ([t[1]], [p0 + 1], [[0]], 0)
        tickets = []
        for i, (tok, pos, seq, bid) in enumerate(runs):
            if handover and i == 2:
                c.seq_rm(1, p0 + 1, p0 + 2)
            tickets.append(c.start(tok, pos, seq, batch_id=bid))
        if handover:
            assert tickets[2][1].c.n_tokens == 1 and c.lib.pipeinfer_run_first_cell(tickets[2][0]) == c.lib.pipeinfer_run_first_cell(tickets[1][0]), "the canonical run was expected to take the freed cell"
        c.cancel([13])
        got = [None] * 5
        for i in (0, 1, 2, 3):
            got[i] = c.finish(tickets[i])
        assert c.lib.llama_finish_async_decode(c.h, tickets[4][1].c, tickets[4][0]) == 1
        for i, (tok, pos, seq, bid) in enumerate(runs):
            if handover and i == 2:
                ref.seq_rm(1, p0 + 1, p0 + 2)
            want = ref.decode(tok, pos, seq)                                               # one by one, in the same order: same cells
            if i < 4:
                assert np.array_equal(got[i].view(np.uint32), want.view(np.uint32)), (rnd, i, float(np.abs(got[i] - want).max()))
        for x in (c, ref):                                                                # keep the canonical sequence, drop the rest of the round
            x.seq_rm(1, p0 + 1, -1); x.seq_rm(2, p0 + 1, -1); x.seq_rm(0, p0 + 1, -1); x.seq_rm(3, p0 + 1, -1)
            x.seq_cp(0, 1, p0, p0 + 1)
        # top up positions so that the next round continues the canonical sequence
        for x in (c, ref):
            y = x.decode([t[1], t[2], t[3]], [p0 + 1, p0 + 2, p0 + 3], [[0, 1, 2, 3]] * 3)
        coalesced = c.stats()["n_coalesced_passes"]
    st = c.stats()
    c.free(); ref.free(); m.free()
    # the coalescing context's runs — most of them carried by a pass shared with other runs (compose_pass, per-token n_kv and cell table in the attention launches, the
    # same-computation hand-over of round 3, the member cancelled inside its pass) — replayed ONE BY ONE through the oracle on the same cells (tests/oracle_follow.py)
    from tests import oracle_lib as O
    o = O.oracle()
    o.o_set_threads(int(os.environ.get("ORACLE_THREADS", "32")))
    oh = o.o_model_load(model_path.encode(), n_ctx, 8)
    assert oh
    compared, bad = log.replay(o, oh, n_ctx, nv)
    o.o_model_free(oh)
    assert compared >= 6 * 5 and not bad, (compared, bad[:5])
    st["oracle_runs_compared"] = compared
    return st


def test_coalesced_runs_equal_runs_decoded_one_by_one(LL):
    st = _coalescing_scenario(LL, os.path.join(GOLD, "tiny_llama_b_q4_k_m.gguf"), 256, 64, 1, np.random.default_rng(3))
    print("tiny:", {k: st[k] for k in ("n_passes", "n_pass_runs", "n_pass_tokens", "n_coalesced_passes", "n_cancel_runs", "oracle_runs_compared")})
    assert st["n_coalesced_passes"] >= 3 and st["n_pass_runs"] > st["n_passes"]


def test_cancelled_runs_at_the_queue_head_open_no_reaction_window(LL):
    """round 6 (profiles/r06_timeline_70b_before.txt): a cancelled run at the head of the feeder's queue used to travel as an empty pass of its own, and the feeder then sat in
    that pass's reaction window until the caller finished the cancelled run - which the PipeInfer driver does only between two speculative runs.  Now such runs are completed on
    the spot: only passes that put work on the GPU open a window.  Runs 21 and 22 are born cancelled (llama_cancel_run before their start: llama.cpp:5614-5628), so what the
    feeder meets does not depend on timing: [live, cancelled, cancelled, live]."""
    m = LL.Model(os.path.join(GOLD, "tiny_llama_b_q4_k_m.gguf"))
    c, ref = LL.Context(m, n_ctx=256, n_batch=64), LL.Context(m, n_ctx=256, n_batch=64)
    c.set_segments(1)
    c.prepare(); ref.prepare()
    c.set_coalescing(16)
    rng = np.random.default_rng(17)
    prompt = [int(x) for x in rng.integers(3, m.n_vocab, 10)]
    c.decode(prompt, list(range(10)), [[0, 1, 2]] * 10); ref.decode(prompt, list(range(10)), [[0, 1, 2]] * 10)
    for rnd in range(4):
        c.reset_stats()
        t = [int(x) for x in rng.integers(3, m.n_vocab, 7)]
        p0 = 10 + 2 * rnd
        c.cancel([21, 22])
        runs = [([t[0]], [p0], [[0, 1, 2]], 0), ([t[1], t[2], t[3]], [p0 + 1, p0 + 2, p0 + 3], [[1]] * 3, 21), ([t[4], t[5]], [p0 + 1, p0 + 2], [[2]] * 2, 22), ([t[6]], [p0 + 1], [[0]], 0)]              # (the last run lives in sequence 0 only: it must not attend to the cancelled runs' never-written cells)
        tickets = [c.start(tok, pos, seq, batch_id=bid) for tok, pos, seq, bid in runs]
        got0 = c.finish(tickets[0])
        for i in (1, 2):
            assert c.lib.llama_finish_async_decode(c.h, tickets[i][1].c, tickets[i][0]) == 1
        got3 = c.finish(tickets[3])
        st = c.stats()
        assert st["n_passes"] in (1, 2) and st["n_pass_runs"] == 2, st                       # the two live runs, in one pass or two
        assert st["n_reaction_windows"] <= st["n_passes"], st                             # (the old feeder: a window behind each of the two empty passes as well)
        assert st["n_cancel_runs"] == 2, st
        for tok, pos, seq, bid in runs:                                                    # the cancelled runs keep their cells (llama.cpp:5627-5716): the reference context walks the same cells
            want = ref.decode(tok, pos, seq)
            if bid == 0:
                got = got0 if pos[0] == p0 else got3
                assert np.array_equal(want.view(np.uint32), got.view(np.uint32)), (rnd, pos)
        for cc in (c, ref):
            cc.seq_rm(1, p0 + 1, -1); cc.seq_rm(2, p0 + 1, -1)
    for cc in (c, ref):
        cc.free()
    m.free()


@pytest.mark.parametrize("workload", ["70b", "7b"])
def test_coalesced_runs_equal_single_runs_at_full_size_shapes(LL, workload):
    """the same at the shapes of the bench target (Llama-2-70B Q4_K_M cut to 4 layers: GQA 64/8, K = 8192 / 28672, Q4_K + Q5_K / Q6_K attention projections in one launch,
    4..9-token passes on the matrix-core kernel) and of BASELINE configs[1] (7B shapes, 4 layers)"""
    import bench as B
    tspec, tft, _, _ = B.WORKLOADS[workload]
    path = B.ensure_model(os.environ.get("PIPEINFER_CACHE", "/tmp/pipeinfer_models"), tspec, tft, 0x5EED, 1.0, 0.25, layers=(8, 4))
    st = _coalescing_scenario(LL, path, 512, 64, 2, np.random.default_rng(5))
    print(workload, {k: st[k] for k in ("n_passes", "n_pass_runs", "n_pass_tokens", "n_coalesced_passes")})
    assert st["n_coalesced_passes"] >= 3


@pytest.mark.parametrize("workload,n_predict", [("70b-l4", 40), ("7b", 16), pytest.param("70b-l4", 64, marks=pytest.mark.slow), pytest.param("7b", 40, marks=pytest.mark.slow)])              # (the oracle replays every run on the host: the whole 7B model is what a token costs there)
def test_headline_schedule_equals_the_oracle_run_by_run(workload, n_predict):
    """bench.py's `value` configuration itself — PipeInfer driver, target segmented behind its feeder with run coalescing (bench.default_coalesce: 8 tokens), priority draft, one GPU — with every run of the
    generation (prompt passes, single tokens, speculative batches; most of them members of a pass shared with other runs, some cancelled inside one) replayed one by one through
    the oracle on the cells, n_kv and visibility rows the run was given (tests/headline_trace_worker.py; llama.cpp:1593-1644, 5200-5240, speculative.cpp:1252-1255): bit for bit."""
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "headline_trace_worker.py"), workload, str(n_predict)], capture_output=True, text=True, timeout=1500, env=dict(os.environ, PYTHONPATH=root))
    assert r.returncode == 0, (r.stdout[-2500:], r.stderr[-3000:])
    assert "HEADLINE-TRACE OK" in r.stdout, r.stdout[-2000:]
    tail = r.stdout.split("HEADLINE-TRACE OK")[1]
    coalesced, compared = int(tail.split("coalesced_passes")[1].split()[0]), int(tail.split("compared")[1].split()[0])
    # what the comparison is there for: passes that carried several runs, and enough runs behind the bit-for-bit check
    assert coalesced >= 2 and compared >= n_predict // 2, r.stdout[-1500:]
    print(r.stdout[-600:])


@pytest.mark.parametrize("tgt,dft", [("tiny_llama_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_llama_b_q4_k_m", "tiny_llama_q4_k_m"), ("tiny_falcon40_q4_k_m", "tiny_falcon_q4_0"), ("tiny_llama_q3_k_m", "tiny_llama_q4_0")])
@pytest.mark.parametrize("n_draft,n_parallel,p_split,min_spec", [(3, 3, 0.9, 0), (3, 3, 0.9, 1), (8, 1, 0.9, 2), (4, 3, 0.05, 3)])
def test_pipeinfer_driver_is_lossless_with_run_coalescing(LL, tgt, dft, n_draft, n_parallel, p_split, min_spec):
    """the driver's accepted tokens stay the reference target's greedy continuation when the target's feeder carries the runs waiting in its queue together"""
    prompt, want = greedy_reference(tgt, 8, 24)
    mt, md = LL.Model(os.path.join(GOLD, tgt + ".gguf")), LL.Model(os.path.join(GOLD, dft + ".gguf"))
    ct, cd = LL.Context(mt, n_ctx=256, n_batch=64), LL.Context(md, n_ctx=256, n_batch=64)
    ct.set_segments(1); cd.set_priority(1)
    ct.prepare(); cd.prepare()
    ct.set_coalescing(16)
    out, res = LL.speculative(ct, cd, prompt, n_predict=23, n_draft=n_draft, n_parallel=n_parallel, p_accept=0.05, p_split=p_split, ignore_eos=0, min_spec_iters=min_spec)
    assert out[:24] == want, (out, want, res)
    st = ct.stats()
    print(tgt, dft, {k: res[k] for k in ("n_accept_clean", "n_runs_spec", "n_runs_nonspec", "n_runs_canceled")}, {k: st[k] for k in ("n_passes", "n_pass_runs", "n_coalesced_passes")})
    for c in (ct, cd):
        c.free()
    for m in (mt, md):
        m.free()
