"""CPU tests (-m "not gpu"): the oracle against the committed golden vectors (generated from the real reference by
tests/golden/make_golden.py).  These run everywhere, including where oracle/_ref does not exist."""
import os

import numpy as np
import pytest

from tests import oracle_lib as O
from tests import scenarios as S

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
K = np.load(os.path.join(GOLD, "kernels.npz"))
SEEDS = {"tiny_llama_q4_k_m": 11, "tiny_llama_q4_0": 12, "tiny_llama_b_q4_k_m": 13, "tiny_falcon_q4_0": 14, "tiny_falcon40_q4_k_m": 15, "tiny_llama_q3_k_m": 16, "tiny_llama_q2_k": 17, "tiny_llama_q5_1": 18, "tiny_llama_q5_0": 19, "tiny_llama_q4_1": 20, "tiny_llama_hd80_q4_0": 29,
         # the generic-block architectures (goldens made by the reference's own build_* graphs)
         "tiny_baichuan7_q4_0": 21, "tiny_baichuan13_q4_0": 22, "tiny_starcoder_q4_0": 23, "tiny_refact_q4_0": 24, "tiny_bloom_q4_0": 25, "tiny_mpt_q4_k_m": 26, "tiny_stablelm_q4_k_m": 27, "tiny_persimmon_q4_k_m": 28}
MODELS = list(SEEDS)


def bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


K3 = np.load(os.path.join(GOLD, "kernels_q3k.npz"))
K2 = np.load(os.path.join(GOLD, "kernels_q2k.npz"))
KL = np.load(os.path.join(GOLD, "kernels_legacy.npz"))


@pytest.mark.parametrize("t", [O.Q4_0, O.Q4_1, O.Q5_0, O.Q5_1, O.Q8_0, O.Q2_K, O.Q3_K, O.Q4_K, O.Q5_K, O.Q6_K])
def test_quant_golden(t):
    o = O.Impl("oracle")
    name = O.TYPE_NAMES[t]
    blocks, act, qact, dots, deq = ({O.Q3_K: K3, O.Q2_K: K2, O.Q4_1: KL, O.Q5_0: KL, O.Q5_1: KL}.get(t, K)[f"{name}_{s}"] for s in ("blocks", "act", "qact", "dots", "deq"))
    k = act.shape[1]
    for j in range(act.shape[0]):
        assert np.array_equal(o.quantize_act(O.VEC_DOT_TYPE[t], act[j]), qact[j])          # bytes identical
        for i in range(blocks.shape[0]):
            assert np.float32(o.vec_dot(t, k, blocks[i], qact[j])).view(np.uint32) == dots[j, i].view(np.uint32)
    for i in range(blocks.shape[0]):
        assert np.array_equal(bits(o.dequantize(t, blocks[i], k)), bits(deq[i]))
    # error bounds of tests/test-quantize-fns.cpp:16-20: |vec_dot - fp dot| / n < 0.02
    for j in range(act.shape[0]):
        for i in range(blocks.shape[0]):
            assert abs(float(dots[j, i]) - float(np.dot(deq[i].astype(np.float64), act[j].astype(np.float64)))) / k < 0.02


def test_elementwise_golden():
    o = O.Impl("oracle")
    assert np.array_equal(bits(o.rms_norm(K["norm_x"], 1e-5)), bits(K["rms_norm_y"]))
    assert np.array_equal(bits(o.norm(K["norm_x"], 1e-5)), bits(K["layer_norm_y"]))
    assert np.allclose(o.rope(K["rope_x"], K["rope_pos"], 0), K["rope_mode0"], rtol=0, atol=1e-6)
    assert np.allclose(o.rope(K["rope_x"], K["rope_pos"], 2), K["rope_mode2"], rtol=0, atol=1e-6)
    assert np.allclose(o.soft_max(K["softmax_x"]), K["softmax_y"], rtol=1e-6, atol=0)
    assert np.allclose(o.silu(K["unary_x"]), K["silu_y"], rtol=1e-6, atol=0)
    assert np.allclose(o.gelu(K["unary_x"]), K["gelu_y"], rtol=1e-6, atol=0)
    lib = O.oracle()
    assert np.array_equal(np.array([lib.o_fp32_to_fp16(float(v)) for v in K["f16_x"]], np.uint16), K["f16_bits"])


def test_fp16_tables_formula_sample():
    import ctypes as C
    lib = O.oracle()
    idx = K["table_idx"]
    for name, gold in (("o_table_silu_f16", K["table_silu"]), ("o_table_gelu_f16", K["table_gelu"])):
        tab = np.ctypeslib.as_array(C.cast(getattr(lib, name)(), C.POINTER(C.c_uint16)), shape=(65536,))
        assert np.array_equal(tab[idx], gold)


@pytest.mark.parametrize("stem", MODELS)
def test_model_scenarios_golden(stem):
    """whole-path oracle (cell placement + mask + forward) vs logits of the reference kernels, every decode of the
    PipeInfer-style KV script; then the greedy continuation (the lossless-speculation ground truth)"""
    g = np.load(os.path.join(GOLD, stem + ".npz"))
    path = os.path.join(GOLD, stem + ".gguf")
    m = O.OracleModel(path, n_ctx=128, max_seq=8)
    seed = SEEDS[stem]
    outs = S.run_scenario(S.scenario_spec_like(m.n_vocab, seed=seed), m)
    for i, lg in enumerate(outs):
        assert np.allclose(lg, g[f"spec_like_{i}"], rtol=0, atol=1e-5), (stem, i, np.abs(lg - g[f"spec_like_{i}"]).max())
    m.close()
    prompt, n_gen = S.scenario_greedy(m.n_vocab, seed=seed)
    m = O.OracleModel(path, n_ctx=128, max_seq=8)
    lg = m.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
    assert np.allclose(lg, g["greedy_prompt_logits"], rtol=0, atol=1e-5)
    toks, cur = [], int(np.argmax(lg[-1]))
    for i in range(n_gen):
        toks.append(cur)
        lg = m.decode([cur], [len(prompt) + i], [[0]])
        cur = int(np.argmax(lg[0]))
    assert toks == list(g["greedy_tokens"])
    m.close()


@pytest.mark.parametrize("stem", MODELS)
def test_model_scenarios_through_the_reference_graph_builders(stem):
    """the strongest pin of the whole-model path: the reference's own llama.cpp — GGUF loader, KV cache, build_llama / build_falcon graphs, ggml kernels —
    decodes the PipeInfer-style KV script and the greedy continuation of every fixture (O.RefLlama); the oracle's logits are the same BITS on every decode,
    and the committed goldens (made through our own one-node graphs on the reference kernels) agree"""
    if O.ref_kv() is None:
        pytest.skip("oracle/_ref/libref_kv.so is not built here (needs the reference sources)")
    g = np.load(os.path.join(GOLD, stem + ".npz"))
    path = os.path.join(GOLD, stem + ".gguf")
    seed = SEEDS[stem]
    r, m = O.RefLlama(path, n_ctx=128, max_seq=8), O.OracleModel(path, n_ctx=128, max_seq=8)
    ops = S.scenario_spec_like(m.n_vocab, seed=seed)
    outs_r, outs_m = S.run_scenario(ops, r), S.run_scenario(ops, m)
    for i, (a, b) in enumerate(zip(outs_r, outs_m)):
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (stem, i, np.abs(a - b).max())
        assert np.allclose(a, g[f"spec_like_{i}"], rtol=0, atol=1e-5)
    assert r.lib.refkv_head(r.kv) == m.lib.o_kv_head(m.kv) and r.lib.refkv_used(r.kv) == m.lib.o_kv_used(m.kv)
    r.close(); m.close()
    prompt, n_gen = S.scenario_greedy(m.n_vocab, seed=seed)
    r = O.RefLlama(path, n_ctx=128, max_seq=8)
    lg = r.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
    toks, cur = [], int(np.argmax(lg[-1]))
    for i in range(n_gen):
        toks.append(cur)
        cur = int(np.argmax(r.decode([cur], [len(prompt) + i], [[0]])[0]))
    assert toks == list(g["greedy_tokens"])                  # the losslessness ground truth of every engine test
    r.close()


@pytest.mark.parametrize("mode", [0, 2])
def test_rope_composition_property(mode):
    """the reference's tests/test-rope.cpp:138-215: rotating by p0 and then by delta equals rotating by p0 + delta (relative error < 1e-4);
    modes 0 (Llama pairs 2i, 2i+1) and 2 (NeoX halves)"""
    o = O.Impl("oracle")
    rng = np.random.default_rng(5 + mode)
    n_tok, n_head, hd = 7, 4, 128
    x = rng.uniform(-1.0, 1.0, (n_tok, n_head, hd)).astype(np.float32)
    p0 = (np.arange(n_tok) * 3 + 11).astype(np.int32)
    delta = (np.arange(n_tok) * 5 + 2).astype(np.int32)
    r0 = o.rope(x, p0, mode)
    r1 = o.rope(r0, delta, mode)
    r2 = o.rope(x, p0 + delta, mode)
    err = float(np.sqrt(np.sum((r1 - r2).astype(np.float64) ** 2)) / np.sqrt(np.sum(r2.astype(np.float64) ** 2)))
    assert err < 1e-4, err
