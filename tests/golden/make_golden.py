"""Generates the committed fixtures under tests/golden/.  Runs ONLY in the build container (needs oracle/_ref, i.e. the
real reference library compiled from /root/reference).  Outputs are data: tiny quantised GGUF models (weights quantised
by the reference's own quantize_row_*_reference) and the reference kernels' outputs on seeded inputs.

    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from pipeinfer_amd.tools import synth  # noqa: E402
from tests import oracle_lib as O  # noqa: E402
from tests import scenarios as S  # noqa: E402

MODELS = [  # (file stem, spec, ftype, seed)
    ("tiny_llama_q4_k_m", "tiny-llama", "q4_k_m", 11),
    ("tiny_llama_q4_0", "tiny-llama", "q4_0", 12),
    ("tiny_llama_b_q4_k_m", "tiny-llama-b", "q4_k_m", 13),     # "draft"-vs-"target" pair partner, MHA
    ("tiny_falcon_q4_0", "tiny-falcon", "q4_0", 14),           # 7B-style: MQA, single norm, Q8_0 head
    ("tiny_falcon40_q4_k_m", "tiny-falcon40", "q4_k_m", 15),   # 40B-style: GQA, two norms, Q5_K qkv
    ("tiny_llama_q3_k_m", "tiny-llama", "q3_k_m", 16),         # Q3_K / Q4_K / Q5_K / Q6_K mixture of llama.cpp:7919-7961
    ("tiny_llama_q2_k", "tiny-llama", "q2_k", 17),             # Q2_K / Q3_K / Q6_K mixture (llama.cpp:7918, 7936, 7958, 7976)
    ("tiny_llama_q5_1", "tiny-llama", "q5_1", 18),             # legacy 32-blocks with min and fifth bit (x Q8_1 activations), Q6_K head
    ("tiny_llama_q5_0", "tiny-llama", "q5_0", 19),             # signed 5-bit (x Q8_0)
    ("tiny_llama_q4_1", "tiny-llama", "q4_1", 20),
    ("tiny_llama_hd80_q4_0", "tiny-llama-hd80", "q4_0", 29),   # head width 80: ggml_vec_dot_f16's leftover elements (n % 32) in the K.Q products
    # the other architectures: expected logits from the reference's OWN loader + build_* graph (O.RefLlama = oracle/_ref/libref_kv.so), not from our graph over its kernels
    ("tiny_baichuan7_q4_0", "tiny-baichuan7", "q4_0", 21),     # 32 layers: build_baichuan's RoPE arm
    ("tiny_baichuan13_q4_0", "tiny-baichuan13", "q4_0", 22),   # 40 layers: its ALiBi arm
    ("tiny_starcoder_q4_0", "tiny-starcoder", "q4_0", 23),
    ("tiny_refact_q4_0", "tiny-refact", "q4_0", 24),
    ("tiny_bloom_q4_0", "tiny-bloom", "q4_0", 25),
    ("tiny_mpt_q4_k_m", "tiny-mpt", "q4_k_m", 26),
    ("tiny_stablelm_q4_k_m", "tiny-stablelm", "q4_k_m", 27),
    ("tiny_persimmon_q4_k_m", "tiny-persimmon", "q4_k_m", 28),
]
REF_BUILDERS = {"baichuan", "starcoder", "refact", "bloom", "mpt", "stablelm", "persimmon"}


def ref_quantizer():
    r = O.ref()

    def q(ttype, x):
        x = np.ascontiguousarray(x, np.float32)
        if ttype == 0:
            return x.view(np.uint8).reshape(-1)
        out = np.zeros(O.row_size(ttype, x.size), np.uint8)
        r.ref_quantize_reference(ttype, O._ptr(x), O._ptr(out), x.size)
        return out
    return q


def synth_signal(n, seed):
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float32)
    return (0.1 + 2.0 * np.cos(i + seed)).astype(np.float32) + rng.standard_normal(n).astype(np.float32) * 0.5


def kernel_vectors():
    """K1/K2 of SURVEY 8c: per-type quant blocks + activations + reference outputs; elementwise op outputs"""
    r = O.Impl("ref")
    q = ref_quantizer()
    out = {}
    k = 1024
    for t in (O.Q4_0, O.Q8_0, O.Q4_K, O.Q5_K, O.Q6_K):
        rng = np.random.default_rng(100 + t)
        w = (rng.standard_normal((6, k)) * 0.05).astype(np.float32)
        blocks = q(t, w).reshape(6, -1)
        act = np.stack([synth_signal(k, s + t) for s in range(3)])
        vdt = O.VEC_DOT_TYPE[t]
        qa = np.stack([r.quantize_act(vdt, a) for a in act])
        dots = np.array([[r.vec_dot(t, k, blocks[i], qa[j]) for i in range(6)] for j in range(3)], np.float32)
        name = O.TYPE_NAMES[t]
        out[f"{name}_blocks"] = blocks
        out[f"{name}_act"] = act
        out[f"{name}_qact"] = qa
        out[f"{name}_dots"] = dots
        out[f"{name}_deq"] = np.stack([r.dequantize(t, b, k) for b in blocks])
    x = np.stack([synth_signal(512, s) for s in range(3)])
    out["norm_x"] = x
    out["rms_norm_y"] = r.rms_norm(x, 1e-5)
    out["layer_norm_y"] = r.norm(x, 1e-5)
    rng = np.random.default_rng(7)
    rx = rng.standard_normal((5, 3, 64)).astype(np.float32)
    pos = np.array([0, 1, 9, 300, 1023], np.int32)
    out["rope_x"] = rx
    out["rope_pos"] = pos
    out["rope_mode0"] = r.rope(rx, pos, 0)
    out["rope_mode2"] = r.rope(rx, pos, 2)
    sx = (rng.standard_normal((6, 77)) * 3).astype(np.float32)
    sx[:, 60:] = -np.inf
    sx[2, :5] = -np.inf
    out["softmax_x"] = sx
    out["softmax_y"] = r.soft_max(sx)
    ux = (rng.standard_normal(4096) * 3).astype(np.float32)
    out["unary_x"] = ux
    out["silu_y"] = r.silu(ux)
    out["gelu_y"] = r.gelu(ux)
    fx = (rng.standard_normal(4096) * np.float32(10.0) ** rng.integers(-8, 5, 4096)).astype(np.float32)
    out["f16_x"] = fx
    h = np.zeros(fx.size, np.uint16)
    O.ref().ref_fp32_to_fp16_row(O._ptr(fx), O._ptr(h), fx.size)
    out["f16_bits"] = h
    # the three fp16 tables are NOT committed; a 256-entry strided sample pins the formula
    idx = np.arange(0, 65536, 257, dtype=np.int64)
    tx = idx.astype(np.uint16).view(np.float16).astype(np.float32)
    fin = np.isfinite(tx)
    out["table_idx"] = idx[fin].astype(np.int32)
    out["table_silu"] = O.f32_to_f16_bits(r.silu(tx[fin]))
    out["table_gelu"] = O.f32_to_f16_bits(r.gelu(tx[fin]))
    np.savez_compressed(os.path.join(HERE, "kernels.npz"), **out)


def kernel_vectors_q3k(t=None, fname="kernels_q3k.npz"):
    """same vectors as kernel_vectors() for Q3_K (and Q2_K), in files of their own so that kernels.npz stays byte-identical"""
    r = O.Impl("ref")
    q = ref_quantizer()
    out = {}
    k, t = 1024, (O.Q3_K if t is None else t)
    rng = np.random.default_rng(100 + t)
    w = (rng.standard_normal((6, k)) * 0.05).astype(np.float32)
    blocks = q(t, w).reshape(6, -1)
    act = np.stack([synth_signal(k, s + t) for s in range(3)])
    qa = np.stack([r.quantize_act(O.VEC_DOT_TYPE[t], a) for a in act])
    name = O.TYPE_NAMES[t]
    out[f"{name}_blocks"] = blocks
    out[f"{name}_act"] = act
    out[f"{name}_qact"] = qa
    out[f"{name}_dots"] = np.array([[r.vec_dot(t, k, blocks[i], qa[j]) for i in range(6)] for j in range(3)], np.float32)
    out[f"{name}_deq"] = np.stack([r.dequantize(t, b, k) for b in blocks])
    np.savez_compressed(os.path.join(HERE, fname), **out)


def kernel_vectors_legacy():
    """the same vectors for the legacy 32-blocks Q4_1 / Q5_0 / Q5_1 (x Q8_1 / Q8_0 activations), in a file of their own"""
    r = O.Impl("ref")
    q = ref_quantizer()
    out = {}
    k = 1024
    for t in (O.Q4_1, O.Q5_0, O.Q5_1):
        rng = np.random.default_rng(100 + t)
        w = (rng.standard_normal((6, k)) * 0.05 + 0.01).astype(np.float32)
        blocks = q(t, w).reshape(6, -1)
        act = np.stack([synth_signal(k, s + t) for s in range(3)])
        qa = np.stack([r.quantize_act(O.VEC_DOT_TYPE[t], a) for a in act])
        name = O.TYPE_NAMES[t]
        out[f"{name}_blocks"] = blocks
        out[f"{name}_act"] = act
        out[f"{name}_qact"] = qa
        out[f"{name}_dots"] = np.array([[r.vec_dot(t, k, blocks[i], qa[j]) for i in range(6)] for j in range(3)], np.float32)
        out[f"{name}_deq"] = np.stack([r.dequantize(t, b, k) for b in blocks])
    np.savez_compressed(os.path.join(HERE, "kernels_legacy.npz"), **out)


QUANT_TYPES = [(2, "q4_0"), (3, "q4_1"), (6, "q5_0"), (7, "q5_1"), (8, "q8_0"), (10, "q2_K"), (11, "q3_K"), (12, "q4_K"), (13, "q5_K"), (14, "q6_K")]


def quantize_inputs(k=1024, seed=77):
    """rows that exercise the searches of the offline quantisers: gaussian weights, zeros, a constant, mixed magnitudes, one-sided, heavy tails, sparse"""
    rng = np.random.default_rng(seed)
    rows = [rng.standard_normal(k) * 0.02, np.zeros(k), np.full(k, 0.37), rng.standard_normal(k) * rng.choice([1e-6, 1, 300], k), rng.uniform(0, 1, k), -rng.uniform(0, 1, k),
            rng.standard_t(2, k), np.where(rng.uniform(size=k) < 0.9, 0, rng.standard_normal(k)), rng.standard_normal(k) * 1e-7, rng.standard_normal(k) * 2e3]
    return np.stack(rows).astype(np.float32)


def quantize_vectors():
    """tests/golden/quantize.npz: the reference's quantize_row_*_reference (what llama_model_quantize writes) on quantize_inputs(), one byte row per (type, input row)"""
    r = O.ref()
    x = quantize_inputs()
    out = {"x": x}
    for t, name in QUANT_TYPES:
        n = O.row_size(t, x.shape[1])
        y = np.zeros((x.shape[0], n), np.uint8)
        for i in range(x.shape[0]):
            r.ref_quantize_reference(t, O._ptr(x[i]), O._ptr(y[i]), x.shape[1])
        out[name] = y
    np.savez_compressed(os.path.join(HERE, "quantize.npz"), **out)
    print("quantize.npz", {k: v.shape for k, v in out.items()})


def model_vectors(only=None):
    q = ref_quantizer()
    for stem, spec, ftype, seed in MODELS:
        if only and stem not in only:
            continue
        path = os.path.join(HERE, stem + ".gguf")
        synth.write_model(path, spec, ftype, seed=seed, std=0.08, quantizer=q)
        n_vocab = synth.SPECS[spec][6]
        res = {}
        Ref = O.RefLlama if synth.SPECS[spec][0] in REF_BUILDERS else O.RefModel
        m = Ref(path, n_ctx=128, max_seq=8)
        ops = S.scenario_spec_like(n_vocab, seed=seed)
        for i, lg in enumerate(S.run_scenario(ops, m)):
            res[f"spec_like_{i}"] = lg
        m.close()
        # greedy continuation: prompt batched, then token by token (the lossless-speculation ground truth)
        prompt, n_gen = S.scenario_greedy(n_vocab, seed=seed)
        m = Ref(path, n_ctx=128, max_seq=8)
        lg = m.decode(prompt, list(range(len(prompt))), [[0]] * len(prompt))
        res["greedy_prompt_logits"] = lg
        toks = []
        cur = int(np.argmax(lg[-1]))
        for i in range(n_gen):
            toks.append(cur)
            lg = m.decode([cur], [len(prompt) + i], [[0]])
            cur = int(np.argmax(lg[0]))
        res["greedy_tokens"] = np.array(toks, np.int32)
        res["greedy_last_logits"] = lg
        m.close()
        np.savez_compressed(os.path.join(HERE, stem + ".npz"), **res)
        print(stem, os.path.getsize(path), "bytes; greedy:", toks[:8])


if __name__ == "__main__":
    assert O.ref() is not None, "needs oracle/_ref (make -C oracle ref)"
    if sys.argv[1:] == ["legacy"]:
        kernel_vectors_legacy()
        model_vectors(only=["tiny_llama_q5_1", "tiny_llama_q5_0", "tiny_llama_q4_1"])
    elif sys.argv[1:] == ["quantize"]:
        quantize_vectors()
    elif sys.argv[1:] == ["q3k"]:                    # later addition: does not regenerate the other fixtures
        kernel_vectors_q3k()
        model_vectors(only=["tiny_llama_q3_k_m"])
    elif sys.argv[1:] == ["persimmon"]:
        model_vectors(only=["tiny_persimmon_q4_k_m"])
    elif sys.argv[1:] == ["archs"]:                  # round 3: the generic-block architectures
        model_vectors(only=[m[0] for m in MODELS if synth.SPECS[m[1]][0] in REF_BUILDERS])
    elif sys.argv[1:] == ["hd80"]:                   # round 5: a head width that is neither 64 nor 128
        model_vectors(only=["tiny_llama_hd80_q4_0"])
    elif sys.argv[1:] == ["q2k"]:
        kernel_vectors_q3k(O.Q2_K, "kernels_q2k.npz")
        model_vectors(only=["tiny_llama_q2_k"])
    else:
        kernel_vectors()
        kernel_vectors_q3k()
        kernel_vectors_q3k(O.Q2_K, "kernels_q2k.npz")
        kernel_vectors_legacy()
        quantize_vectors()
        model_vectors()
