"""Pins the oracle (oracle/*.c, our restatement) against the REAL reference kernels (oracle/_ref, built from
/root/reference by oracle/Makefile).  Runs wherever oracle/_ref exists (this container and the GPU box, which
receives the prebuilt .so); skipped otherwise — tests/test_oracle_golden.py then covers the same ground from
committed vectors."""
import numpy as np
import pytest

from tests import oracle_lib as O

pytestmark = pytest.mark.skipif(O.ref() is None, reason="oracle/_ref not built (needs /root/reference)")

QTYPES = [O.Q4_0, O.Q4_1, O.Q5_0, O.Q5_1, O.Q8_0, O.Q2_K, O.Q3_K, O.Q4_K, O.Q5_K, O.Q6_K]


def synth(n, seed):
    """the reference's own test signal (tests/test-quantize-fns.cpp:26-30) plus gaussian noise"""
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float32)
    return (0.1 + 2.0 * np.cos(i + seed)).astype(np.float32) + rng.standard_normal(n).astype(np.float32) * 0.5


def make_weights(t, rows, k, seed):
    r = O.Impl("ref")
    rng = np.random.default_rng(seed)
    w = (rng.standard_normal((rows, k)) * 0.05).astype(np.float32)
    out = np.zeros(rows * O.row_size(t, k), np.uint8)
    r.lib.ref_quantize_reference(t, O._ptr(w), O._ptr(out), rows * k)
    return out.reshape(rows, -1)


def test_fp16_conversion_exhaustive():
    lib = O.oracle()
    r = O.ref()
    bits = np.arange(65536, dtype=np.uint16)
    f_ref = np.zeros(65536, np.float32)
    r.ref_fp16_to_fp32_row(O._ptr(bits), O._ptr(f_ref), 65536)
    f_or = np.array([lib.o_fp16_to_fp32(int(b)) for b in bits], np.float32)
    assert np.array_equal(f_ref.view(np.uint32), f_or.view(np.uint32))
    # f32 -> f16 on a dense sample incl. ties, subnormals, overflow
    rng = np.random.default_rng(0)
    x = np.concatenate([
        f_ref[np.isfinite(f_ref)],
        (f_ref[np.isfinite(f_ref)][:-1] + f_ref[np.isfinite(f_ref)][1:]) / 2,     # exact ties between neighbours
        rng.standard_normal(200000).astype(np.float32) * np.float32(10.0) ** rng.integers(-9, 6, 200000).astype(np.float32),
        np.array([65504, 65519.99, 65520, 65536, 1e9, -1e9, 5.96e-8, 2.98e-8, 2.9802322e-8, 1e-10, 0.0, -0.0], np.float32),
    ]).astype(np.float32)
    h_ref = np.zeros(x.size, np.uint16)
    r.ref_fp32_to_fp16_row(O._ptr(x), O._ptr(h_ref), x.size)
    h_or = np.array([lib.o_fp32_to_fp16(float(v)) for v in x], np.uint16)
    assert np.array_equal(h_ref, h_or)
    assert np.array_equal(h_ref, x.astype(np.float16).view(np.uint16))           # numpy agrees too (RNE)


@pytest.mark.parametrize("vdt", [O.Q8_K, O.Q8_0, O.Q8_1])
def test_activation_quantisers_bit_exact(vdt):
    o, r = O.Impl("oracle"), O.Impl("ref")
    for seed in range(6):
        x = synth(4096, seed) * (0.01 if seed % 2 else 3.0)
        if seed == 5:
            x[256:512] = 0.0        # an all-zero block
            x[7] = 0.5; x[8] = 1.5  # exact .5 ties after scaling are likely somewhere
        a, b = o.quantize_act(vdt, x), r.quantize_act(vdt, x)
        if vdt == O.Q8_K:
            # all-zero block: the reference leaves bsums uninitialised (ggml-quants.c:2345-2349); compare d and qs only there
            a = a.reshape(-1, 292).copy(); b = b.reshape(-1, 292).copy()
            zero = np.all(a[:, 4:260] == 0, axis=1)
            a[zero, 260:] = 0; b[zero, 260:] = 0
        assert np.array_equal(a, b)


@pytest.mark.parametrize("t", QTYPES)
def test_dequantize_bit_exact(t):
    o, r = O.Impl("oracle"), O.Impl("ref")
    w = make_weights(t, 4, 1024, 3)
    for row in w:
        assert np.array_equal(o.dequantize(t, row, 1024).view(np.uint32), r.dequantize(t, row, 1024).view(np.uint32))


@pytest.mark.parametrize("t", QTYPES)
@pytest.mark.parametrize("k", [256, 1024, 4096, 11008 // 256 * 256])
def test_vec_dot_bit_exact(t, k):
    """same AVX2 lane order => identical fp32 bits"""
    o, r = O.Impl("oracle"), O.Impl("ref")
    w = make_weights(t, 8, k, k)
    for i, row in enumerate(w):
        y = r.quantize_act(O.VEC_DOT_TYPE[t], synth(k, i))
        a, b = o.vec_dot(t, k, row, y), r.vec_dot(t, k, row, y)
        assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32), (a, b)


@pytest.mark.parametrize("n", [32, 64, 128, 33, 100, 1000])
def test_vec_dot_f16_bit_exact(n):
    o, r = O.Impl("oracle"), O.Impl("ref")
    rng = np.random.default_rng(n)
    x = O.f32_to_f16_bits(rng.standard_normal(n)); y = O.f32_to_f16_bits(rng.standard_normal(n))
    a, b = o.vec_dot(O.F16, n, x, y), r.vec_dot(O.F16, n, x, y)
    assert np.float32(a).view(np.uint32) == np.float32(b).view(np.uint32)


@pytest.mark.parametrize("t", QTYPES + [O.F16])
def test_mul_mat_bit_exact(t):
    o, r = O.Impl("oracle"), O.Impl("ref")
    k, rows = 512, 48
    if t == O.F16:
        w = O.f32_to_f16_bits(np.random.default_rng(1).standard_normal((rows, k)) * 0.1)
    else:
        w = make_weights(t, rows, k, 11)
    act = np.stack([synth(k, s) for s in range(5)])
    a, b = o.mul_mat(t, w, k, rows, act), r.mul_mat(t, w, k, rows, act)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_norms_bit_exact():
    o, r = O.Impl("oracle"), O.Impl("ref")
    x = np.stack([synth(4096, s) for s in range(4)])
    assert np.array_equal(o.rms_norm(x, 1e-5).view(np.uint32), r.rms_norm(x, 1e-5).view(np.uint32))
    assert np.array_equal(o.norm(x, 1e-5).view(np.uint32), r.norm(x, 1e-5).view(np.uint32))
    x = np.stack([synth(4544, s) for s in range(3)])           # Falcon-7B width
    assert np.array_equal(o.norm(x, 1e-5).view(np.uint32), r.norm(x, 1e-5).view(np.uint32))


@pytest.mark.parametrize("mode,hd", [(0, 128), (0, 64), (2, 64), (2, 128)])
def test_rope_bit_exact(mode, hd):
    o, r = O.Impl("oracle"), O.Impl("ref")
    rng = np.random.default_rng(hd + mode)
    x = rng.standard_normal((6, 4, hd)).astype(np.float32)
    pos = np.array([0, 1, 2, 17, 511, 1023], np.int32)
    assert np.array_equal(o.rope(x, pos, mode).view(np.uint32), r.rope(x, pos, mode).view(np.uint32))


def test_rope_consistency_property():
    """tests/test-rope.cpp:138-215: rope(rope(x, p0), d) ~= rope(x, p0 + d)"""
    o = O.Impl("oracle")
    rng = np.random.default_rng(5)
    x = rng.standard_normal((3, 2, 128)).astype(np.float32)
    for mode in (0, 2):
        a = o.rope(o.rope(x, np.array([10, 20, 30]), mode), np.array([7, 7, 7]), mode)
        b = o.rope(x, np.array([17, 27, 37]), mode)
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-4


def test_softmax_bit_exact():
    o, r = O.Impl("oracle"), O.Impl("ref")
    rng = np.random.default_rng(9)
    x = (rng.standard_normal((16, 200)) * 4).astype(np.float32)
    x[:, 150:] = -np.inf
    x[3, :7] = -np.inf
    a, b = o.soft_max(x), r.soft_max(x)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
    assert np.all(a[:, 150:] == 0.0)


def test_softmax_known_answers():
    """tests/test-sampling.cpp softmax-derived probabilities; the fp16 exp table costs ~1e-3 relative"""
    o = O.Impl("oracle")
    p = o.soft_max(np.log(np.array([[0.1, 0.2, 0.3, 0.4]], np.float32)))
    assert np.allclose(p, [[0.1, 0.2, 0.3, 0.4]], rtol=2e-3)


def test_silu_gelu_all_fp16_inputs_bit_exact():
    o, r = O.Impl("oracle"), O.Impl("ref")
    bits = np.arange(65536, dtype=np.uint16)
    x = bits.view(np.float16).astype(np.float32)
    x = x[np.isfinite(x)]
    rng = np.random.default_rng(2)
    x = np.concatenate([x, (rng.standard_normal(50000) * 3).astype(np.float32)])
    for fn in ("silu", "gelu"):
        a, b = getattr(o, fn)(x), getattr(r, fn)(x)
        assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), fn


@pytest.mark.parametrize("t", QTYPES + [O.F16, O.F32])
def test_get_rows_bit_exact(t):
    o, r = O.Impl("oracle"), O.Impl("ref")
    k, rows = 512, 20
    if t == O.F16:
        w = O.f32_to_f16_bits(np.random.default_rng(1).standard_normal((rows, k)))
    elif t == O.F32:
        w = np.random.default_rng(1).standard_normal((rows, k)).astype(np.float32)
    else:
        w = make_weights(t, rows, k, 4)
    ids = np.array([0, 19, 7, 7, 3], np.int32)
    assert np.array_equal(o.get_rows(t, w, k, rows, ids).view(np.uint32), r.get_rows(t, w, k, rows, ids).view(np.uint32))


@pytest.mark.parametrize("hd,n_head,n_head_kv,n_kv", [(64, 8, 2, 32), (128, 4, 4, 45), (64, 8, 1, 70), (64, 6, 2, 33)])
def test_attention_block_bit_exact(hd, n_head, n_head_kv, n_kv):
    """KV store (f16, V transposed) + KQ + scale + mask + softmax + KQV, GQA/MQA, ragged n_kv, partial masks"""
    o, r = O.Impl("oracle"), O.Impl("ref")
    rng = np.random.default_rng(hd + n_kv)
    n_ctx, n_tok, kv_head = 96, 3, n_kv - 3
    g = hd * n_head_kv
    kc = O.f32_to_f16_bits(rng.standard_normal((n_ctx, g))).copy()
    vc = O.f32_to_f16_bits(rng.standard_normal((g, n_ctx))).copy()
    q = rng.standard_normal((n_tok, n_head, hd)).astype(np.float32)
    k = rng.standard_normal((n_tok, n_head_kv, hd)).astype(np.float32)
    v = rng.standard_normal((n_tok, g)).astype(np.float32)
    mask = np.zeros((n_tok, n_kv), np.float32)
    mask[0, kv_head + 1:] = -np.inf
    mask[1, kv_head + 2:] = -np.inf
    mask[:, 5:9] = -np.inf
    kc2, vc2 = kc.copy(), vc.copy()
    a = o.attention(kc, vc, n_ctx, q, k, v, mask, hd, n_head, n_head_kv, kv_head, n_kv)
    b = r.attention(kc2, vc2, n_ctx, q, k, v, mask, hd, n_head, n_head_kv, kv_head, n_kv)
    assert np.array_equal(kc, kc2) and np.array_equal(vc, vc2)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))
