"""-m gpu: every hot-path HIP operator, called through the C ABI (include/pipeinfer_ops.h), against the oracle on the same
seeded inputs and against the committed golden vectors (outputs of the real reference kernels).

The bar here is BIT-EXACT, fp32 included: the mat-vec and attention kernels keep the reference AVX2 arm's accumulation order
(eight fp32 lanes per row, fma per block in block order, hsum tree), because f16 rounding and Q8 activation quantisation
downstream amplify a 1-ulp difference into 1e-2 logit differences.  Only the norms (double-precision sums in a different
order) carry a tolerance, stated at the assert."""
import ctypes as C
import os

import numpy as np
import pytest

from pipeinfer_amd.tools import gguf as G
from pipeinfer_amd.tools import synth
from tests import oracle_lib as O

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
LEGACY = [O.Q4_1, O.Q5_0, O.Q5_1]
QTYPES = [O.Q4_0, O.Q8_0, O.Q2_K, O.Q3_K, O.Q4_K, O.Q5_K, O.Q6_K] + LEGACY


@pytest.fixture(scope="module")
def gpu():
    from tests import gpu_util as U
    lib = U.init()
    return lib, U


def signal(n, seed):
    rng = np.random.default_rng(seed)
    i = np.arange(n, dtype=np.float32)
    return (0.1 + 2.0 * np.cos(i + seed)).astype(np.float32) + rng.standard_normal(n).astype(np.float32) * 0.5


def upload_weight(lib, t, rows, k, blocks):
    blocks = np.ascontiguousarray(blocks)
    w = lib.pi_weight_upload(t, k, rows, blocks.ctypes.data_as(C.c_void_p), None)
    assert w, lib.pi_last_error()
    assert lib.pi_weight_nbytes(w) == rows * O.row_size(t, k)          # HBM bytes == ggml_nbytes (roofline accounting)
    return w


def gpu_quantize(lib, U, vdt, x):
    n_tok, k = x.shape
    xd = U.Dev(x.astype(np.float32))
    q8 = U.Dev(nbytes=lib.pi_q8_nbytes(vdt, k, n_tok))
    from pipeinfer_amd import lib as L
    L.check(lib.pi_op_quantize(vdt, xd.ptr, k, n_tok, q8.ptr, None), "quantize")
    L.check(lib.pi_stream_sync(None))            # xd is released on return
    return q8


def gpu_mul_mat(lib, U, t, w, k, rows, act, residual=None):
    from pipeinfer_amd import lib as L
    n_tok = act.shape[0]
    q8 = gpu_quantize(lib, U, O.VEC_DOT_TYPE[t], act)
    out = U.Dev(nbytes=4 * n_tok * rows)
    res = U.Dev(residual.astype(np.float32)) if residual is not None else None
    L.check(lib.pi_op_mul_mat(w, q8.ptr, n_tok, out.ptr, res.ptr if res else None, None), "mul_mat")
    return out.get(np.float32, (n_tok, rows))


@pytest.mark.parametrize("vdt", [O.Q8_K, O.Q8_0, O.Q8_1])
def test_activation_quantiser_bytes_identical(gpu, vdt):
    lib, U = gpu
    o = O.Impl("oracle")
    for k in (256, 1024, 4096, 11008 // 256 * 256) + ((4544, 18176) if vdt != O.Q8_K else ()):
        x = np.stack([signal(k, s) * (0.01 if s % 2 else 3.0) for s in range(5)])
        x[4, :256] = 0.0                              # all-zero block
        x[3, 10] = 2.5; x[3, 11] = -2.5               # equal |x| with opposite signs: first one fixes the sign of the scale
        q8 = gpu_quantize(lib, U, vdt, x)
        got = U.q8_as_ref_blocks(vdt, *U.q8_planes(lib, vdt, k, 5, q8))
        for t in range(5):
            assert np.array_equal(got[t], o.quantize_act(vdt, x[t])), (k, t)


@pytest.mark.parametrize("vdt", [O.Q8_K, O.Q8_0, O.Q8_1])
def test_activation_quantiser_golden(gpu, vdt):
    lib, U = gpu
    K = np.load(os.path.join(GOLD, "kernels_legacy.npz" if vdt == O.Q8_1 else "kernels.npz"))
    for t in ([O.Q4_K, O.Q5_K, O.Q6_K] if vdt == O.Q8_K else [O.Q4_1, O.Q5_1] if vdt == O.Q8_1 else [O.Q4_0, O.Q8_0]):
        name = O.TYPE_NAMES[t]
        act, qact = K[f"{name}_act"], K[f"{name}_qact"]
        q8 = gpu_quantize(lib, U, vdt, act)
        got = U.q8_as_ref_blocks(vdt, *U.q8_planes(lib, vdt, act.shape[1], act.shape[0], q8))
        assert np.array_equal(got, qact)


@pytest.mark.parametrize("t", QTYPES)
def test_mul_mat_golden_blocks(gpu, t):
    """weights quantised by the reference's own quantiser + reference vec_dot results (tests/golden/kernels.npz)"""
    lib, U = gpu
    K = np.load(os.path.join(GOLD, {O.Q3_K: "kernels_q3k.npz", O.Q2_K: "kernels_q2k.npz", O.Q4_1: "kernels_legacy.npz", O.Q5_0: "kernels_legacy.npz", O.Q5_1: "kernels_legacy.npz"}.get(t, "kernels.npz")))
    name = O.TYPE_NAMES[t]
    blocks, act, dots = K[f"{name}_blocks"], K[f"{name}_act"], K[f"{name}_dots"]
    rows, k = blocks.shape[0], act.shape[1]
    w = upload_weight(lib, t, rows, k, blocks)
    got = gpu_mul_mat(lib, U, t, w, k, rows, act)
    assert np.array_equal(got.view(np.uint32), dots.view(np.uint32))          # same fp32 bits as the reference's vec_dot
    lib.pi_weight_free(w)


@pytest.mark.parametrize("t", QTYPES)
@pytest.mark.parametrize("k,rows,n_tok", [(256, 37, 3), (768, 64, 1), (1024, 100, 8), (2048, 256, 2), (4096, 515, 1), (4096, 515, 5), (11008, 96, 11),
                                          (28672, 40, 8), (11008, 96, 1), (4544, 71, 4), (18176, 24, 2), (256, 1024, 7), (512, 8, 6),
                                          # the loop-free decode instances: single tile up to 4 tokens, tile per wave (2..16 tiles) up to 2, longer rows fall back to the loop
                                          (4096, 33, 3), (4096, 33, 4), (8192, 40, 1), (28672, 17, 1), (28672, 17, 2), (65536, 5, 1), (69632, 6, 1), (69632, 6, 2), (3840, 9, 2),
                                          (8192, 24, 3), (8192, 24, 4), (5120, 10, 3), (8192, 24, 2), (14336, 12, 2), (14336, 12, 3),
                                          # 70B-sized launches: four rows per wave group (K >= 8192 and >= 8192 rows), ragged last group
                                          (8192, 8198, 1), (8192, 8198, 2),
                                          # matrix-core kernel (2..16 tokens per pass, 1/2/4/8 waves per 16-row group, ragged row groups, prompts in chunks of 16)
                                          (4096, 64, 16), (4096, 49, 9), (8192, 2064, 12), (28672, 40, 16), (2048, 16, 2), (512, 4100, 13), (4096, 80, 17), (1024, 32, 37),
                                          # ... batches of <= 8 tokens reach it from rows x tokens >= 16384 (half-size fragments); 513..768 row groups run three to a workgroup
                                          (512, 4100, 5), (1024, 2100, 8), (256, 5500, 3), (512, 9222, 5), (256, 8200, 3), (768, 12288, 16), (512, 12000, 2), (512, 7000, 6),
                                          # ... Q6_K: rows of K >= 16384 from 3 tokens, shorter ones from 5 (round 6: the threshold depends on the row length)
                                          (16384, 5500, 3), (16384, 4100, 4), (4096, 4100, 4), (4096, 4100, 5),
                                          # ... 17..32 tokens: two chunks of 16 over ONE weight stream (every waves-per-row-group choice; a ragged second chunk; prompts in steps of 32)
                                          (4096, 2064, 32), (2048, 9000, 24), (1024, 14000, 33), (512, 20000, 32), (768, 12288, 48), (256, 30000, 31), (8192, 130, 64), (1024, 6200, 18),
                                          # ... 33..64 tokens: four chunks over ONE weight stream, activations staged through registers (prompt passes; a ragged last chunk, 64 + 36 tokens,
                                          # every waves-per-row-group choice: <= 383 / 384..512 / 513..768 / more row groups)
                                          (4096, 2064, 64), (2048, 9000, 40), (512, 20000, 100), (256, 7000, 128), (28672, 40, 50), (1024, 10000, 57),
                                          # Q4_0 / Q8_0 range-per-wave kernel: 4, 8, 12 groups per wave, ragged last wave, rows not a multiple of 8
                                          (1024, 19, 1), (4544, 71, 1), (4544, 71, 2), (16384, 9, 1), (18176, 13, 1), (24576, 11, 1), (24576, 11, 2), (32768, 8, 1),
                                          # rows shorter than a 256-value chunk / with a last half chunk (32-blocks only): the tiny Baichuan, Refact and Bloom fixtures
                                          (128, 40, 1), (128, 40, 4), (384, 20, 2), (384, 20, 9), (32, 16, 1)])
def test_mul_mat_vs_oracle(gpu, t, k, rows, n_tok):
    if k % O.BLCK[t]:
        pytest.skip("K-quants need K % 256 == 0")
    lib, U = gpu
    o = O.Impl("oracle")
    rng = np.random.default_rng(k + rows + n_tok)
    blocks = synth.random_blocks(t, rows * k, rng, 0.05).reshape(rows, -1)
    act = np.stack([signal(k, s + n_tok) for s in range(n_tok)])
    res = rng.standard_normal((n_tok, rows)).astype(np.float32)
    w = upload_weight(lib, t, rows, k, blocks)
    want = o.mul_mat(t, blocks, k, rows, act)
    got = gpu_mul_mat(lib, U, t, w, k, rows, act)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()
    got_r = gpu_mul_mat(lib, U, t, w, k, rows, act, residual=res)
    assert np.array_equal(got_r.view(np.uint32), (res + want).view(np.uint32))
    lib.pi_weight_free(w)


@pytest.mark.parametrize("t", QTYPES)
def test_mul_mat_integer_sums_exact(gpu, t):
    """Scales forced to powers of two and integer activations small enough that every partial sum is an integer < 2^24:
    any fp32 summation order then gives the same bits, so equality with the oracle proves the int8 dot products exact."""
    lib, U = gpu
    o = O.Impl("oracle")
    k, rows, n_tok = 1024, 64, 4
    rng = np.random.default_rng(99 + t)
    blocks = synth.random_blocks(t, rows * k, rng, 0.05).reshape(rows * (k // O.BLCK[t]), O.TSIZE[t]).copy()
    one = np.array([0x3C00], np.uint16).view(np.uint8)          # f16 1.0
    if t in (O.Q4_0, O.Q8_0, O.Q5_0):
        blocks[:, 0:2] = one
    elif t in (O.Q4_1, O.Q5_1):
        blocks[:, 0:2] = one; blocks[:, 2:4] = one                                       # d = m = 1: summs = sum of the block sums, an integer
    elif t in (O.Q4_K, O.Q5_K):
        blocks[:, 0:2] = one; blocks[:, 2:4] = one
    elif t == O.Q2_K:
        blocks[:, 80:82] = one; blocks[:, 82:84] = one                                   # scales and mins are 4-bit already
    elif t == O.Q3_K:
        blocks[:, 108:110] = one                                                         # |scale - 32| <= 19 as generated
    else:
        blocks[:, 208:210] = one
        blocks[:, 192:208] = (blocks[:, 192:208].view(np.int8) // 16).view(np.uint8)     # |scale| <= 8
    blocks = blocks.reshape(rows, -1)
    act = rng.integers(-8, 9, (n_tok, k)).astype(np.float32)
    blk = O.BLCK[O.VEC_DOT_TYPE[t]]
    act.reshape(n_tok, -1, blk)[:, :, 0] = -128.0 if O.VEC_DOT_TYPE[t] == O.Q8_K else -127.0     # pins d_y = 1 exactly
    w = upload_weight(lib, t, rows, k, blocks)
    want = o.mul_mat(t, blocks, k, rows, act)
    got = gpu_mul_mat(lib, U, t, w, k, rows, act)
    assert np.abs(want).max() < 2 ** 24 and np.all(want == np.rint(want))
    assert np.array_equal(got, want)
    lib.pi_weight_free(w)


@pytest.mark.parametrize("t", QTYPES + [O.F32, O.F16])
def test_get_rows_exact(gpu, t):
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    k, rows = 1024, 50
    rng = np.random.default_rng(t)
    if t == O.F32:
        blocks = rng.standard_normal((rows, k)).astype(np.float32)
    elif t == O.F16:
        blocks = O.f32_to_f16_bits(rng.standard_normal((rows, k)))
    else:
        blocks = synth.random_blocks(t, rows * k, rng, 0.05).reshape(rows, -1)
    w = upload_weight(lib, t, rows, k, blocks)
    ids = np.array([0, 49, 7, 7, 13], np.int32)
    out, idd = U.Dev(nbytes=4 * ids.size * k), U.Dev(ids)
    L.check(lib.pi_op_get_rows(w, idd.ptr, ids.size, out.ptr, None))
    assert np.array_equal(out.get(np.float32, (ids.size, k)).view(np.uint32), o.get_rows(t, blocks, k, rows, ids).view(np.uint32))
    lib.pi_weight_free(w)


@pytest.mark.parametrize("n", [256, 2048, 4096, 4544, 8192])
def test_norms(gpu, n):
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    x = np.stack([signal(n, s) for s in range(3)])
    rng = np.random.default_rng(n)
    w = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    b = (0.1 * rng.standard_normal(n)).astype(np.float32)
    xd, wd, bd = U.Dev(x), U.Dev(w), U.Dev(b)
    y = U.Dev(nbytes=x.nbytes)
    L.check(lib.pi_op_norm(xd.ptr, wd.ptr, None, y.ptr, n, 3, 1e-5, 1, None))
    want = o.rms_norm(x, 1e-5) * w
    got = y.get(np.float32, x.shape)
    assert np.abs(got - want).max() <= 2e-7 * np.abs(want).max() + 1e-12          # double-sum order: at most 1 ulp of the scale
    L.check(lib.pi_op_norm(xd.ptr, wd.ptr, bd.ptr, y.ptr, n, 3, 1e-5, 0, None))
    want = o.norm(x, 1e-5) * w + b
    got = y.get(np.float32, x.shape)
    assert np.abs(got - want).max() <= 4e-7 * np.abs(want).max() + 1e-12


@pytest.mark.parametrize("neox,hd,n_head,n_head_kv", [(0, 128, 8, 8), (0, 64, 32, 4), (1, 64, 71, 1), (1, 64, 16, 2)])
def test_rope_kv_store_exact(gpu, neox, hd, n_head, n_head_kv):
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    rng = np.random.default_rng(hd + n_head)
    n_tok, n_ctx, kv_head = 5, 64, 20
    pos = np.array([0, 1, 9, 300, 1023], np.int32)
    q = rng.standard_normal((n_tok, n_head, hd)).astype(np.float32)
    k = rng.standard_normal((n_tok, n_head_kv, hd)).astype(np.float32)
    v = rng.standard_normal((n_tok, n_head_kv * hd)).astype(np.float32)
    table = lib.pi_rope_table_create(1024, hd, 10000.0, 1.0, neox)
    assert table
    g = n_head_kv * hd
    q16, kc, vc = U.Dev(nbytes=2 * q.size), U.Dev(nbytes=2 * n_ctx * g), U.Dev(nbytes=2 * n_ctx * g)
    qd, kd, vd, pd = U.Dev(q), U.Dev(k), U.Dev(v), U.Dev(pos)      # keep the buffers alive across the launch
    L.check(lib.pi_op_rope_kv_store(qd.ptr, kd.ptr, vd.ptr, n_head * hd, n_head_kv * hd, n_head_kv * hd, pd.ptr, table, neox, hd, n_head, n_head_kv, n_tok, kv_head,
                                    q16.ptr, kc.ptr, vc.ptr, None, None))
    mode = 2 if neox else 0
    assert np.array_equal(q16.get(np.uint16, q.shape), O.f32_to_f16_bits(o.rope(q, pos, mode)))
    kc_h = kc.get(np.uint16, (n_ctx, g))
    assert np.array_equal(kc_h[kv_head:kv_head + n_tok].reshape(k.shape), O.f32_to_f16_bits(o.rope(k, pos, mode)))
    assert not kc_h[:kv_head].any() and not kc_h[kv_head + n_tok:].any()
    assert np.array_equal(vc.get(np.uint16, (n_ctx, g))[kv_head:kv_head + n_tok], O.f32_to_f16_bits(v))
    lib.pi_rope_table_free(table)


@pytest.mark.parametrize("hd,n_rot,n_head,n_head_kv", [(64, 16, 4, 2), (128, 32, 2, 2), (64, 64, 3, 1), (128, 64, 4, 1), (64, 2, 2, 2)])
def test_partial_rotary_neox_blocks_exact(gpu, hd, n_rot, n_head, n_head_kv):
    """StableLM rotates with n_dims = n_rot < head_dim in NeoX mode (llama.cpp:4860-4871): the reference walks head_dim / n_rot blocks, pairs (i, i + n_rot/2) inside a
    block, theta_scale from n_rot, theta advancing across the blocks (ggml.c:11133-11160)"""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    rng = np.random.default_rng(hd + n_rot)
    n_tok, n_ctx, kv_head = 4, 32, 7
    pos = np.array([0, 3, 77, 500], np.int32)
    q = rng.standard_normal((n_tok, n_head, hd)).astype(np.float32)
    k = rng.standard_normal((n_tok, n_head_kv, hd)).astype(np.float32)
    v = rng.standard_normal((n_tok, n_head_kv * hd)).astype(np.float32)
    table = lib.pi_rope_table_create_ex(512, hd, n_rot, 10000.0, 1.0, 1)
    assert table
    g = n_head_kv * hd
    q16, kc, vc = U.Dev(nbytes=2 * q.size), U.Dev(np.zeros(n_ctx * g, np.uint16)), U.Dev(np.zeros(n_ctx * g, np.uint16))
    qd, kd, vd, pd = U.Dev(q), U.Dev(k), U.Dev(v), U.Dev(pos)
    L.check(lib.pi_op_rope_kv_store(qd.ptr, kd.ptr, vd.ptr, n_head * hd, g, g, pd.ptr, table, 1, hd, n_head, n_head_kv, n_tok, kv_head, q16.ptr, kc.ptr, vc.ptr, None, None))
    assert np.array_equal(q16.get(np.uint16, q.shape), O.f32_to_f16_bits(o.rope_ex(q, pos, 2, n_rot)))
    assert np.array_equal(kc.get(np.uint16, (n_ctx, g))[kv_head:kv_head + n_tok].reshape(k.shape), O.f32_to_f16_bits(o.rope_ex(k, pos, 2, n_rot)))
    lib.pi_rope_table_free(table)
    # no rotation at all (ALiBi / learned positions): the identity table leaves f16(x)
    table = lib.pi_rope_table_create_ex(512, hd, hd, 10000.0, 1.0, -1)
    L.check(lib.pi_op_rope_kv_store(qd.ptr, kd.ptr, vd.ptr, n_head * hd, g, g, pd.ptr, table, 0, hd, n_head, n_head_kv, n_tok, kv_head, q16.ptr, kc.ptr, vc.ptr, None, None))
    assert np.array_equal(q16.get(np.uint16, q.shape), O.f32_to_f16_bits(q))
    assert np.array_equal(kc.get(np.uint16, (n_ctx, g))[kv_head:kv_head + n_tok].reshape(k.shape), O.f32_to_f16_bits(k))
    lib.pi_rope_table_free(table)


def test_bias_and_clamp_exact(gpu):
    """ggml_add of a projection's bias row (llama.cpp:3628-3630) and ggml_clamp (ggml.c:10893-10931)"""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    rng = np.random.default_rng(8)
    n, n_tok = 1000, 7
    x = rng.standard_normal((n_tok, n)).astype(np.float32); b = rng.standard_normal(n).astype(np.float32)
    xd, bd = U.Dev(x), U.Dev(b)
    L.check(lib.pi_op_add_bias(xd.ptr, bd.ptr, n, n_tok, None))
    assert np.array_equal(xd.get(np.float32, x.shape), x + b[None, :])
    L.check(lib.pi_op_clamp(xd.ptr, -0.75, 0.5, x.size, None))
    assert np.array_equal(xd.get(np.float32, x.shape), np.maximum(np.minimum(x + b[None, :], np.float32(0.5)), np.float32(-0.75)))


def mask_bits(mask):
    """additive 0/-inf [n_tok, n_kv] -> uint32 bitmap"""
    n_tok, n_kv = mask.shape
    words = (n_kv + 31) // 32
    bits = np.zeros((n_tok, words * 32), np.uint8)
    bits[:, :n_kv] = (mask == 0)
    return np.packbits(bits.reshape(n_tok, words, 32), axis=2, bitorder="little").view(np.uint32).reshape(n_tok, words)


@pytest.mark.parametrize("hd,n_head,n_head_kv,n_kv,n_tok,max_bias", [(64, 4, 4, 45, 3, 8.0), (128, 6, 6, 70, 2, 8.0), (64, 12, 1, 33, 1, 8.0), (128, 32, 32, 300, 9, 8.0), (64, 6, 2, 200, 5, 4.0),
                                                                      (64, 71, 1, 64, 16, 8.0), (128, 8, 8, 6000, 3, 8.0), (80, 6, 2, 70, 3, 8.0), (96, 5, 5, 130, 2, 4.0)])
def test_attention_with_alibi_vs_oracle(gpu, hd, n_head, n_head_kv, n_kv, n_tok, max_bias):
    """llm_build_kqv with max_alibi_bias > 0 (Baichuan-13B, Refact, Bloom, MPT; llama.cpp:3735-3740): score = i * m_k + kq * scale with the slopes of
    ggml_compute_forward_alibi_f32 (ggml.c:10768-10789; head counts that are not a power of two take the second series), then the mask and the soft-max"""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    rng = np.random.default_rng(hd + n_kv + n_head)
    n_ctx, kv_head = n_kv + 8, n_kv - n_tok
    g = hd * n_head_kv
    k_rows = O.f32_to_f16_bits(rng.standard_normal((n_ctx, g)))
    v_rows = O.f32_to_f16_bits(rng.standard_normal((n_ctx, g)))
    q = rng.standard_normal((n_tok, n_head, hd)).astype(np.float32)
    mask = np.zeros((n_tok, n_kv), np.float32)
    for t in range(n_tok):
        mask[t, kv_head + t + 1:] = -np.inf
    mask[:, 5:9] = -np.inf
    kc_o, vc_o = k_rows.copy(), np.ascontiguousarray(v_rows.T)
    kcur = k_rows[kv_head:kv_head + n_tok].view(np.float16).astype(np.float32).reshape(n_tok, n_head_kv, hd)
    vcur = v_rows[kv_head:kv_head + n_tok].view(np.float16).astype(np.float32)
    want = o.attention(kc_o, vc_o, n_ctx, q, kcur, vcur, mask, hd, n_head, n_head_kv, kv_head, n_kv, max_alibi_bias=max_bias)
    plain = o.attention(kc_o, vc_o, n_ctx, q, kcur, vcur, mask, hd, n_head, n_head_kv, kv_head, n_kv)
    assert not np.array_equal(want, plain)                                                          # the bias is in the numbers
    slopes = np.zeros(n_head, np.float32)
    lib.pi_alibi_slopes(n_head, max_bias, slopes.ctypes.data_as(C.c_void_p))
    q16 = O.f32_to_f16_bits(q)
    out = U.Dev(nbytes=4 * n_tok * n_head * hd)
    qd, kd, vd, md, sd = U.Dev(q16), U.Dev(k_rows), U.Dev(v_rows), U.Dev(mask_bits(mask)), U.Dev(slopes)
    L.check(lib.pi_op_attention_reserve(hd, n_kv))
    L.check(lib.pi_op_attention_alibi(qd.ptr, kd.ptr, vd.ptr, md.ptr, hd, n_head, n_head_kv, n_tok, n_kv, out.ptr, None, 0, sd.ptr, None))
    got = out.get(np.float32, want.shape)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()


@pytest.mark.parametrize("hd,n_head,n_head_kv,n_kv,n_tok", [(128, 8, 8, 45, 3), (64, 32, 4, 70, 2), (64, 71, 1, 33, 1), (128, 64, 8, 300, 9), (64, 8, 2, 1024, 4),
                                                             (64, 4, 1, 32, 7), (64, 4, 4, 32, 8), (128, 32, 32, 256, 1), (128, 16, 2, 97, 5), (64, 4, 2, 64, 16),
                                                             (128, 64, 8, 256, 9), (128, 64, 8, 300, 2), (64, 64, 8, 300, 9),    # many rows: hit f16 ties of the fp32 product
                                                             (64, 64, 8, 6000, 32), (128, 8, 8, 20000, 1), (128, 64, 8, 9000, 3),   # long contexts: fewer score rows per workgroup (8 -> 4 / 1 / 2 rows in LDS)
                                                             # other head widths (the plain kernel): 80 = StableLM-3B's (16 leftover elements per K.Q product), 96, 48, 32, 256, 100
                                                             (80, 8, 2, 70, 3), (96, 4, 4, 33, 1), (80, 32, 32, 300, 9), (48, 6, 2, 200, 5), (256, 4, 1, 45, 2), (80, 4, 2, 6000, 2), (32, 4, 4, 64, 8), (100, 2, 2, 97, 4)])
def test_attention_vs_oracle(gpu, hd, n_head, n_head_kv, n_kv, n_tok):
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    rng = np.random.default_rng(hd + n_kv)
    n_ctx, kv_head = n_kv + 8, n_kv - n_tok
    g = hd * n_head_kv
    k_rows = O.f32_to_f16_bits(rng.standard_normal((n_ctx, g)))            # [cell][n_embd_gqa]
    v_rows = O.f32_to_f16_bits(rng.standard_normal((n_ctx, g)))
    q = rng.standard_normal((n_tok, n_head, hd)).astype(np.float32)
    mask = np.zeros((n_tok, n_kv), np.float32)
    for t in range(n_tok):
        mask[t, kv_head + t + 1:] = -np.inf                                   # causal among the new tokens
    mask[:, 5:9] = -np.inf                                                    # a hole (cells of another sequence)
    # oracle: reference layouts (V transposed), no new tokens written (n_tok rows already in the cache)
    kc_o, vc_o = k_rows.copy(), np.ascontiguousarray(v_rows.T)
    kcur = k_rows[kv_head:kv_head + n_tok].view(np.float16).astype(np.float32).reshape(n_tok, n_head_kv, hd)
    vcur = v_rows[kv_head:kv_head + n_tok].view(np.float16).astype(np.float32)
    want = o.attention(kc_o, vc_o, n_ctx, q, kcur, vcur, mask, hd, n_head, n_head_kv, kv_head, n_kv)
    q16 = O.f32_to_f16_bits(q)
    out = U.Dev(nbytes=4 * n_tok * n_head * hd)
    qd, kd, vd, md = U.Dev(q16), U.Dev(k_rows), U.Dev(v_rows), U.Dev(mask_bits(mask))
    L.check(lib.pi_op_attention_reserve(hd, n_kv))
    L.check(lib.pi_op_attention(qd.ptr, kd.ptr, vd.ptr, md.ptr, hd, n_head, n_head_kv, n_tok, n_kv, out.ptr, None, None))
    got = out.get(np.float32, want.shape)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), np.abs(got - want).max()      # vec_dot_f16 order + exact soft-max sum


def test_elementwise_exact(gpu):
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    K = np.load(os.path.join(GOLD, "kernels.npz"))
    x = K["unary_x"]
    u = np.random.default_rng(3).standard_normal(x.size).astype(np.float32)
    xd, ud, out = U.Dev(x), U.Dev(u), U.Dev(nbytes=x.nbytes)
    L.check(lib.pi_op_silu_mul(xd.ptr, ud.ptr, out.ptr, x.size, None))
    assert np.array_equal(out.get(np.float32).view(np.uint32), (K["silu_y"] * u).view(np.uint32))
    L.check(lib.pi_op_gelu(xd.ptr, out.ptr, x.size, None))
    assert np.array_equal(out.get(np.float32).view(np.uint32), K["gelu_y"].view(np.uint32))
    L.check(lib.pi_op_add(xd.ptr, ud.ptr, out.ptr, x.size, None))
    assert np.array_equal(out.get(np.float32), x + u)
    # every finite fp16 input of the tables
    allx = np.arange(65536, dtype=np.uint16).view(np.float16).astype(np.float32)
    allx = allx[np.isfinite(allx)]
    ad, od = U.Dev(allx), U.Dev(nbytes=allx.nbytes)
    L.check(lib.pi_op_gelu(ad.ptr, od.ptr, allx.size, None))
    assert np.array_equal(od.get(np.float32).view(np.uint32), o.gelu(allx).view(np.uint32))
    ones = U.Dev(np.ones_like(allx))
    L.check(lib.pi_op_silu_mul(ad.ptr, ones.ptr, od.ptr, allx.size, None))
    assert np.array_equal(od.get(np.float32).view(np.uint32), o.silu(allx).view(np.uint32))


@pytest.mark.parametrize("n_vocab", [512, 32000, 65024])
def test_topk(gpu, n_vocab):
    lib, U = gpu
    from pipeinfer_amd import lib as L
    rng = np.random.default_rng(n_vocab)
    logits = (rng.standard_normal((3, n_vocab)) * 3).astype(np.float32)
    logits[1, 100] = logits[1, 7] = logits[1].max() + 1.0            # tie for the maximum: lower id first
    logits[2, 5] = -np.inf
    ids, vals, se = U.Dev(nbytes=4 * 3 * 8), U.Dev(nbytes=4 * 3 * 8), U.Dev(nbytes=4 * 3)
    ld = U.Dev(logits)
    scr = U.Dev(nbytes=lib.pi_op_topk_scratch_bytes(3))
    L.check(lib.pi_op_topk(ld.ptr, n_vocab, 3, 8, ids.ptr, vals.ptr, se.ptr, scr.ptr, None))
    gi, gv, gs = ids.get(np.int32, (3, 8)), vals.get(np.float32, (3, 8)), se.get(np.float32)
    for t in range(3):
        order = np.lexsort((np.arange(n_vocab), -logits[t]))[:8]
        assert list(gi[t]) == list(order)
        assert np.array_equal(gv[t], logits[t][order])
        want = np.exp((logits[t] - logits[t].max()).astype(np.float64)).sum()
        assert abs(gs[t] - want) / want < 1e-5
    assert gi[1, 0] == 7 and gi[1, 1] == 100
    # ... and against the oracle's restatement of llama_sample_softmax (llama.cpp:7019-7046: sorted by logit, p = expf(l - max) / sum over the FULL vocabulary in fp32) and of
    # llama_sample_token_greedy (:7496-7510: the first maximum): ids and logits exactly; probabilities to 2e-4 relative — the reference adds its 32 000 / 65 024 exponentials
    # one after the other in fp32 (sorted order; measured 7e-5 off the exact sum at 32 000 terms), the device adds them pairwise per chunk (the driver only compares p with
    # the thresholds p_accept / p_split)
    o = O.oracle()
    for t in range(3):
        oi, ol, op = np.zeros(8, np.int32), np.zeros(8, np.float32), np.zeros(8, np.float32)
        row = np.ascontiguousarray(logits[t])
        o.o_softmax_topk(row.ctypes.data, n_vocab, 8, oi.ctypes.data, ol.ctypes.data, op.ctypes.data)
        assert list(gi[t]) == list(oi) and np.array_equal(gv[t], ol)
        assert gi[t, 0] == o.o_argmax(row.ctypes.data, n_vocab)
        p_dev = np.exp(gv[t] - gv[t, 0]) / gs[t]
        assert np.all(np.abs(p_dev - op) <= 2e-4 * op + 1e-12), (p_dev, op)


@pytest.mark.parametrize("vdt,n", [(O.Q8_K, 256), (O.Q8_K, 4096), (O.Q8_K, 8192), (O.Q8_K, 11008), (O.Q8_0, 4544), (O.Q8_0, 256), (O.Q8_0, 18176), (O.Q8_0, 128), (O.Q8_0, 384), (O.Q8_0, 4224), (O.Q8_0, 32)])
def test_fused_norm_and_activation_quantisers(gpu, vdt, n):
    """norm+quantise and act+quantise launches == the separate operators (same bytes as the oracle's quantiser on the oracle's op output)"""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    n_tok = 3
    x = np.stack([signal(n, s) for s in range(n_tok)])
    rng = np.random.default_rng(n)
    w = (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
    b = (0.1 * rng.standard_normal(n)).astype(np.float32)
    xd, wd, bd = U.Dev(x), U.Dev(w), U.Dev(b)
    for rms in (1, 0):
        q8 = U.Dev(nbytes=lib.pi_q8_nbytes(vdt, n, n_tok))
        yf = U.Dev(nbytes=x.nbytes)
        L.check(lib.pi_op_norm_quantize(xd.ptr, wd.ptr, bd.ptr if not rms else None, yf.ptr, n, n_tok, 1e-5, rms, vdt, q8.ptr, None))
        want_y = (o.rms_norm(x, 1e-5) * w) if rms else (o.norm(x, 1e-5) * w + b)
        got_y = yf.get(np.float32, x.shape)
        assert np.abs(got_y - want_y).max() <= 4e-7 * np.abs(want_y).max()
        got = U.q8_as_ref_blocks(vdt, *U.q8_planes(lib, vdt, n, n_tok, q8))
        for t in range(n_tok):
            assert np.array_equal(got[t], o.quantize_act(vdt, got_y[t]))
    g = np.stack([signal(n, 10 + s) for s in range(n_tok)]); u = np.stack([signal(n, 20 + s) for s in range(n_tok)])
    gd, ud = U.Dev(g), U.Dev(u)
    for which in (0, 1):
        q8 = U.Dev(nbytes=lib.pi_q8_nbytes(vdt, n, n_tok))
        L.check(lib.pi_op_act_quantize(gd.ptr, ud.ptr if which == 0 else None, which, n, n_tok, vdt, q8.ptr, None))
        h = o.silu(g) * u if which == 0 else o.gelu(g)
        got = U.q8_as_ref_blocks(vdt, *U.q8_planes(lib, vdt, n, n_tok, q8))
        for t in range(n_tok):
            assert np.array_equal(got[t], o.quantize_act(vdt, h[t]))


@pytest.mark.parametrize("wtype", [O.Q4_K, O.Q6_K, O.Q4_0])
@pytest.mark.parametrize("n_tok", [1, 3, 8])
def test_mul_mat_multi_equals_single_launches(gpu, wtype, n_tok):
    """grouped launch (wq/wk/wv, up/gate) == one launch per weight, bit for bit"""
    import ctypes as C
    lib, U = gpu
    from pipeinfer_amd import lib as L
    from pipeinfer_amd.tools import synth
    rng = np.random.default_rng(77)
    k, ms = 1024, [512, 129, 256]
    vdt = O.VEC_DOT_TYPE[wtype]
    ws = []
    for m in ms:
        blocks = synth.random_blocks(wtype, m * k, rng, 0.05)
        ws.append(lib.pi_weight_upload(wtype, k, m, blocks.ctypes.data_as(C.c_void_p), None))
    act = rng.standard_normal((n_tok, k)).astype(np.float32)
    xd = U.Dev(act); q8 = U.Dev(nbytes=lib.pi_q8_nbytes(vdt, k, n_tok))
    L.check(lib.pi_op_quantize(vdt, xd.ptr, k, n_tok, q8.ptr, None))
    singles = [U.Dev(nbytes=4 * n_tok * m) for m in ms]
    multis = [U.Dev(nbytes=4 * n_tok * m) for m in ms]
    for w, o in zip(ws, singles):
        L.check(lib.pi_op_mul_mat(w, q8.ptr, n_tok, o.ptr, None, None))
    warr = (C.c_void_p * 3)(*ws); oarr = (C.c_void_p * 3)(*[o.ptr for o in multis])
    L.check(lib.pi_op_mul_mat_multi(3, warr, q8.ptr, n_tok, oarr, None, None))
    lib.pi_device_sync()
    for m, a, b in zip(ms, singles, multis):
        assert np.array_equal(a.get(np.float32, (n_tok, m)).view(np.uint32), b.get(np.float32, (n_tok, m)).view(np.uint32))
    for w in ws:
        lib.pi_weight_free(w)


@pytest.mark.parametrize("k,rows,n_tok", [(512, 20000, 5),       # 1 250 row groups (the smallest launches that take it), 2 blocks: one tile
                                          (768, 40008, 7),       # 2 501 row groups (odd: the last wave's second row group does not exist), 3 blocks: a last tile with one block
                                          (256, 65536, 4),       # 4 096 row groups = the most a launch of eight-wave workgroups takes; one block
                                          (256, 65552, 8),       # ... one more falls back to k_mmq_kq
                                          (1280, 18450, 3),      # 1 154 row groups, the last with 2 rows; 5 blocks
                                          (8192, 18464, 6),      # 70B row length (32 blocks: six trips of the unrolled tile loop)
                                          (2816, 24000, 8)])     # 11 blocks (odd)
def test_mmq_two_row_groups_per_wave_vs_oracle(gpu, k, rows, n_tok):
    """k_mmq_g2 (mmq_mfma.hip): Q4_K launches of > 1152 row groups at 3..8 tokens — two row groups per wave, packed accumulators (accumulator lanes k and k + 4 share an MFMA
    tile through zero token rows), hand-counted weight ring, fold in registers — give the oracle's (= the reference's, ggml-quants.c:5318-5381) fp32 bits"""
    from pipeinfer_amd.tools import synth
    lib, U = gpu
    rng = np.random.default_rng(k + rows + n_tok)
    blocks = synth.random_blocks(O.Q4_K, rows * k, rng, 0.05).reshape(rows, -1)
    act = (rng.standard_normal((n_tok, k)) * rng.uniform(0.2, 3.0, (n_tok, 1))).astype(np.float32)
    res = rng.standard_normal((n_tok, rows)).astype(np.float32)
    w = upload_weight(lib, O.Q4_K, rows, k, blocks)
    want = O.Impl("oracle").mul_mat(O.Q4_K, blocks, k, rows, act)
    got = gpu_mul_mat(lib, U, O.Q4_K, w, k, rows, act)
    assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), float(np.abs(got - want).max())
    got = gpu_mul_mat(lib, U, O.Q4_K, w, k, rows, act, residual=res)                                    # ffn_down-style launch: residual + product (one fp32 add)
    assert np.array_equal(got.view(np.uint32), (res + want).view(np.uint32))
    lib.pi_weight_free(w)


@pytest.mark.parametrize("n_tok", [3, 6, 8])
def test_mmq_two_row_groups_per_wave_fused_up_gate(gpu, n_tok):
    """the launch the kernel was built for: ffn_up and ffn_gate of a layer as ONE launch (pi_op_mul_mat_multi, two segments; a row group never straddles them) — each
    segment's output equals its own single launch and the oracle"""
    import ctypes as C
    from pipeinfer_amd import lib as L
    from pipeinfer_amd.tools import synth
    lib, U = gpu
    rng = np.random.default_rng(900 + n_tok)
    k, ms = 1024, [14336, 14344]                                                                         # 896 + 897 row groups (the second segment ends in a half group)
    ws, bl = [], []
    for m in ms:
        b = synth.random_blocks(O.Q4_K, m * k, rng, 0.05).reshape(m, -1)
        bl.append(b); ws.append(upload_weight(lib, O.Q4_K, m, k, b))
    act = rng.standard_normal((n_tok, k)).astype(np.float32)
    q8 = gpu_quantize(lib, U, O.Q8_K, act)
    outs = [U.Dev(nbytes=4 * n_tok * m) for m in ms]
    warr = (C.c_void_p * 2)(*ws); oarr = (C.c_void_p * 2)(*[o.ptr for o in outs])
    L.check(lib.pi_op_mul_mat_multi(2, warr, q8.ptr, n_tok, oarr, None, None))
    lib.pi_device_sync()
    for m, b, o in zip(ms, bl, outs):
        want = O.Impl("oracle").mul_mat(O.Q4_K, b, k, m, act)
        got = o.get(np.float32, (n_tok, m))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (m, float(np.abs(got - want).max()))
    for w in ws:
        lib.pi_weight_free(w)


@pytest.mark.parametrize("types", [(O.Q4_K, O.Q4_K, O.Q4_K), (O.Q4_K, O.Q4_K, O.Q6_K), (O.Q5_K, O.Q6_K, O.Q4_K)])
@pytest.mark.parametrize("hd,n_head,n_head_kv,n_tok,kv_head", [(128, 8, 8, 1, 5), (64, 8, 2, 3, 0), (128, 4, 1, 8, 17), (64, 4, 4, 11, 2),
                                                               (128, 64, 8, 1, 3), (128, 64, 8, 2, 9), (128, 64, 8, 4, 1), (128, 40, 40, 1, 0),    # K = 8192 / 5120: the tile-per-wave instance with the RoPE epilogue
                                                               (64, 4, 4, 20, 2), (128, 64, 8, 32, 1), (64, 8, 2, 27, 5)])                            # two chunks of 16 per launch of the matrix-core kernel
def test_qkv_rope_epilogue_equals_separate_launches(gpu, types, hd, n_head, n_head_kv, n_tok, kv_head):
    """RoPE + f16 rounding + KV-cache writes done by the projection's own waves == mat-vecs into fp32 + pi_op_rope_kv_store"""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    rng = np.random.default_rng(hd + n_tok)
    k, n_ctx = 512, 40
    n_embd, n_gqa = n_head * hd, n_head_kv * hd
    ws = []
    for t, m in zip(types, (n_embd, n_gqa, n_gqa)):
        blocks = synth.random_blocks(t, m * k, rng, 0.05)
        ws.append(lib.pi_weight_upload(t, k, m, blocks.ctypes.data_as(C.c_void_p), None))
    assert lib.pi_op_mul_mat_qkv_rope_ok(ws[0], ws[1], ws[2], 0, hd) == 1
    act = rng.standard_normal((n_tok, k)).astype(np.float32)
    pos = rng.integers(0, 200, n_tok).astype(np.int32)
    table = lib.pi_rope_table_create(256, hd, 10000.0, 1.0, 0)
    xd, pd = U.Dev(act), U.Dev(pos)
    q8 = U.Dev(nbytes=lib.pi_q8_nbytes(O.Q8_K, k, n_tok))
    L.check(lib.pi_op_quantize(O.Q8_K, xd.ptr, k, n_tok, q8.ptr, None))
    qf, kf, vf = U.Dev(nbytes=4 * n_tok * n_embd), U.Dev(nbytes=4 * n_tok * n_gqa), U.Dev(nbytes=4 * n_tok * n_gqa)
    for w, o in zip(ws, (qf, kf, vf)):
        L.check(lib.pi_op_mul_mat(w, q8.ptr, n_tok, o.ptr, None, None))
    dyn = U.Dev(np.array([kv_head, 32, n_tok, 0], np.int32))
    res = []
    for fused in (False, True):
        q16 = U.Dev(np.zeros(n_tok * n_embd, np.uint16)); kc = U.Dev(np.zeros(n_ctx * n_gqa, np.uint16)); vc = U.Dev(np.zeros(n_ctx * n_gqa, np.uint16))
        if fused:
            L.check(lib.pi_op_mul_mat_qkv_rope(ws[0], ws[1], ws[2], q8.ptr, n_tok, pd.ptr, table, hd, n_head, n_head_kv, -1, dyn.ptr, q16.ptr, kc.ptr, vc.ptr, None))
        else:
            L.check(lib.pi_op_rope_kv_store(qf.ptr, kf.ptr, vf.ptr, n_embd, n_gqa, n_gqa, pd.ptr, table, 0, hd, n_head, n_head_kv, n_tok, kv_head, q16.ptr, kc.ptr, vc.ptr, None, None))
        lib.pi_device_sync()
        res.append((q16.get(np.uint16), kc.get(np.uint16), vc.get(np.uint16)))
    for a, b, name in zip(res[0], res[1], "qkv"):
        assert np.array_equal(a, b), name
    assert res[0][1].any() and res[0][2].any()
    lib.pi_rope_table_free(table)
    for w in ws:
        lib.pi_weight_free(w)


@pytest.mark.parametrize("types", [(O.Q4_K, O.Q4_K, O.Q5_K), (O.Q4_K, O.Q4_K, O.Q6_K), (O.Q4_K, O.Q6_K, O.Q6_K), (O.Q5_K, O.Q4_K, O.Q4_K)])
@pytest.mark.parametrize("k,n_tok", [(2048, 1), (4096, 1), (4096, 2), (8192, 1), (8192, 2), (5120, 1), (16384, 2)])
def test_two_weight_types_share_one_launch(gpu, types, k, n_tok):
    """attn_q / attn_k in Q4_K with attn_v in Q5_K or Q6_K — the Q4_K_M mixture (llama.cpp:7923-7933) — go out as ONE launch at decode widths (k_mmvq_kq2: the segment's
    type is chosen per wave / per workgroup); the bits are those of the oracle's vec_dot per matrix, RoPE and the cache writes those of the separate operators.
    Rows of one tile (K <= 4096), of two to four tiles (the tile-per-wave instance) and a ragged last tile (K = 5120)."""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    rng = np.random.default_rng(k + n_tok)
    hd, n_head, n_head_kv, n_ctx, kv_head = 128, 6, 2, 24, 7
    n_embd, n_gqa = n_head * hd, n_head_kv * hd
    ws, blocks = [], []
    for t, m in zip(types, (n_embd, n_gqa, n_gqa)):
        b = synth.random_blocks(t, m * k, rng, 0.05).reshape(m, -1)
        blocks.append(b)
        ws.append(upload_weight(lib, t, m, k, b))
    assert lib.pi_op_mul_mat_qkv_rope_ok(ws[0], ws[1], ws[2], 0, hd) == 1
    act = rng.standard_normal((n_tok, k)).astype(np.float32)
    pos = rng.integers(0, 200, n_tok).astype(np.int32)
    table = lib.pi_rope_table_create(256, hd, 10000.0, 1.0, 0)
    xd, pd = U.Dev(act), U.Dev(pos)
    q8 = U.Dev(nbytes=lib.pi_q8_nbytes(O.Q8_K, k, n_tok))
    L.check(lib.pi_op_quantize(O.Q8_K, xd.ptr, k, n_tok, q8.ptr, None))
    # (1) plain multi launch with fp32 outputs: every matrix equals the oracle bit for bit
    outs = [U.Dev(nbytes=4 * n_tok * m) for m in (n_embd, n_gqa, n_gqa)]
    wl = (C.c_void_p * 3)(*ws); ol = (C.c_void_p * 3)(*[o.ptr for o in outs])
    L.check(lib.pi_op_mul_mat_multi(3, wl, q8.ptr, n_tok, ol, None, None))
    o = O.Impl("oracle")
    for t, m, b, dv in zip(types, (n_embd, n_gqa, n_gqa), blocks, outs):
        want = o.mul_mat(t, b, k, m, act)
        got = dv.get(np.float32, (n_tok, m))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (O.TYPE_NAMES[t], float(np.abs(got - want).max()))
    # (2) with the RoPE / KV-store epilogue == the separate operators on those fp32 outputs
    dyn = U.Dev(np.array([kv_head, 32, n_tok, 0], np.int32))
    res = []
    for fused in (False, True):
        q16 = U.Dev(np.zeros(n_tok * n_embd, np.uint16)); kc = U.Dev(np.zeros(n_ctx * n_gqa, np.uint16)); vc = U.Dev(np.zeros(n_ctx * n_gqa, np.uint16))
        if fused:
            L.check(lib.pi_op_mul_mat_qkv_rope(ws[0], ws[1], ws[2], q8.ptr, n_tok, pd.ptr, table, hd, n_head, n_head_kv, -1, dyn.ptr, q16.ptr, kc.ptr, vc.ptr, None))
        else:
            L.check(lib.pi_op_rope_kv_store(outs[0].ptr, outs[1].ptr, outs[2].ptr, n_embd, n_gqa, n_gqa, pd.ptr, table, 0, hd, n_head, n_head_kv, n_tok, kv_head, q16.ptr, kc.ptr, vc.ptr, None, None))
        lib.pi_device_sync()
        res.append((q16.get(np.uint16), kc.get(np.uint16), vc.get(np.uint16)))
    for a, b, name in zip(res[0], res[1], "qkv"):
        assert np.array_equal(a, b), name
    assert res[0][1].any() and res[0][2].any()
    lib.pi_rope_table_free(table)
    for w in ws:
        lib.pi_weight_free(w)


@pytest.mark.parametrize("types", [(O.Q4_K, O.Q4_K, O.Q5_K), (O.Q4_K, O.Q4_K, O.Q6_K), (O.Q4_K, O.Q6_K, O.Q6_K)])
@pytest.mark.parametrize("k,n_tok,n_head,n_head_kv", [(2048, 3, 48, 2), (8192, 6, 48, 4), (4096, 8, 40, 8), (5120, 5, 32, 32), (8192, 4, 64, 8),
                                                       # prompt passes: launches of 64 tokens (four chunks, register-staged activations), the last one with what is left
                                                       (4096, 64, 48, 4), (2048, 40, 40, 8), (2048, 100, 48, 2), (1024, 130, 16, 16)])
def test_two_weight_types_share_one_matrix_core_launch(gpu, types, k, n_tok, n_head, n_head_kv):
    """The same mixture at verify widths (3..8 tokens): attn_v (Q5_K / Q6_K) rides in the q + k matrix-core launch (k_mmq_kq2t: the first workgroups walk the Q4_K row
    groups, the last ones the second type's, each with its own tile format) instead of being a launch of its own behind it.  Bits: the oracle's vec_dot per matrix
    (ggml-quants.c:5318-5381, 5967-6050, 6741-6810); RoPE and the cache writes those of the separate operators.  Shapes: 70B-like GQA (64 / 8 heads of 128), MHA, a
    ragged row count per workgroup, K = 5120 (a last tile that is not full)."""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    rng = np.random.default_rng(k + 7 * n_tok)
    hd, kv_head = 128, 7
    n_ctx = max(24, kv_head + n_tok + 1)
    n_embd, n_gqa = n_head * hd, n_head_kv * hd
    ws, blocks = [], []
    for t, m in zip(types, (n_embd, n_gqa, n_gqa)):
        b = synth.random_blocks(t, m * k, rng, 0.05).reshape(m, -1)
        blocks.append(b)
        ws.append(upload_weight(lib, t, m, k, b))
    act = rng.standard_normal((n_tok, k)).astype(np.float32)
    pos = rng.integers(0, 200, n_tok).astype(np.int32)
    table = lib.pi_rope_table_create(256, hd, 10000.0, 1.0, 0)
    xd, pd = U.Dev(act), U.Dev(pos)
    q8 = U.Dev(nbytes=lib.pi_q8_nbytes(O.Q8_K, k, n_tok))
    L.check(lib.pi_op_quantize(O.Q8_K, xd.ptr, k, n_tok, q8.ptr, None))
    outs = [U.Dev(nbytes=4 * n_tok * m) for m in (n_embd, n_gqa, n_gqa)]
    wl = (C.c_void_p * 3)(*ws); ol = (C.c_void_p * 3)(*[o.ptr for o in outs])
    lib.pi_kernel_timing(1, 1 << 30)                                   # count the launches: ONE matrix-core launch for the three matrices
    L.check(lib.pi_op_mul_mat_multi(3, wl, q8.ptr, n_tok, ol, None, None))
    lib.pi_device_sync()
    import bench as B
    rows = [r for r in B.timing_rows(lib) if r["family"] <= 1]
    lib.pi_kernel_timing(0, 8)
    # one launch while both parts fit one round of workgroups (the dispatcher's rule, mmq_mfma.hip: pi_mmq2_applies); otherwise one launch per type, as before
    n_a = sum(1 for t in types if t == O.Q4_K)
    groups_a = sum((m + 15) // 16 for m in (n_embd, n_gqa, n_gqa)[:n_a]); groups_b = sum((m + 15) // 16 for m in (n_embd, n_gqa, n_gqa)[n_a:])
    r = 0 if groups_a > 768 else 3 if groups_a > 512 else 2 if groups_a >= 384 else 1
    one_round = r > 0 and ((groups_a + r - 1) // r + groups_b <= 256 or (groups_a + r - 1) // r + (groups_b + r - 1) // r <= 256)
    if n_tok > 32:
        assert (all(r_["family"] == 1 for r_ in rows) and sum(r_["launches"] for r_ in rows) == (n_tok + 63) // 64) if r > 0 else sum(r_["launches"] for r_ in rows) >= 2, rows
    elif one_round:
        assert len(rows) == 1 and rows[0]["family"] == 1 and rows[0]["launches"] == 1, rows
    else:
        assert sum(r_["launches"] for r_ in rows) >= 2, rows
    o = O.Impl("oracle")
    for t, m, b, dv in zip(types, (n_embd, n_gqa, n_gqa), blocks, outs):
        want = o.mul_mat(t, b, k, m, act)
        got = dv.get(np.float32, (n_tok, m))
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), (O.TYPE_NAMES[t], float(np.abs(got - want).max()))
    dyn = U.Dev(np.array([kv_head, 32, n_tok, 0], np.int32))
    res = []
    for fused in (False, True):
        q16 = U.Dev(np.zeros(n_tok * n_embd, np.uint16)); kc = U.Dev(np.zeros(n_ctx * n_gqa, np.uint16)); vc = U.Dev(np.zeros(n_ctx * n_gqa, np.uint16))
        if fused:
            L.check(lib.pi_op_mul_mat_qkv_rope(ws[0], ws[1], ws[2], q8.ptr, n_tok, pd.ptr, table, hd, n_head, n_head_kv, -1, dyn.ptr, q16.ptr, kc.ptr, vc.ptr, None))
        else:
            L.check(lib.pi_op_rope_kv_store(outs[0].ptr, outs[1].ptr, outs[2].ptr, n_embd, n_gqa, n_gqa, pd.ptr, table, 0, hd, n_head, n_head_kv, n_tok, kv_head, q16.ptr, kc.ptr, vc.ptr, None, None))
        lib.pi_device_sync()
        res.append((q16.get(np.uint16), kc.get(np.uint16), vc.get(np.uint16)))
    for a, b, name in zip(res[0], res[1], "qkv"):
        assert np.array_equal(a, b), name
    assert res[0][1].any() and res[0][2].any()
    lib.pi_rope_table_free(table)
    for w in ws:
        lib.pi_weight_free(w)


def test_abort_word_stops_kernels_of_the_cancelled_run_only(gpu):
    """Kernel-granular early cancellation (reference: the abort callback polled between graph nodes, ggml.c:15748-15811, llama.cpp:5665-5687): every hot-path
    kernel launched under pi_set_abort_word reads (run id << 1 | aborted) on entry.  pi_op_abort_cancel is a compare-and-swap on the run id: it stops the run that
    is executing, never another one; a run that is not aborted computes the oracle's bits."""
    from pipeinfer_amd import lib as L
    lib, U = gpu
    t, k, rows, n_tok = O.Q4_K, 4096, 256, 2
    rng = np.random.default_rng(5)
    blocks = synth.random_blocks(t, rows * k, rng, 0.05).reshape(rows, -1)
    act = rng.standard_normal((n_tok, k)).astype(np.float32)
    want = O.Impl("oracle").mul_mat(t, blocks, k, rows, act)
    w = upload_weight(lib, t, rows, k, blocks)
    word = U.Dev(np.zeros(16, np.int32))                               # the abort block: word 0 = (run id << 1 | aborted), words 8..14 = launches aborted on entry, per kernel family
    xd = U.Dev(act)
    sentinel = np.full((n_tok, rows), 12345.0, np.float32)

    def run(expect_computed, before):
        """quantise + mat-vec + a norm under the abort word, after `before` has set the word up; the outputs either stay the sentinel or are the oracle's"""
        q8 = U.Dev(nbytes=lib.pi_q8_nbytes(O.Q8_K, k, n_tok)); out = U.Dev(sentinel); y = U.Dev(np.full((n_tok, k), 777.0, np.float32))
        before()
        lib.pi_set_abort_word(word.ptr)
        try:
            L.check(lib.pi_op_quantize(O.Q8_K, xd.ptr, k, n_tok, q8.ptr, None))
            L.check(lib.pi_op_mul_mat(w, q8.ptr, n_tok, out.ptr, None, None))
            L.check(lib.pi_op_norm(xd.ptr, None, None, y.ptr, k, n_tok, 1e-5, 1, None))
        finally:
            lib.pi_set_abort_word(None)
        got, gy = out.get(np.float32, (n_tok, rows)), y.get(np.float32, (n_tok, k))
        if expect_computed:
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)) and not np.any(gy == 777.0)
        else:
            assert np.array_equal(got, sentinel) and np.all(gy == 777.0), "kernels of an aborted run must return before they write anything"

    flag = np.zeros(1, np.int32)                                        # (the engine keeps these flags in pinned host memory; NULL = no flag to pick up)
    run(True, lambda: L.check(lib.pi_op_abort_begin(word.ptr, 7, None, None)))
    assert word.get(np.int32)[0] == 7 << 1
    run(True, lambda: L.check(lib.pi_op_abort_cancel(word.ptr, 6, None)))                       # a cancel for ANOTHER run (one that is not executing) changes nothing
    assert word.get(np.int32)[0] == 7 << 1
    run(False, lambda: L.check(lib.pi_op_abort_cancel(word.ptr, 7, None)))                      # the executing run: stopped
    assert word.get(np.int32)[0] == (7 << 1) | 1
    run(False, lambda: None)                                                                    # ... and stays stopped: every later kernel of it returns on entry
    # the two aborted rounds above are on the device's books, per family (pi_kt_row.family: 0 mat-vec, 3 quantise, 6 other = k_norm): what bench.py takes out of the roofline
    counters = word.get(np.int32)[8:15]
    assert counters[0] == 2 and counters[3] == 2 and counters[6] == 2 and counters[1] == counters[2] == counters[4] == counters[5] == 0, counters
    run(True, lambda: L.check(lib.pi_op_abort_begin(word.ptr, 8, None, None)))                  # the next run starts clean
    run(True, lambda: L.check(lib.pi_op_abort_cancel(word.ptr, 7, None)))                       # a late cancel of the old run cannot touch it
    assert word.get(np.int32)[0] == 8 << 1
    assert np.array_equal(word.get(np.int32)[8:15], counters), "launches that did the work are not counted"
    lib.pi_weight_free(w)
    del flag


def test_persimmon_attention_inputs_and_relu_sqr_exact(gpu):
    """build_persimmon (llama.cpp:4372-4460): fused QKV rows laid out [head][q | k | v][64], a LayerNorm of Q and of K over the 64 values of their head (sequential double
    sums like ggml_compute_forward_norm_f32), NeoX RoPE with n_dims = 32 over a view of the first 32 values while the other 32 pass, f16 rounding into the query buffer and
    the cache rows; and its activation relu(x)^2 (LLM_FFN_RELU_SQR) in front of the Q8_K quantiser"""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    o = O.Impl("oracle")
    rng = np.random.default_rng(64)
    n_tok, n_head, hd, n_ctx, kv_head = 5, 6, 64, 40, 11
    n_embd = n_head * hd
    qkv = rng.standard_normal((n_tok, n_head, 3, hd)).astype(np.float32)
    qw, qb, kw, kb = [(s0 + 0.1 * rng.standard_normal(hd)).astype(np.float32) for s0 in (1.0, 0.0, 1.0, 0.0)]
    pos = np.array([0, 2, 9, 130, 400], np.int32)
    eps = 1e-5

    def want(x, w, b):                                                      # [n_tok, n_head, hd]
        y = o.norm(x.reshape(-1, hd), eps).reshape(x.shape) * w + b
        out = y.copy()
        out[:, :, :32] = o.rope_ex(np.ascontiguousarray(y[:, :, :32]), pos, 2, 32)
        return O.f32_to_f16_bits(out)
    table = lib.pi_rope_table_create_ex(512, 32, 32, 10000.0, 1.0, 1)
    q16, kc, vc = U.Dev(nbytes=2 * n_tok * n_embd), U.Dev(np.zeros(n_ctx * n_embd, np.uint16)), U.Dev(np.zeros(n_ctx * n_embd, np.uint16))
    xd, pd = U.Dev(qkv), U.Dev(pos)
    wd = [U.Dev(a) for a in (qw, qb, kw, kb)]
    L.check(lib.pi_op_persimmon_qkv_store(xd.ptr, n_head, hd, wd[0].ptr, wd[1].ptr, wd[2].ptr, wd[3].ptr, eps, pd.ptr, table, n_tok, kv_head, q16.ptr, kc.ptr, vc.ptr, None, None))
    assert np.array_equal(q16.get(np.uint16, (n_tok, n_head, hd)), want(qkv[:, :, 0], qw, qb))
    assert np.array_equal(kc.get(np.uint16, (n_ctx, n_head, hd))[kv_head:kv_head + n_tok], want(qkv[:, :, 1], kw, kb))
    assert np.array_equal(vc.get(np.uint16, (n_ctx, n_head, hd))[kv_head:kv_head + n_tok], O.f32_to_f16_bits(qkv[:, :, 2]))
    lib.pi_rope_table_free(table)
    n = 1024
    g = np.stack([signal(n, 30 + s0) for s0 in range(3)])
    gd = U.Dev(g)
    q8 = U.Dev(nbytes=lib.pi_q8_nbytes(O.Q8_K, n, 3))
    L.check(lib.pi_op_act_quantize(gd.ptr, None, 2, n, 3, O.Q8_K, q8.ptr, None))
    h = np.maximum(g, np.float32(0.0)); h = h * h
    got = U.q8_as_ref_blocks(O.Q8_K, *U.q8_planes(lib, O.Q8_K, n, 3, q8))
    for t in range(3):
        assert np.array_equal(got[t], o.quantize_act(O.Q8_K, h[t]))


@pytest.mark.parametrize("n_tok,nv,skew", [(1, 32000, 0), (3, 1001, 0), (2, 4096, 1), (5, 259, 3)])
def test_copy_out_writes_the_page_locked_buffers(gpu, n_tok, nv, skew):
    """pi_op_copy_out (runtime.hip): a run's logits rows, top-8 ids / values and soft-max sums written by a kernel into page-locked host memory are the device's bytes —
    also when a row count or a row offset keeps the logits off the 16-byte vector path (skew: floats the device pointer starts past an aligned address)."""
    lib, U = gpu
    from pipeinfer_amd import lib as L
    hip = C.CDLL("libamdhip64.so")
    hip.hipHostMalloc.argtypes = [C.POINTER(C.c_void_p), C.c_size_t, C.c_uint]; hip.hipHostFree.argtypes = [C.c_void_p]
    lib.pi_op_copy_out.argtypes = [C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]
    rng = np.random.default_rng(n_tok * 7919 + nv)
    n = n_tok * nv
    logits = rng.standard_normal(n + skew).astype(np.float32)
    ids = rng.integers(0, nv, 8 * n_tok).astype(np.int32); vals = rng.standard_normal(8 * n_tok).astype(np.float32); sums = rng.standard_normal(n_tok).astype(np.float32)
    d_log, d_ids, d_vals, d_sum = U.Dev(logits), U.Dev(ids), U.Dev(vals), U.Dev(sums)
    nbytes = 4 * n + 16 + 4 * 17 * n_tok + 64
    host = C.c_void_p()
    assert hip.hipHostMalloc(C.byref(host), nbytes, 0) == 0
    try:
        C.memset(host, 0xA5, nbytes)
        h_log, h_ids = host.value, host.value + ((4 * n + 15) // 16) * 16
        h_vals, h_sum = h_ids + 32 * n_tok, h_ids + 64 * n_tok
        L.check(lib.pi_op_copy_out(d_log.ptr + 4 * skew, h_log, n, d_ids.ptr, h_ids, d_vals.ptr, h_vals, d_sum.ptr, h_sum, n_tok, None), "pi_op_copy_out")
        L.check(lib.pi_stream_sync(None))
        got = lambda p, cnt, dt: np.frombuffer((C.c_char * (cnt * 4)).from_address(p), dtype=dt).copy()
        assert np.array_equal(got(h_log, n, np.uint32), logits[skew:].view(np.uint32))
        assert np.array_equal(got(h_ids, 8 * n_tok, np.int32), ids) and np.array_equal(got(h_vals, 8 * n_tok, np.uint32), vals.view(np.uint32)) and np.array_equal(got(h_sum, n_tok, np.uint32), sums.view(np.uint32))
        tail = np.frombuffer((C.c_char * 16).from_address(h_sum + 4 * n_tok), dtype=np.uint8)
        assert (tail == 0xA5).all(), "the kernel wrote past the buffers it was given"
    finally:
        hip.hipHostFree(host)
