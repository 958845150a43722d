import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import gpu_util as U, oracle_lib as O
from tests.test_gpu_ops import upload_weight, gpu_mul_mat, signal, mask_bits
from pipeinfer_amd.tools import synth
from pipeinfer_amd import lib as L
lib = U.init()
o = O.Impl("oracle")
bad = 0
for t in [O.Q4_0, O.Q8_0, O.Q4_K, O.Q5_K, O.Q6_K]:
    for k in (256, 512, 768, 1024):
        for rows in (64, 384, 512):
            rng = np.random.default_rng(k + rows)
            blocks = synth.random_blocks(t, rows * k, rng, 0.05).reshape(rows, -1)
            w = upload_weight(lib, t, rows, k, blocks)
            for n_tok in (1, 2, 4, 7, 8):
                act = np.stack([signal(k, s + n_tok) for s in range(n_tok)])
                want = o.mul_mat(t, blocks, k, rows, act)
                got = gpu_mul_mat(lib, U, t, w, k, rows, act)
                err = np.abs(got - want).max() / np.abs(want).max()
                if err > 2e-5:
                    bad += 1; print("MMVQ BAD", O.TYPE_NAMES[t], k, rows, n_tok, err)
            lib.pi_weight_free(w)
print("mmvq sweep done, bad =", bad)
for (hd, n_head, n_head_kv, n_kv, n_tok) in [(64, 4, 1, 32, 7), (64, 4, 2, 32, 1), (64, 4, 4, 32, 1), (64, 4, 2, 32, 7), (64, 4, 1, 32, 1), (64, 4, 4, 32, 8)]:
    rng = np.random.default_rng(hd + n_kv + n_tok)
    n_ctx, kv_head = 128, 22
    g = hd * n_head_kv
    k_rows = O.f32_to_f16_bits(rng.standard_normal((n_ctx, g))); v_rows = O.f32_to_f16_bits(rng.standard_normal((n_ctx, g)))
    q = rng.standard_normal((n_tok, n_head, hd)).astype(np.float32)
    mask = np.full((n_tok, n_kv), -np.inf, np.float32)
    for t in range(n_tok):
        mask[t, :kv_head + t + 1] = 0
    kc_o, vc_o = k_rows.copy(), np.ascontiguousarray(v_rows.T)
    kcur = k_rows[kv_head:kv_head + n_tok].view(np.float16).astype(np.float32).reshape(n_tok, n_head_kv, hd)
    vcur = v_rows[kv_head:kv_head + n_tok].view(np.float16).astype(np.float32)
    want = o.attention(kc_o, vc_o, n_ctx, q, kcur, vcur, mask, hd, n_head, n_head_kv, kv_head, n_kv)
    q16 = O.f32_to_f16_bits(q)
    out = U.Dev(nbytes=4 * n_tok * n_head * hd)
    qd, kd, vd, md = U.Dev(q16), U.Dev(k_rows), U.Dev(v_rows), U.Dev(mask_bits(mask))
    L.check(lib.pi_op_attention(qd.ptr, kd.ptr, vd.ptr, md.ptr, hd, n_head, n_head_kv, n_tok, n_kv, out.ptr, None, None))
    got = out.get(np.float32, want.shape)
    print("ATT", hd, n_head, n_head_kv, n_kv, n_tok, "err", np.abs(got - want).max())
