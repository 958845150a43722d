import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tests import oracle_lib as O, gpu_util as U
from tests.test_gpu_ops import mask_bits
from pipeinfer_amd import lib as L
lib = U.init()
hd, n_head, n_head_kv, n_kv, n_tok = [int(x) for x in sys.argv[1:6]]
o = O.Impl("oracle")
rng = np.random.default_rng(hd + n_kv)
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
want = o.attention(kc_o, vc_o, n_ctx, q, kcur, vcur, mask, hd, n_head, n_head_kv, kv_head, n_kv)
q16 = O.f32_to_f16_bits(q)
out = U.Dev(nbytes=4 * n_tok * n_head * hd)
qd, kd, vd, md = U.Dev(q16), U.Dev(k_rows), U.Dev(v_rows), U.Dev(mask_bits(mask))
probe = U.Dev(nbytes=4 * 2 * n_head * n_tok * n_kv + 16 * n_head * n_tok)
lib.pi_debug_attention_probe.argtypes = [__import__('ctypes').c_void_p]
lib.pi_debug_attention_probe(probe.ptr)
L.check(lib.pi_op_attention_reserve(hd, n_kv))
L.check(lib.pi_op_attention(qd.ptr, kd.ptr, vd.ptr, md.ptr, hd, n_head, n_head_kv, n_tok, n_kv, out.ptr, None, None))
got = out.get(np.float32, want.shape).reshape(n_tok, n_head, hd)
want = want.reshape(n_tok, n_head, hd)
bad = got.view(np.uint32) != want.view(np.uint32)
print("mismatching elements", bad.sum(), "of", bad.size, "max abs", np.abs(got - want).max())
print("per token", bad.sum(axis=(1, 2)))
print("per head", bad.sum(axis=(0, 2)))
print("per dim", bad.sum(axis=(0, 1)))

pr = probe.get(np.float32, (2, n_head, n_tok, n_kv), 0, 2 * n_head * n_tok * n_kv)
extra = probe.get(np.float32, (n_head, n_tok, 4), 4 * 2 * n_head * n_tok * n_kv, n_head * n_tok * 4)
r = n_head // n_head_kv
ti, hi = np.argwhere(bad.any(axis=2))[0]
print("first bad (token, head)", ti, hi)
kf = k_rows.reshape(n_ctx, n_head_kv, hd)
sc = np.array([o.vec_dot(O.F16, hd, kf[c, hi // r], q16[ti, hi]) for c in range(n_kv)], np.float32)
scale = np.float32(1.0 / np.sqrt(np.float32(hd)))
want_s = (sc * scale + mask[ti]).astype(np.float32)
got_s = pr[0, hi, ti]
ds = np.where((want_s.view(np.uint32) != got_s.view(np.uint32)))[0]
print("score mismatches at cells", ds[:20], "of", len(ds))
for c in ds[:5]: print("  cell", c, "want", want_s[c], "got", got_s[c], (want_s[c] - got_s[c]))
want_p = o.soft_max(want_s[None])[0]
want_p16 = O.f32_to_f16_bits(want_p).view(np.float16).astype(np.float32)
dp = np.where(want_p16 != pr[1, hi, ti])[0]
print("prob mismatches at cells", dp[:20], "of", len(dp))
for c in dp[:3]:
    print("cell", c, "want_p f32 %.10e" % want_p[c], "want_p16 %.10e" % want_p16[c], "got %.10e" % pr[1, hi, ti, c], "score", want_s[c], "max", want_s.max())
    h = np.float32(want_p[c]).astype(np.float16)
    print("   f16 neighbours", np.nextafter(h, np.float16(0)).astype(np.float32), h.astype(np.float32), np.nextafter(h, np.float16(1)).astype(np.float32))
# emulate: table exp via the oracle soft_max on [x, max] pairs is awkward; use sums
vals = want_p / want_p.max()
print("sum of want_p (f64)", want_p.astype(np.float64).sum())

import ctypes as C
o.lib.o_tables_init(); o.lib.o_table_exp_f16.restype = C.POINTER(C.c_uint16)
tab = np.ctypeslib.as_array(o.lib.o_table_exp_f16(), (65536,)).copy()
mxr = want_s.max()
arg = O.f32_to_f16_bits((want_s - mxr).astype(np.float32))
v = tab[arg].view(np.float16).astype(np.float32); v[np.isinf(want_s)] = 0
sum_ref = v.astype(np.float64).sum(); inv_ref = np.float32(1.0 / sum_ref)
gi = extra[hi, ti]
print("inv ref %.9e gpu %.9e  sum ref %.12f gpu %.12f  max ref %.8f gpu %.8f" % (inv_ref, gi[0], sum_ref, (gi[1] * 16777216.0 + gi[2]) / 16777216.0, mxr, gi[3]))
pp = (v * inv_ref).astype(np.float32)
print("p_ref[c] %.10e" % pp[dp[0]], "v", v[dp[0]])
