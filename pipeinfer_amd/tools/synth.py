"""Shape-exact synthetic GGUF models (no network => no real checkpoints).

Tensor names/shapes/types are exactly what the reference loader expects for each architecture
(llama.cpp:2756-2824 Llama, :2891-2968 Falcon) and what its quantiser's mixture rules would emit
(get_k_quant_type, llama.cpp:7896-8006).  Block *contents* are synthetic:

  structure="random": every weight block is filled directly with random quants and a scale chosen so the dequantised
                      weights are ~zero-mean with the requested std (fast: no quantiser pass over 7e9 floats).
  structure="bigram": as "random", but token_embd/output encode a fixed random permutation pi (next = pi[cur]) and
                      the per-layer output projections are scaled by `noise`, so draft and target of DIFFERENT
                      shapes agree on most greedy tokens — a controllable stand-in for a distilled draft model.
"""
import numpy as np

from . import gguf as G

# name -> (arch, n_layer, n_embd, n_head, n_head_kv, n_ff, n_vocab, n_ctx_train)
SPECS = {
    "tinyllama-1.1b": ("llama", 22, 2048, 32, 4, 5632, 32000, 2048),
    "llama-2-7b":     ("llama", 32, 4096, 32, 32, 11008, 32000, 4096),
    "llama-2-70b":    ("llama", 80, 8192, 64, 8, 28672, 32000, 4096),
    "goliath-120b":   ("llama", 137, 8192, 64, 8, 28672, 32000, 4096),
    "falcon-7b":      ("falcon", 32, 4544, 71, 1, 18176, 65024, 2048),
    "falcon-40b":     ("falcon", 60, 8192, 128, 8, 32768, 65024, 2048),
    # small shapes for tests
    "tiny-llama":     ("llama", 2, 256, 4, 2, 512, 512, 512),
    "tiny-llama-b":   ("llama", 3, 256, 4, 4, 768, 512, 512),
    "tiny-llama-hd80": ("llama", 2, 320, 4, 2, 512, 512, 512),             # head width 80 (StableLM-3B's; neither 64 nor 128): the plain attention kernel; n_embd 320 -> 32-blocks only
    "tiny-falcon":    ("falcon", 2, 256, 4, 1, 1024, 512, 512),
    "tiny-falcon40":  ("falcon", 2, 256, 4, 2, 1024, 512, 512),   # two-norm (40B-style) variant
    # the other graph builders of the reference (llama.cpp:3985-4939); depths 32 / 40 are what build_baichuan tells its 7B (RoPE) from its 13B (ALiBi) by
    "tiny-baichuan7":  ("baichuan", 32, 128, 2, 2, 256, 256, 512),
    "tiny-baichuan13": ("baichuan", 40, 128, 2, 2, 256, 256, 512),
    "tiny-starcoder":  ("starcoder", 2, 256, 4, 1, 1024, 512, 512),    # MQA, learned positions, biases everywhere
    "tiny-refact":     ("refact", 2, 384, 6, 1, 512, 512, 512),        # 6 heads: ALiBi slopes of both series (4 + 2); MQA
    "tiny-bloom":      ("bloom", 2, 384, 6, 6, 768, 512, 512),
    "tiny-mpt":        ("mpt", 2, 256, 4, 4, 1024, 512, 512),
    "tiny-stablelm":   ("stablelm", 2, 256, 4, 4, 512, 512, 512),      # rotates n_rot = 16 of the 64 values of a head
    "tiny-persimmon":  ("persimmon", 2, 256, 4, 4, 512, 512, 512),     # head width 64 as the reference requires; per-head Q / K LayerNorm, relu^2
}
# extra hyper-parameters of the synthetic files (name -> value): MPT's ALiBi bound and QKV clamp, StableLM's partial rotary width
EXTRA_KV = {
    "tiny-mpt": {"attention.max_alibi_bias": 8.0, "attention.clamp_kqv": 1.5},
    "tiny-stablelm": {"rope.dimension_count": 16},
}
RMS_ARCHS = ("llama", "baichuan", "refact")
NORM_BIAS_ARCHS = ("falcon", "starcoder", "bloom", "stablelm", "persimmon")
PROJ_BIAS_ARCHS = ("starcoder", "bloom", "persimmon")
FUSED_QKV_ARCHS = ("falcon", "starcoder", "bloom", "mpt", "persimmon")
GATED_ARCHS = ("llama", "baichuan", "refact", "stablelm")

FTYPE_IDS = {"f32": 0, "f16": 1, "q4_0": 2, "q8_0": 7, "q4_k_s": 14, "q4_k_m": 15, "q5_k_s": 16, "q5_k_m": 17, "q6_k": 18}


def use_more_bits(i, n):
    return i < n // 8 or i >= 7 * n // 8 or (i - n // 8) % 3 == 2


def tensor_plan(spec_name, ftype, falcon_two_norms=None, layers=None):
    """-> list of (name, ne(ggml order), ttype, role). Roles drive the synthetic fill.
    layers = (offset, count): a depth-truncated copy — `count` layers named blk.0.. that carry the tensor types of layers offset.. of the full model
    (the quantiser's mixture rules depend on the layer index); used for the per-layer CPU baseline of models too slow to decode whole."""
    arch, n_layer, n_embd, n_head, n_head_kv, n_ff, n_vocab, _ = SPECS[spec_name]
    hd = n_embd // n_head
    n_gqa = hd * n_head_kv
    ftype = ftype.lower()
    base = {"q4_0": G.T_Q4_0, "q4_1": G.T_Q4_1, "q5_0": G.T_Q5_0, "q5_1": G.T_Q5_1, "q8_0": G.T_Q8_0, "q2_k": G.T_Q2_K, "q3_k_m": G.T_Q3_K, "q4_k_m": G.T_Q4_K, "q4_k_s": G.T_Q4_K, "q5_k_m": G.T_Q5_K, "q5_k_s": G.T_Q5_K,
            "q6_k": G.T_Q6_K, "f16": G.T_F16, "f32": G.T_F32}[ftype]
    kq = ftype.startswith(("q2_k", "q3_k", "q4_k", "q5_k", "q6_k"))
    is_70b = arch == "llama" and n_layer == 80 and n_head != n_head_kv    # llama.cpp MODEL_70B detection (n_layer 80)

    def fix(t, nx):
        # K-quants need ne0 % 256 == 0; fallbacks per llama.cpp:7985-8003 that stay inside our kernel set
        if t in (G.T_Q2_K, G.T_Q3_K, G.T_Q4_K, G.T_Q5_K, G.T_Q6_K) and nx % 256 != 0:
            return G.T_Q8_0 if t == G.T_Q6_K else G.T_Q4_0
        return t

    def out_type():
        if base in (G.T_F16, G.T_F32):
            return base
        if arch == "falcon" or n_embd % 256 != 0:
            return G.T_Q8_0
        return base if base == G.T_Q8_0 else G.T_Q6_K

    def v_type(i):
        t = base
        if ftype == "q2_k":                                             # llama.cpp:7918
            t = G.T_Q3_K
        elif ftype == "q3_k_m":                                         # llama.cpp:7919-7921
            t = G.T_Q5_K if i < 2 else G.T_Q4_K
        elif ftype in ("q4_k_m", "q5_k_m") and use_more_bits(i, n_layer):
            t = G.T_Q6_K
        elif ftype == "q4_k_s" and i < 4:
            t = G.T_Q5_K
        if is_70b and t == G.T_Q4_K:
            t = G.T_Q5_K
        return t

    def down_type(i):
        t = base
        if ftype == "q2_k":                                             # llama.cpp:7936
            t = G.T_Q3_K
        elif ftype == "q3_k_m":                                         # llama.cpp:7937-7941
            t = G.T_Q5_K if i < 2 else (G.T_Q4_K if arch != "falcon" or use_more_bits(i, n_layer) else G.T_Q3_K)
        elif ftype == "q4_k_m":
            if arch == "falcon":
                t = G.T_Q6_K if i < 2 else (G.T_Q5_K if use_more_bits(i, n_layer) else G.T_Q4_K)
            elif use_more_bits(i, n_layer):
                t = G.T_Q6_K
        elif ftype == "q5_k_m" and use_more_bits(i, n_layer):
            t = G.T_Q6_K
        elif ftype == "q4_k_s" and arch != "falcon" and i < 4:
            t = G.T_Q5_K
        return t

    plan = [("token_embd.weight", (n_embd, n_vocab), fix(base, n_embd), "embd")]
    if arch == "starcoder":
        plan.append(("position_embd.weight", (n_embd, SPECS[spec_name][7]), fix(base, n_embd), "pos"))
    if arch == "bloom":
        plan += [("token_embd_norm.weight", (n_embd,), G.T_F32, "gain"), ("token_embd_norm.bias", (n_embd,), G.T_F32, "bias")]
    two_norms = falcon_two_norms if falcon_two_norms is not None else (arch == "falcon" and n_head_kv > 1)
    l_off, l_cnt = layers if layers else (0, n_layer)
    for li in range(l_cnt):
        p = f"blk.{li}."
        i = l_off + li                                                    # index the type rules see
        if arch == "llama":
            plan += [
                (p + "attn_norm.weight", (n_embd,), G.T_F32, "ones"),
                (p + "attn_q.weight", (n_embd, n_embd), fix(base, n_embd), "proj"),
                (p + "attn_k.weight", (n_embd, n_gqa), fix(base, n_embd), "proj"),
                (p + "attn_v.weight", (n_embd, n_gqa), fix(v_type(i) if kq else base, n_embd), "proj"),
                (p + "attn_output.weight", (n_embd, n_embd), fix(G.T_Q4_K if ftype == "q3_k_m" else (G.T_Q3_K if ftype == "q2_k" else base), n_embd), "out"),     # llama.cpp:7958-7961
                (p + "ffn_norm.weight", (n_embd,), G.T_F32, "ones"),
                (p + "ffn_gate.weight", (n_embd, n_ff), fix(G.T_Q3_K if ftype == "q2_k" else base, n_embd), "proj"),       # llama.cpp:7976-7978
                (p + "ffn_down.weight", (n_ff, n_embd), fix(down_type(i) if kq else base, n_ff), "out"),
                (p + "ffn_up.weight", (n_embd, n_ff), fix(G.T_Q3_K if ftype == "q2_k" else base, n_embd), "proj"),
            ]
        elif arch != "falcon":
            # tensor sets of llama.cpp:2825-2890 (baichuan; refact: llama's), 2969-3047 (starcoder), 3116-3197 (bloom), 3198-3264 (mpt), 3265-3345 (stablelm); norm gains and all
            # biases are random here (the reference's own files carry trained ones): a bias dropped or added in the wrong place must show
            nb_, pb_ = arch in NORM_BIAS_ARCHS, arch in PROJ_BIAS_ARCHS
            qkv_t = G.T_Q5_K if ftype == "q4_k_m" else (G.T_Q6_K if ftype == "q5_k_m" else (G.T_Q4_K if ftype == "q3_k_m" else base))
            plan.append((p + "attn_norm.weight", (n_embd,), G.T_F32, "gain"))
            if nb_:
                plan.append((p + "attn_norm.bias", (n_embd,), G.T_F32, "bias"))
            if arch in FUSED_QKV_ARCHS:
                plan.append((p + "attn_qkv.weight", (n_embd, n_embd + 2 * n_gqa), fix(qkv_t, n_embd), "proj"))
                if pb_:
                    plan.append((p + "attn_qkv.bias", (n_embd + 2 * n_gqa,), G.T_F32, "bias"))
            else:
                plan += [(p + "attn_q.weight", (n_embd, n_embd), fix(base, n_embd), "proj"), (p + "attn_k.weight", (n_embd, n_gqa), fix(base, n_embd), "proj"),
                         (p + "attn_v.weight", (n_embd, n_gqa), fix(v_type(i) if kq else base, n_embd), "proj")]
            if arch == "persimmon":                                         # llama.cpp:3105-3108: vectors of 64
                plan += [(p + "attn_q_norm.weight", (64,), G.T_F32, "gain"), (p + "attn_q_norm.bias", (64,), G.T_F32, "bias"),
                         (p + "attn_k_norm.weight", (64,), G.T_F32, "gain"), (p + "attn_k_norm.bias", (64,), G.T_F32, "bias")]
            plan.append((p + "attn_output.weight", (n_embd, n_embd), fix(base, n_embd), "out"))
            if pb_:
                plan.append((p + "attn_output.bias", (n_embd,), G.T_F32, "bias"))
            plan.append((p + "ffn_norm.weight", (n_embd,), G.T_F32, "gain"))
            if nb_:
                plan.append((p + "ffn_norm.bias", (n_embd,), G.T_F32, "bias"))
            if arch in GATED_ARCHS:
                plan.append((p + "ffn_gate.weight", (n_embd, n_ff), fix(base, n_embd), "proj"))
            plan.append((p + "ffn_down.weight", (n_ff, n_embd), fix(down_type(i) if kq else base, n_ff), "out"))
            if pb_:
                plan.append((p + "ffn_down.bias", (n_embd,), G.T_F32, "bias"))
            plan.append((p + "ffn_up.weight", (n_embd, n_ff), fix(base, n_embd), "proj"))
            if pb_:
                plan.append((p + "ffn_up.bias", (n_ff,), G.T_F32, "bias"))
        else:
            plan += [(p + "attn_norm.weight", (n_embd,), G.T_F32, "ones"), (p + "attn_norm.bias", (n_embd,), G.T_F32, "zeros")]
            if two_norms:
                plan += [(p + "attn_norm_2.weight", (n_embd,), G.T_F32, "ones"), (p + "attn_norm_2.bias", (n_embd,), G.T_F32, "zeros")]
            qkv_t = G.T_Q5_K if ftype == "q4_k_m" else (G.T_Q6_K if ftype == "q5_k_m" else (G.T_Q4_K if ftype == "q3_k_m" else base))
            plan += [
                (p + "attn_qkv.weight", (n_embd, n_embd + 2 * n_gqa), fix(qkv_t, n_embd), "proj"),
                (p + "attn_output.weight", (n_embd, n_embd), fix(base, n_embd), "out"),
                (p + "ffn_up.weight", (n_embd, n_ff), fix(base, n_embd), "proj"),
                (p + "ffn_down.weight", (n_ff, n_embd), fix(down_type(i) if kq else base, n_ff), "out"),
            ]
    plan.append(("output_norm.weight", (n_embd,), G.T_F32, "ones" if arch in ("llama", "falcon") else "gain"))
    if arch == "falcon":
        plan.append(("output_norm.bias", (n_embd,), G.T_F32, "zeros"))
    elif arch in NORM_BIAS_ARCHS:
        plan.append(("output_norm.bias", (n_embd,), G.T_F32, "bias"))
    plan.append(("output.weight", (n_embd, n_vocab), out_type(), "lm_head"))
    return plan


# ---- direct random block fill ------------------------------------------------------------------------------

def _rand_u8(rng, n):
    return np.frombuffer(rng.bytes(int(n)), dtype=np.uint8)


def _f16(x):
    return np.asarray(x, np.float32).astype(np.float16).view(np.uint16)


def _pack_k4_scales(sc, m):
    """inverse of get_scale_min_k4 (ggml-quants.c:1446-1453): sc, m uint8 [nb, 8] (6-bit) -> [nb, 12]"""
    q = np.zeros((sc.shape[0], 12), np.uint8)
    q[:, 0:4] = sc[:, 0:4] | ((sc[:, 4:8] >> 4) << 6)
    q[:, 4:8] = m[:, 0:4] | ((m[:, 4:8] >> 4) << 6)
    q[:, 8:12] = (sc[:, 4:8] & 0xF) | ((m[:, 4:8] & 0xF) << 4)
    return q


def random_blocks(ttype, n_elem, rng, std):
    """uint8 [n_blocks * type_size] whose dequantised values are ~zero-mean with the given std"""
    nb = n_elem // G.BLCK[ttype]
    if ttype == G.T_F32:
        return (rng.standard_normal(n_elem, dtype=np.float32) * np.float32(std)).view(np.uint8)
    if ttype == G.T_F16:
        return (rng.standard_normal(n_elem, dtype=np.float32) * np.float32(std)).astype(np.float16).view(np.uint8)
    out = _rand_u8(rng, nb * G.TSIZE[ttype]).reshape(nb, G.TSIZE[ttype]).copy()
    jitter = (0.75 + 0.5 * rng.random(nb)).astype(np.float32)
    if ttype == G.T_Q4_0:
        out[:, 0:2] = _f16(std / 4.61 * jitter).view(np.uint8).reshape(nb, 2)
    elif ttype == G.T_Q5_0:                                                  # d | qh 4 | qs 16: (5-bit - 16) * d, uniform 0..31 has std 9.23
        out[:, 0:2] = _f16(std / 9.23 * jitter).view(np.uint8).reshape(nb, 2)
    elif ttype in (G.T_Q4_1, G.T_Q5_1):                                      # d | m | (qh) | qs: q * d + m with m = -mid * d
        mid, spread = (7.5, 4.61) if ttype == G.T_Q4_1 else (15.5, 9.23)
        d = std / spread * jitter
        out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
        out[:, 2:4] = _f16(-mid * d).view(np.uint8).reshape(nb, 2)
    elif ttype == G.T_Q8_0:
        out[:, 0:2] = _f16(std / 73.3 * jitter).view(np.uint8).reshape(nb, 2)
        q = out[:, 2:].view(np.int8)
        q[q == -128] = -127
    elif ttype in (G.T_Q4_K, G.T_Q5_K):
        mid, spread = (7.5, 4.61) if ttype == G.T_Q4_K else (15.5, 9.23)
        ratio = 8.0 if ttype == G.T_Q4_K else 16.0
        sc = (20 + out[:, 4:12] % 25).astype(np.uint8)                    # 20..44
        m = np.rint(mid * sc / ratio).astype(np.uint8)
        d = std / (32.0 * spread) * jitter
        out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
        out[:, 2:4] = _f16(d * ratio).view(np.uint8).reshape(nb, 2)
        out[:, 4:16] = _pack_k4_scales(sc, m)
    elif ttype == G.T_Q2_K:
        # scales 16 (4-bit scale | 4-bit min << 4) | qs 64 | d | dmin: x = d*sc*q - dmin*m with q in 0..3; mins centred on 1.5 scales
        sc = 4 + out[:, 0:16] % 12                                                              # 4..15
        m = np.clip(np.rint(1.5 * sc), 0, 15).astype(np.uint8)
        out[:, 0:16] = sc | (m << 4)
        d = std / (9.5 * 1.118) * jitter                                                         # q - 1.5 has std 1.118
        out[:, 80:82] = _f16(d).view(np.uint8).reshape(nb, 2)
        out[:, 82:84] = _f16(d).view(np.uint8).reshape(nb, 2)
    elif ttype == G.T_Q3_K:
        # hmask 32 | qs 64 | scales 12 | d 2: random 3-bit weights (-4..3, std 2.29); the sixteen stored scales become 32 +- (4..19)
        mag = 4 + out[:, 96:108].astype(np.int32) % 16                                          # reuse the random bytes: 12 per block
        mag = np.concatenate([mag, 4 + (out[:, 32:36].astype(np.int32) % 16)], axis=1)          # 16 magnitudes
        sign = np.where(out[:, 40:56] & 1, 1, -1)
        sc6 = (32 + sign * mag).astype(np.uint8)                                                # 6-bit values 13..51
        lo, hi = sc6 & 0xF, sc6 >> 4
        out[:, 96:104] = lo[:, 0:8] | (lo[:, 8:16] << 4)
        out[:, 104:108] = hi[:, 0:4] | (hi[:, 4:8] << 2) | (hi[:, 8:12] << 4) | (hi[:, 12:16] << 6)
        out[:, 108:110] = _f16(std / (11.5 * 2.29) * jitter).view(np.uint8).reshape(nb, 2)
    elif ttype == G.T_Q6_K:
        sc = (20 + out[:, 192:208] % 25).astype(np.int8)
        sc[out[:, 192:208] >= 128] *= -1
        out[:, 192:208] = sc.view(np.uint8)
        out[:, 208:210] = _f16(std / (32.0 * 18.47) * jitter).view(np.uint8).reshape(nb, 2)
    else:
        raise ValueError(ttype)
    return out.reshape(-1)


# ---- simple (non-reference) quantisers for tensors whose VALUES matter (bigram structure) --------------------

def quantize_simple(ttype, x):
    """x: float32 [rows, k] -> uint8 blocks. Plain min/max rounding, not the reference's search — only used for synthetic data."""
    x = np.ascontiguousarray(x, np.float32)
    rows, k = x.shape
    if ttype == G.T_F32:
        return x.view(np.uint8).reshape(-1)
    if ttype == G.T_F16:
        return x.astype(np.float16).view(np.uint8).reshape(-1)
    if ttype == G.T_Q8_0:
        b = x.reshape(-1, 32)
        d = np.abs(b).max(1) / 127.0
        q = np.rint(b / np.where(d == 0, 1, d)[:, None]).astype(np.int8)
        out = np.zeros((b.shape[0], 34), np.uint8)
        out[:, 0:2] = _f16(d).view(np.uint8).reshape(-1, 2)
        out[:, 2:] = q.view(np.uint8)
        return out.reshape(-1)
    if ttype == G.T_Q4_0:
        b = x.reshape(-1, 32)
        d = np.abs(b).max(1) / 7.5
        q = np.clip(np.rint(b / np.where(d == 0, 1, d)[:, None] + 8.0), 0, 15).astype(np.uint8)
        out = np.zeros((b.shape[0], 18), np.uint8)
        out[:, 0:2] = _f16(d).view(np.uint8).reshape(-1, 2)
        out[:, 2:] = q[:, :16] | (q[:, 16:] << 4)
        return out.reshape(-1)
    if ttype == G.T_Q2_K:
        b = x.reshape(-1, 16, 16)                                          # [nb, 16 groups, 16]
        lo, hi = b.min(2), b.max(2)
        lo = np.minimum(lo, 0.0)                                           # x = scale*q - min with min >= 0
        gs = (hi - lo) / 3.0
        d = np.where(gs.max(1) == 0, 1e-8, gs.max(1) / 15.0)
        dm = np.where((-lo).max(1) == 0, 1e-8, (-lo).max(1) / 15.0)
        sc = np.clip(np.rint(gs / d[:, None]), 1, 15).astype(np.int32)
        m = np.clip(np.rint(-lo / dm[:, None]), 0, 15).astype(np.int32)
        q = np.clip(np.rint((b + (dm[:, None] * m)[:, :, None]) / (d[:, None] * sc)[:, :, None]), 0, 3).astype(np.uint8).reshape(-1, 2, 4, 32)
        nb = q.shape[0]
        out = np.zeros((nb, 84), np.uint8)
        out[:, 0:16] = (sc | (m << 4)).astype(np.uint8)
        out[:, 16:80] = (q[:, :, 0] | (q[:, :, 1] << 2) | (q[:, :, 2] << 4) | (q[:, :, 3] << 6)).reshape(nb, 64)
        out[:, 80:82] = _f16(d).view(np.uint8).reshape(nb, 2)
        out[:, 82:84] = _f16(dm).view(np.uint8).reshape(nb, 2)
        return out.reshape(-1)
    if ttype == G.T_Q3_K:
        b = x.reshape(-1, 16, 16)                                          # [nb, 16 groups, 16]
        gs = np.abs(b).max(2) / 4.0
        d = gs.max(1) / 31.0
        d = np.where(d == 0, 1e-8, d)
        sc = np.clip(np.rint(gs / d[:, None]), 1, 31).astype(np.int32)
        q = (np.clip(np.rint(b / (d[:, None, None] * sc[:, :, None])), -4, 3).astype(np.int32) + 4).reshape(-1, 2, 4, 32)   # [nb, half n, j, l] 0..7
        nb = q.shape[0]
        out = np.zeros((nb, 110), np.uint8)
        lo, hi = (q & 3).astype(np.uint8), (q >> 2).astype(np.uint8)
        out[:, 32:96] = (lo[:, :, 0] | (lo[:, :, 1] << 2) | (lo[:, :, 2] << 4) | (lo[:, :, 3] << 6)).reshape(nb, 64)      # qs[32n + l]
        hm = np.zeros((nb, 32), np.uint8)
        for n in range(2):
            for j in range(4):
                hm |= hi[:, n, j] << (4 * n + j)
        out[:, 0:32] = hm
        sc6 = (sc + 32).astype(np.uint8)
        l4, h2 = sc6 & 0xF, sc6 >> 4
        out[:, 96:104] = l4[:, 0:8] | (l4[:, 8:16] << 4)
        out[:, 104:108] = h2[:, 0:4] | (h2[:, 4:8] << 2) | (h2[:, 8:12] << 4) | (h2[:, 12:16] << 6)
        out[:, 108:110] = _f16(d).view(np.uint8).reshape(nb, 2)
        return out.reshape(-1)
    if ttype == G.T_Q6_K:
        b = x.reshape(-1, 16, 16)                                          # [nb, 16 groups, 16]
        gs = np.abs(b).max(2) / 31.0                                       # group scale
        d = gs.max(1) / 127.0
        d = np.where(d == 0, 1e-8, d)
        sc = np.clip(np.rint(gs / d[:, None]), 1, 127).astype(np.int8)
        q = np.clip(np.rint(b / (d[:, None, None] * sc[:, :, None])) + 32, 0, 63).astype(np.uint8).reshape(-1, 256)
        nb = q.shape[0]
        out = np.zeros((nb, 210), np.uint8)
        qq = q.reshape(nb, 2, 4, 32)                                       # [half, group g, l]
        lo = qq & 0xF
        out[:, 0:128] = np.stack([lo[:, :, 0] | (lo[:, :, 2] << 4), lo[:, :, 1] | (lo[:, :, 3] << 4)], 2).reshape(nb, 128)
        hi = qq >> 4
        out[:, 128:192] = (hi[:, :, 0] | (hi[:, :, 1] << 2) | (hi[:, :, 2] << 4) | (hi[:, :, 3] << 6)).reshape(nb, 64)
        out[:, 192:208] = sc.view(np.uint8)
        out[:, 208:210] = _f16(d).view(np.uint8).reshape(nb, 2)
        return out.reshape(-1)
    if ttype in (G.T_Q4_K, G.T_Q5_K):
        qmax = 15 if ttype == G.T_Q4_K else 31
        b = x.reshape(-1, 8, 32)
        mn = np.minimum(b.min(2), 0.0)
        scale = (b.max(2) - mn) / qmax
        d = scale.max(1) / 63.0
        dmin = (-mn).max(1) / 63.0
        d = np.where(d == 0, 1e-8, d)
        dmin = np.where(dmin == 0, 1e-8, dmin)
        d = _f16(d).view(np.float16).astype(np.float32); dmin = _f16(dmin).view(np.float16).astype(np.float32)
        d = np.where(d == 0, 1e-7, d); dmin = np.where(dmin == 0, 1e-7, dmin)
        sc = np.clip(np.rint(scale / d[:, None]), 0, 63).astype(np.uint8)
        m = np.clip(np.rint(-mn / dmin[:, None]), 0, 63).astype(np.uint8)
        eff = d[:, None] * sc
        q = np.clip(np.rint((b + (dmin[:, None] * m)[:, :, None]) / np.where(eff == 0, 1, eff)[:, :, None]), 0, qmax).astype(np.uint8)
        nb = q.shape[0]
        out = np.zeros((nb, G.TSIZE[ttype]), np.uint8)
        out[:, 0:2] = _f16(d).view(np.uint8).reshape(nb, 2)
        out[:, 2:4] = _f16(dmin).view(np.uint8).reshape(nb, 2)
        out[:, 4:16] = _pack_k4_scales(sc, m)
        qp = q.reshape(nb, 4, 2, 32)
        packed = ((qp[:, :, 0] & 0xF) | ((qp[:, :, 1] & 0xF) << 4)).reshape(nb, 128)
        if ttype == G.T_Q4_K:
            out[:, 16:144] = packed
        else:
            hb = (qp >> 4).astype(np.uint8)                                # [nb, 4, 2, 32]
            qh = np.zeros((nb, 32), np.uint8)
            for j in range(4):
                for h in range(2):
                    qh |= hb[:, j, h] << (2 * j + h)
            out[:, 16:48] = qh
            out[:, 48:176] = packed
        return out.reshape(-1)
    raise ValueError(ttype)


def _vocab_kv(w, arch, n_vocab, vocab=None):
    """tokenizer section. `vocab` = (tokens, scores, types[, merges]) from a real vocab file when available; otherwise a
    synthetic byte-fallback SPM vocab (ids 0..2 specials, 3..258 <0xXX>, rest unique dummy pieces)."""
    if vocab is None:
        tokens = [b"<unk>", b"<s>", b"</s>"] + [b"<0x%02X>" % i for i in range(256)]
        types = [2, 3, 3] + [6] * 256
        while len(tokens) < n_vocab:
            tokens.append(("▁t%d" % len(tokens)).encode())
            types.append(1)
        tokens, types = tokens[:n_vocab], types[:n_vocab]
        scores = [0.0] * 259 + [-float(i) for i in range(n_vocab - 259)] if n_vocab > 259 else [0.0] * n_vocab
        vocab = (tokens, scores[:n_vocab], types)
    w.add("tokenizer.ggml.model", G.STR, "llama" if arch == "llama" or len(vocab) < 4 else "gpt2")
    w.add("tokenizer.ggml.tokens", G.ARR, (G.STR, vocab[0]))
    w.add("tokenizer.ggml.scores", G.ARR, (G.F32, vocab[1]))
    w.add("tokenizer.ggml.token_type", G.ARR, (G.I32, vocab[2]))
    if len(vocab) > 3:
        w.add("tokenizer.ggml.merges", G.ARR, (G.STR, vocab[3]))
    w.add("tokenizer.ggml.bos_token_id", G.U32, 1 if arch == "llama" else 11)
    w.add("tokenizer.ggml.eos_token_id", G.U32, 2 if arch == "llama" else 11)
    if arch == "llama":
        w.add("tokenizer.ggml.unknown_token_id", G.U32, 0)


def write_model(path, spec_name, ftype, seed=0x5EED, std=0.02, structure="random", noise=1.0, quantizer=None, vocab=None,
                falcon_two_norms=None, pi_seed=1234, signal=1.0, head_gain=1.0, only=None, pi_flip=0.0, layers=None):
    """quantizer(ttype, f32 [rows,k]) -> uint8 blocks: when given (tests: the REAL reference quantiser), every weight is
    drawn as N(0,std^2) floats and quantised with it; otherwise blocks are filled directly (see module docstring).
    only(name) -> bool: tensors for which it is False become holes in the file (every tensor has its own seed, so a pipeline
    stage that writes just its layers gets the same bytes as a full file)."""
    arch, n_layer, n_embd, n_head, n_head_kv, n_ff, n_vocab, n_ctx_train = SPECS[spec_name]
    w = G.GGUFWriter(path)
    w.add("general.architecture", G.STR, arch)
    w.add("general.name", G.STR, f"synthetic-{spec_name}-{ftype}")
    w.add("general.file_type", G.U32, FTYPE_IDS.get(ftype.lower(), 0))
    w.add(f"{arch}.context_length", G.U32, n_ctx_train)
    w.add(f"{arch}.embedding_length", G.U32, n_embd)
    w.add(f"{arch}.block_count", G.U32, layers[1] if layers else n_layer)
    w.add(f"{arch}.feed_forward_length", G.U32, n_ff)
    w.add(f"{arch}.attention.head_count", G.U32, n_head)
    w.add(f"{arch}.attention.head_count_kv", G.U32, n_head_kv)
    if arch == "llama":
        w.add("llama.rope.dimension_count", G.U32, n_embd // n_head)
        w.add("llama.attention.layer_norm_rms_epsilon", G.F32, 1e-5)
    elif arch in RMS_ARCHS:
        w.add(f"{arch}.attention.layer_norm_rms_epsilon", G.F32, 1e-6)
    else:
        w.add(f"{arch}.attention.layer_norm_epsilon", G.F32, 1e-5)
    for key, val in EXTRA_KV.get(spec_name, {}).items():
        w.add(f"{arch}.{key}", G.U32 if isinstance(val, int) else G.F32, val)
    _vocab_kv(w, arch, n_vocab, vocab)

    pi = np.random.default_rng(pi_seed).permutation(n_vocab) if structure == "bigram" else None
    if pi is not None and pi_flip > 0.0:
        # a DRAFT model that is wrong about a fixed fraction of the vocabulary: for those tokens its bigram points somewhere else.  The
        # target (pi_flip = 0) follows pi, so the draft's per-token acceptance is ~ 1 - pi_flip BY CONSTRUCTION, with large logit margins in
        # both models — unlike disagreement produced by layer noise, which only appears where the target's top two logits are nearly tied
        # (and a near tie can flip with the KV-cell layout, see DESIGN.md on losslessness)
        fr = np.random.default_rng(pi_seed ^ 0xF11B)
        wrong = fr.random(n_vocab) < pi_flip
        pi = pi.copy()
        pi[wrong] = (pi[wrong] + 1 + fr.integers(0, n_vocab - 1, int(wrong.sum()))) % n_vocab
    depth_scale = min(1.0, float(np.sqrt((11008.0 * 4096.0 * 32.0) / (float(n_ff) * n_embd * n_layer))))
    emb_cache = {}

    def embedding_matrix():
        if "e" not in emb_cache:
            r = np.random.default_rng(seed ^ 0xE3BD)
            emb_cache["e"] = r.standard_normal((n_vocab, n_embd), dtype=np.float32)
        return emb_cache["e"]

    for idx, (name, ne, ttype, role) in enumerate(tensor_plan(spec_name, ftype, falcon_two_norms, layers)):
        def make(name=name, ne=ne, ttype=ttype, role=role, idx=idx):
            rng = np.random.default_rng(seed + idx)
            n = int(np.prod(ne))
            if role == "ones":
                return np.ones(n, np.float32)
            if role == "zeros":
                return np.zeros(n, np.float32)
            if role == "gain":
                return (1.0 + 0.1 * rng.standard_normal(n)).astype(np.float32)
            if role == "bias":
                return (0.05 * rng.standard_normal(n)).astype(np.float32)
            rows, k = (ne[1], ne[0])
            if structure == "bigram" and role in ("embd", "lm_head"):
                e = embedding_matrix()
                if role == "embd":
                    mat = e * np.float32(signal)
                else:
                    if pi_flip > 0.0:                                        # not a permutation any more: rows with several predecessors add up, rows with none stay zero
                        mat = np.zeros_like(e)
                        np.add.at(mat, pi, e * np.float32(head_gain / np.sqrt(n_embd)))
                    else:
                        mat = np.empty_like(e)
                        mat[pi] = e * np.float32(head_gain / np.sqrt(n_embd))   # logit of pi[t] ~ head_gain*sqrt(n_embd)*cos(h, e_t); unrelated logits ~ N(0, head_gain^2)
                return (quantizer or quantize_simple)(ttype, mat) if quantizer else quantize_simple(ttype, mat)
            # "bigram": the residual-writing tensors (wo, ffn_down) carry the noise.  What a layer adds to the stream grows like
            # sqrt(n_ff * n_embd) and accumulates over n_layer, so the knob is normalised to the Llama-2-7B shape (never upwards:
            # small draft models stay quiet) — the same `noise` then means a similar next-token margin for 7B and 70B targets
            s = std * (noise * depth_scale if (structure == "bigram" and role == "out") else 1.0)
            if quantizer is not None:
                return quantizer(ttype, rng.standard_normal((rows, k), dtype=np.float32) * np.float32(s))
            return random_blocks(ttype, n, rng, s)
        w.add_tensor(name, ne, ttype, make if (only is None or only(name)) else None)
    w.write()
    emb_cache.clear()
    return {"arch": arch, "n_layer": n_layer, "n_embd": n_embd, "n_head": n_head, "n_head_kv": n_head_kv, "n_ff": n_ff,
            "n_vocab": n_vocab, "pi": pi}


def model_nbytes(spec_name, ftype):
    return sum(G.nbytes(t, ne) for _, ne, t, _ in tensor_plan(spec_name, ftype))
