// Offline weight quantisation: fp32 rows -> the block formats the engine runs, and llama_model_quantize on top of it.
//
// Restates the reference's deterministic "reference" quantisers (ggml-quants.c: quantize_row_q4_0_reference :428, q4_1 :469, q5_0 :510,
// q5_1 :558, q8_0 :607, q2_K :1458, q3_K :1597, q4_K :1810, q5_K :1975, q6_K :2181, and the searches they share: nearest_int :1191,
// make_qx_quants :1198, make_q3_quants :1262, make_qkx2_quants :1364) and the file driver llama_model_quantize_internal with its type
// mixture rules (llama.cpp:7896-8006, 8008-8286).  The searches compare fp32 error sums, so the OUTPUT BYTES depend on the order of every
// fp32 operation: each sum below runs in the reference's element order and association, and this file is compiled without contraction
// (csrc/Makefile: -ffp-contract=off), like the reference's own ISO-C build.  tests/test_quantize.py compares the bytes with the real reference
// functions (built as test infrastructure) and with committed golden blocks.
//
// Host code: the reference quantises model files on the CPU too; nothing here is on the per-token path.
#include "engine.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <thread>

namespace {

// ---- scalar helpers ----------------------------------------------------------------------------------------------------------------
inline float half_to_float(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000u) << 16, em = h & 0x7fffu;
    uint32_t bits;
    if (em >= 0x7c00u) bits = sign | 0x7f800000u | ((em & 0x3ffu) << 13);                 // inf / nan
    else if (em >= 0x0400u) bits = sign | ((em + (112u << 10)) << 13);                      // normal
    else if (em == 0) bits = sign;
    else {                                                                                  // subnormal: value = em * 2^-24
        float f = (float) em * 5.9604644775390625e-8f;
        memcpy(&bits, &f, 4); bits |= sign;
    }
    float out; memcpy(&out, &bits, 4); return out;
}
// round to nearest even, like _cvtss_sh(x, 0) (GGML_FP32_TO_FP16 on x86, ggml-impl.h)
inline uint16_t float_to_half(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x >= 0x7f800000u) return (uint16_t) (sign | (x > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (x >= 0x477ff000u) return (uint16_t) (sign | 0x7c00u);                               // rounds to >= 65520 -> inf
    if (x < 0x38800000u) {                                                                  // below the smallest normal half: subnormal or zero
        if (x < 0x33000000u) return sign;                                                   // < 2^-25 -> 0 (2^-25 itself ties to even = 0)
        const int e = (int) (x >> 23);                                                      // biased exponent, 102..112
        const uint32_t m = (x & 0x7fffffu) | 0x800000u;
        const int shift = 126 - e;                                                          // 14..24
        uint32_t q = m >> shift;
        const uint32_t rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
        if (rem > halfway || (rem == halfway && (q & 1u))) q++;
        return (uint16_t) (sign | q);
    }
    uint32_t q = (x - 0x38000000u) >> 13;
    const uint32_t rem = x & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (q & 1u))) q++;
    return (uint16_t) (sign | q);
}
// the reference's magic-number rounding (ggml-quants.c:1191-1196): round half to even of a float below 2^22
inline int round_magic(float v) {
    const float shifted = v + 12582912.f;
    int bits; memcpy(&bits, &shifted, 4);
    return (bits & 0x007fffff) - 0x00400000;
}
inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// ---- searches ----------------------------------------------------------------------------------------------------------------------
// Weighted affine fit x ~ scale * L + min over L in [0, nmax] (make_qkx2_quants): a first guess from the range, then nstep + 1 trial
// scales; a trial replaces the best one when its weighted error (absolute or squared) is smaller.  Returns scale; neg_min = -min.
float fit_affine(int n, int nmax, const float * x, const float * w, uint8_t * L, float * neg_min, uint8_t * trial, float rmin, float rdelta, int nstep, bool abs_err) {
    float lo = x[0], hi = x[0], sum_w = w[0], sum_wx = sum_w * x[0];
    for (int i = 1; i < n; i++) {
        if (x[i] < lo) lo = x[i];
        if (x[i] > hi) hi = x[i];
        sum_w += w[i];
        sum_wx += w[i] * x[i];
    }
    if (lo > 0) lo = 0;
    if (hi == lo) {
        memset(L, 0, (size_t) n);
        *neg_min = -lo;
        return 0.f;
    }
    float inv = nmax / (hi - lo), scale = 1 / inv, best = 0;
    for (int i = 0; i < n; i++) {
        L[i] = (uint8_t) clampi(round_magic(inv * (x[i] - lo)), 0, nmax);
        float e = scale * L[i] + lo - x[i];
        e = abs_err ? fabsf(e) : e * e;
        best += w[i] * e;
    }
    if (nstep < 1) { *neg_min = -lo; return scale; }
    for (int step = 0; step <= nstep; step++) {
        inv = (rmin + rdelta * step + nmax) / (hi - lo);
        float s_l = 0, s_l2 = 0, s_xl = 0;
        for (int i = 0; i < n; i++) {
            const int l = clampi(round_magic(inv * (x[i] - lo)), 0, nmax);
            trial[i] = (uint8_t) l;
            const float wi = w[i];
            s_l += wi * l;
            s_l2 += wi * l * l;
            s_xl += wi * l * x[i];
        }
        const float det = sum_w * s_l2 - s_l * s_l;
        if (det > 0) {
            float t_scale = (sum_w * s_xl - sum_wx * s_l) / det;
            float t_min = (s_l2 * sum_wx - s_l * s_xl) / det;
            if (t_min > 0) { t_min = 0; t_scale = s_xl / s_l2; }
            float err = 0;
            for (int i = 0; i < n; i++) {
                float e = t_scale * trial[i] + t_min - x[i];
                e = abs_err ? fabsf(e) : e * e;
                err += w[i] * e;
            }
            if (err < best) {
                memcpy(L, trial, (size_t) n);
                best = err; scale = t_scale; lo = t_min;
            }
        }
    }
    *neg_min = -lo;
    return scale;
}

// Symmetric fit x ~ scale * (L - nmax), weights x^2, 18 perturbed scales around -nmax / max (make_qx_quants with rmse_type 1: the Q6_K call)
float fit_symmetric(int n, int nmax, const float * x, int8_t * L) {
    float peak = 0, apeak = 0;
    for (int i = 0; i < n; i++) { const float a = fabsf(x[i]); if (a > apeak) { apeak = a; peak = x[i]; } }
    if (apeak < 1e-30f) { memset(L, 0, (size_t) n); return 0.f; }
    float inv = -nmax / peak, s_lx = 0, s_l2 = 0;
    for (int i = 0; i < n; i++) {
        const int l = clampi(round_magic(inv * x[i]), -nmax, nmax - 1);
        L[i] = (int8_t) (l + nmax);
        const float wi = x[i] * x[i];
        s_lx += wi * x[i] * l;
        s_l2 += wi * l * l;
    }
    float scale = s_lx / s_l2, best = scale * s_lx;
    for (int is = -9; is <= 9; is++) {
        if (is == 0) continue;
        inv = -(nmax + 0.1f * is) / peak;
        s_lx = s_l2 = 0;
        for (int i = 0; i < n; i++) {
            const int l = clampi(round_magic(inv * x[i]), -nmax, nmax - 1);
            const float wi = x[i] * x[i];
            s_lx += wi * x[i] * l;
            s_l2 += wi * l * l;
        }
        if (s_l2 > 0 && s_lx * s_lx > best * s_l2) {
            for (int i = 0; i < n; i++) L[i] = (int8_t) (nmax + clampi(round_magic(inv * x[i]), -nmax, nmax - 1));
            scale = s_lx / s_l2; best = scale * s_lx;
        }
    }
    return scale;
}

// Q3_K's 16-value search (make_q3_quants, do_rmse): start from -nmax / max, then up to five passes that move one value at a time when the
// weighted correlation improves.  L comes back offset by nmax.
float fit_q3(int n, int nmax, const float * x, int8_t * L) {
    float peak = 0, apeak = 0;
    for (int i = 0; i < n; i++) { const float a = fabsf(x[i]); if (a > apeak) { apeak = a; peak = x[i]; } }
    if (!apeak) { memset(L, 0, (size_t) n); return 0.f; }
    const float inv = -nmax / peak;
    float s_lx = 0, s_l2 = 0;
    for (int i = 0; i < n; i++) {
        const int l = clampi(round_magic(inv * x[i]), -nmax, nmax - 1);
        L[i] = (int8_t) l;
        const float wi = x[i] * x[i];
        s_lx += wi * x[i] * l;
        s_l2 += wi * l * l;
    }
    for (int pass = 0; pass < 5; pass++) {
        int moved = 0;
        for (int i = 0; i < n; i++) {
            const float wi = x[i] * x[i];
            float lx = s_lx - wi * x[i] * L[i];
            if (lx > 0) {
                float l2 = s_l2 - wi * L[i] * L[i];
                const int cand = clampi(round_magic(x[i] * l2 / lx), -nmax, nmax - 1);
                if (cand != L[i]) {
                    lx += wi * x[i] * cand;
                    l2 += wi * cand * cand;
                    if (l2 > 0 && lx * lx * s_l2 > s_lx * s_lx * l2) { L[i] = (int8_t) cand; s_lx = lx; s_l2 = l2; moved++; }
                }
            }
        }
        if (!moved) break;
    }
    for (int i = 0; i < n; i++) L[i] = (int8_t) (L[i] + nmax);
    return s_lx / s_l2;
}

// ---- 32-value formats --------------------------------------------------------------------------------------------------------------
struct leg { int bytes; bool affine; int bits; };
// Q4_0 / Q5_0: d = max / -2^(bits-1) with max the value of largest magnitude; Q4_1 / Q5_1: d = (max - min) / (2^bits - 1), m = min
template <int BITS, bool AFFINE>
void row_legacy(const float * x, uint8_t * out, int64_t k) {
    constexpr int BYTES = 2 + (AFFINE ? 2 : 0) + (BITS == 5 ? 4 : 0) + 16, TOP = (1 << BITS) - 1;
    for (int64_t b = 0; b < k / 32; b++, x += 32, out += BYTES) {
        float d, lo = 0.f;
        if (AFFINE) {
            float mn = 3.402823466e+38f, mx = -3.402823466e+38f;
            for (int j = 0; j < 32; j++) { if (x[j] < mn) mn = x[j]; if (x[j] > mx) mx = x[j]; }
            d = (mx - mn) / TOP; lo = mn;
        } else {
            float apeak = 0.f, peak = 0.f;
            for (int j = 0; j < 32; j++) if (apeak < fabsf(x[j])) { apeak = fabsf(x[j]); peak = x[j]; }
            d = peak / -(1 << (BITS - 1));
        }
        const float id = d ? 1.0f / d : 0.0f;
        const uint16_t dh = float_to_half(d);
        memcpy(out, &dh, 2);
        uint8_t * p = out + 2;
        if (AFFINE) { const uint16_t mh = float_to_half(lo); memcpy(p, &mh, 2); p += 2; }
        uint8_t * qs = p + (BITS == 5 ? 4 : 0);
        uint32_t fifth = 0;
        const float bias = AFFINE ? 0.5f : (float) (1 << (BITS - 1)) + 0.5f;
        for (int j = 0; j < 16; j++) {
            const float a = AFFINE ? (x[j] - lo) * id : x[j] * id, c = AFFINE ? (x[16 + j] - lo) * id : x[16 + j] * id;
            uint8_t q0, q1;
            if (AFFINE && BITS == 5) { q0 = (uint8_t) (a + bias); q1 = (uint8_t) (c + bias); }       // (q5_1 has no upper clamp)
            else { q0 = (uint8_t) std::min(TOP, (int) (int8_t) (a + bias)); q1 = (uint8_t) std::min(TOP, (int) (int8_t) (c + bias)); }
            qs[j] = (uint8_t) ((q0 & 0x0F) | ((q1 & 0x0F) << 4));
            if (BITS == 5) { fifth |= (uint32_t) ((q0 & 0x10u) >> 4) << j; fifth |= (uint32_t) ((q1 & 0x10u) >> 4) << (j + 16); }
        }
        if (BITS == 5) memcpy(p, &fifth, 4);
    }
}
void row_q8_0(const float * x, uint8_t * out, int64_t k) {
    for (int64_t b = 0; b < k / 32; b++, x += 32, out += 34) {
        float amax = 0.0f;
        for (int j = 0; j < 32; j++) amax = std::max(amax, fabsf(x[j]));
        const float d = amax / 127, id = d ? 1.0f / d : 0.0f;
        const uint16_t dh = float_to_half(d);
        memcpy(out, &dh, 2);
        for (int j = 0; j < 32; j++) out[2 + j] = (uint8_t) (int8_t) roundf(x[j] * id);
    }
}

// ---- 256-value formats -------------------------------------------------------------------------------------------------------------
// 2-bit planes of four 32-value groups per byte, 128 values at a time (shared by Q2_K and Q3_K)
void pack_2bit(const uint8_t * L, uint8_t * qs) {
    for (int j = 0; j < 256; j += 128)
        for (int l = 0; l < 32; l++) qs[j / 4 + l] = (uint8_t) (L[j + l] | (L[j + l + 32] << 2) | (L[j + l + 64] << 4) | (L[j + l + 96] << 6));
}
void row_q2_K(const float * x, uint8_t * out, int64_t k) {                                    // { scales[16], qs[64], d, dmin }
    uint8_t L[256], trial[16];
    float w[16], mins[16], scales[16];
    for (int64_t b = 0; b < k / 256; b++, x += 256, out += 84) {
        uint8_t * sc = out, * qs = out + 16;
        float top_scale = 0, top_min = 0;
        for (int j = 0; j < 16; j++) {
            for (int l = 0; l < 16; l++) w[l] = fabsf(x[16 * j + l]);
            scales[j] = fit_affine(16, 3, x + 16 * j, w, L + 16 * j, &mins[j], trial, -0.5f, 0.1f, 15, true);
            if (scales[j] > top_scale) top_scale = scales[j];
            if (mins[j] > top_min) top_min = mins[j];
        }
        uint16_t dh, mh;
        if (top_scale > 0) {
            const float inv = 15.f / top_scale;
            for (int j = 0; j < 16; j++) sc[j] = (uint8_t) round_magic(inv * scales[j]);
            dh = float_to_half(top_scale / 15.f);
        } else { memset(sc, 0, 16); dh = float_to_half(0.f); }
        if (top_min > 0) {
            const float inv = 15.f / top_min;
            for (int j = 0; j < 16; j++) sc[j] = (uint8_t) (sc[j] | (round_magic(inv * mins[j]) << 4));
            mh = float_to_half(top_min / 15.f);
        } else mh = float_to_half(0.f);
        for (int j = 0; j < 16; j++) {                                                        // requantise with the stored (rounded) scale and min
            const float d = half_to_float(dh) * (sc[j] & 0xF);
            if (!d) continue;
            const float dm = half_to_float(mh) * (sc[j] >> 4);
            for (int i = 0; i < 16; i++) L[16 * j + i] = (uint8_t) clampi(round_magic((x[16 * j + i] + dm) / d), 0, 3);
        }
        pack_2bit(L, qs);
        memcpy(out + 80, &dh, 2); memcpy(out + 82, &mh, 2);
    }
}
void row_q3_K(const float * x, uint8_t * out, int64_t k) {                                    // { hmask[32], qs[64], scales[12], d }
    int8_t L[256];
    float scales[16];
    for (int64_t b = 0; b < k / 256; b++, x += 256, out += 110) {
        uint8_t * hmask = out, * qs = out + 32, * sc = out + 96;
        float top = 0, atop = 0;
        for (int j = 0; j < 16; j++) {
            scales[j] = fit_q3(16, 4, x + 16 * j, L + 16 * j);
            const float a = fabsf(scales[j]);
            if (a > atop) { atop = a; top = scales[j]; }
        }
        memset(sc, 0, 12);
        uint16_t dh;
        if (top) {
            const float inv = -32.f / top;
            for (int j = 0; j < 16; j++) {                                                    // 6-bit scales: low nibbles in bytes 0..7, top two bits in bytes 8..11
                int8_t l = (int8_t) round_magic(inv * scales[j]);
                l = (int8_t) (clampi(l, -32, 31) + 32);
                if (j < 8) sc[j] = (uint8_t) (l & 0xF); else sc[j - 8] = (uint8_t) (sc[j - 8] | ((l & 0xF) << 4));
                l = (int8_t) (l >> 4);
                sc[j % 4 + 8] = (uint8_t) (sc[j % 4 + 8] | (l << (2 * (j / 4))));
            }
            dh = float_to_half(1 / inv);
        } else dh = float_to_half(0.f);
        for (int j = 0; j < 16; j++) {
            int8_t s6 = (int8_t) (j < 8 ? sc[j] & 0xF : sc[j - 8] >> 4);
            s6 = (int8_t) ((s6 | (((sc[8 + j % 4] >> (2 * (j / 4))) & 3) << 4)) - 32);
            const float d = half_to_float(dh) * s6;
            if (!d) continue;
            for (int i = 0; i < 16; i++) L[16 * j + i] = (int8_t) (clampi(round_magic(x[16 * j + i] / d), -4, 3) + 4);
        }
        memset(hmask, 0, 32);
        for (int j = 0, m = 0, bit = 1; j < 256; j++) {                                       // third bit: value j -> byte j % 32, bit j / 32
            if (L[j] > 3) { hmask[m] = (uint8_t) (hmask[m] | bit); L[j] = (int8_t) (L[j] - 4); }
            if (++m == 32) { m = 0; bit <<= 1; }
        }
        pack_2bit((const uint8_t *) L, qs);
        memcpy(out + 108, &dh, 2);
    }
}
// 6-bit (scale, min) pairs of eight 32-value groups in 12 bytes, and back (get_scale_min_k4, ggml-quants.c:1446-1453)
void pack_scale_min(int j, uint8_t ls, uint8_t lm, uint8_t * q) {
    if (j < 4) { q[j] = ls; q[j + 4] = lm; }
    else {
        q[j + 4] = (uint8_t) ((ls & 0xF) | ((lm & 0xF) << 4));
        q[j - 4] = (uint8_t) (q[j - 4] | ((ls >> 4) << 6));
        q[j] = (uint8_t) (q[j] | ((lm >> 4) << 6));
    }
}
void unpack_scale_min(int j, const uint8_t * q, uint8_t & s, uint8_t & m) {
    if (j < 4) { s = q[j] & 63; m = q[j + 4] & 63; }
    else { s = (uint8_t) ((q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4)); m = (uint8_t) ((q[j + 4] >> 4) | ((q[j] >> 6) << 4)); }
}
template <int BITS>                                                                           // 4: Q4_K { d, dmin, scales[12], qs[128] }; 5: Q5_K { d, dmin, scales[12], qh[32], qs[128] }
void row_q45_K(const float * x, uint8_t * out, int64_t k) {
    constexpr int NMAX = (1 << BITS) - 1, BYTES = BITS == 4 ? 144 : 176;
    uint8_t L[256], trial[32];
    float w[32], mins[8], scales[8];
    for (int64_t b = 0; b < k / 256; b++, x += 256, out += BYTES) {
        uint8_t * sc = out + 4;
        float top_scale = 0, top_min = 0;
        for (int j = 0; j < 8; j++) {
            float sq = 0;
            for (int l = 0; l < 32; l++) sq += x[32 * j + l] * x[32 * j + l];
            const float rms = sqrtf(sq / 32);
            for (int l = 0; l < 32; l++) w[l] = rms + fabsf(x[32 * j + l]);
            scales[j] = BITS == 4 ? fit_affine(32, NMAX, x + 32 * j, w, L + 32 * j, &mins[j], trial, -1.f, 0.1f, 20, false)
                                  : fit_affine(32, NMAX, x + 32 * j, w, L + 32 * j, &mins[j], trial, -0.5f, 0.1f, 15, false);
            if (scales[j] > top_scale) top_scale = scales[j];
            if (mins[j] > top_min) top_min = mins[j];
        }
        const float inv_s = top_scale > 0 ? 63.f / top_scale : 0.f, inv_m = top_min > 0 ? 63.f / top_min : 0.f;
        for (int j = 0; j < 8; j++) {
            uint8_t ls = (uint8_t) round_magic(inv_s * scales[j]), lm = (uint8_t) round_magic(inv_m * mins[j]);
            ls = std::min<uint8_t>(63, ls); lm = std::min<uint8_t>(63, lm);
            pack_scale_min(j, ls, lm, sc);
        }
        const uint16_t dh = float_to_half(top_scale / 63.f), mh = float_to_half(top_min / 63.f);
        memcpy(out, &dh, 2); memcpy(out + 2, &mh, 2);
        for (int j = 0; j < 8; j++) {
            uint8_t s6, m6;
            unpack_scale_min(j, sc, s6, m6);
            const float d = half_to_float(dh) * s6;
            if (!d) continue;
            const float dm = half_to_float(mh) * m6;
            for (int i = 0; i < 32; i++) L[32 * j + i] = (uint8_t) clampi(round_magic((x[32 * j + i] + dm) / d), 0, NMAX);
        }
        if (BITS == 4) {
            uint8_t * q = out + 16;
            for (int j = 0; j < 256; j += 64, q += 32)
                for (int l = 0; l < 32; l++) q[l] = (uint8_t) (L[j + l] | (L[j + l + 32] << 4));
        } else {
            uint8_t * qh = out + 16, * ql = out + 48;
            memset(qh, 0, 32);
            uint8_t m1 = 1, m2 = 2;
            for (int n = 0; n < 256; n += 64, ql += 32, m1 = (uint8_t) (m1 << 2), m2 = (uint8_t) (m2 << 2))
                for (int j = 0; j < 32; j++) {
                    int l1 = L[n + j], l2 = L[n + j + 32];
                    if (l1 > 15) { l1 -= 16; qh[j] = (uint8_t) (qh[j] | m1); }
                    if (l2 > 15) { l2 -= 16; qh[j] = (uint8_t) (qh[j] | m2); }
                    ql[j] = (uint8_t) (l1 | (l2 << 4));
                }
        }
    }
}
void row_q6_K(const float * x, uint8_t * out, int64_t k) {                                    // { ql[128], qh[64], scales[16] int8, d }
    int8_t L[256];
    float scales[16];
    for (int64_t b = 0; b < k / 256; b++, x += 256, out += 210) {
        float top = 0, atop = 0;
        for (int j = 0; j < 16; j++) {
            scales[j] = fit_symmetric(16, 32, x + 16 * j, L + 16 * j);
            const float a = fabsf(scales[j]);
            if (a > atop) { atop = a; top = scales[j]; }
        }
        if (!atop) { memset(out, 0, 210); continue; }
        const float inv = -128.f / top;
        const uint16_t dh = float_to_half(1 / inv);
        int8_t * sc = (int8_t *) (out + 192);
        for (int j = 0; j < 16; j++) sc[j] = (int8_t) std::min(127, round_magic(inv * scales[j]));
        for (int j = 0; j < 16; j++) {
            const float d = half_to_float(dh) * sc[j];
            if (!d) continue;
            for (int i = 0; i < 16; i++) L[16 * j + i] = (int8_t) (clampi(round_magic(x[16 * j + i] / d), -32, 31) + 32);
        }
        uint8_t * ql = out, * qh = out + 128;
        for (int j = 0; j < 256; j += 128, ql += 64, qh += 32)
            for (int l = 0; l < 32; l++) {
                const uint8_t a = L[j + l] & 0xF, c = L[j + l + 32] & 0xF, e = L[j + l + 64] & 0xF, g = L[j + l + 96] & 0xF;
                ql[l] = (uint8_t) (a | (e << 4));
                ql[l + 32] = (uint8_t) (c | (g << 4));
                qh[l] = (uint8_t) ((L[j + l] >> 4) | ((L[j + l + 32] >> 4) << 2) | ((L[j + l + 64] >> 4) << 4) | ((L[j + l + 96] >> 4) << 6));
            }
        memcpy(out + 208, &dh, 2);
    }
}

// ---- back to fp32 (dequantize_row_*, ggml-quants.c:1074-1188, 1535-1575, 1711-1757, 1917-1954, 2117-2165, 2263-2310): what --allow-requantize feeds the quantisers.
//      Every value is ONE product (and at most one add or subtract) in the reference's order, so the floats are the reference's bit for bit. ----
void deq_legacy(int type, const uint8_t * in, float * y, int64_t k) {
    const bool affine = type == 3 || type == 7, five = type == 6 || type == 7;
    const int bytes = 2 + (affine ? 2 : 0) + (five ? 4 : 0) + 16;
    for (int64_t b = 0; b < k / 32; b++, in += bytes, y += 32) {
        uint16_t dh, mh = 0; memcpy(&dh, in, 2);
        if (affine) memcpy(&mh, in + 2, 2);
        const float d = half_to_float(dh), m = half_to_float(mh);
        const uint8_t * p = in + 2 + (affine ? 2 : 0);
        uint32_t qh = 0;
        if (five) { memcpy(&qh, p, 4); p += 4; }
        for (int j = 0; j < 16; j++) {
            const int h0 = five ? (int) (((qh >> j) << 4) & 0x10) : 0, h1 = five ? (int) ((qh >> (j + 12)) & 0x10) : 0;
            const int lo = (p[j] & 0x0F) | h0, hi = (p[j] >> 4) | h1;
            if (affine) { y[j] = lo * d + m; y[j + 16] = hi * d + m; }
            else { const int off = five ? 16 : 8; y[j] = (lo - off) * d; y[j + 16] = (hi - off) * d; }
        }
    }
}
void deq_q8_0(const uint8_t * in, float * y, int64_t k) {
    for (int64_t b = 0; b < k / 32; b++, in += 34, y += 32) {
        uint16_t dh; memcpy(&dh, in, 2);
        const float d = half_to_float(dh);
        for (int j = 0; j < 32; j++) y[j] = (int8_t) in[2 + j] * d;
    }
}
void deq_q2_K(const uint8_t * in, float * y, int64_t k) {
    for (int64_t b = 0; b < k / 256; b++, in += 84) {
        uint16_t dh, mh; memcpy(&dh, in + 80, 2); memcpy(&mh, in + 82, 2);
        const float d = half_to_float(dh), mn = half_to_float(mh);
        const uint8_t * sc = in, * q = in + 16;
        int is = 0;
        for (int n = 0; n < 256; n += 128, q += 32)
            for (int j = 0, shift = 0; j < 4; j++, shift += 2)
                for (int half = 0; half < 2; half++) {
                    const uint8_t s8 = sc[is++];
                    const float dl = d * (s8 & 0xF), ml = mn * (s8 >> 4);
                    for (int l = 0; l < 16; l++) *y++ = dl * ((int8_t) ((q[l + 16 * half] >> shift) & 3)) - ml;
                }
    }
}
void deq_q3_K(const uint8_t * in, float * y, int64_t k) {
    for (int64_t b = 0; b < k / 256; b++, in += 110) {
        uint16_t dh; memcpy(&dh, in + 108, 2);
        const float d_all = half_to_float(dh);
        const uint8_t * hm = in, * q = in + 32, * sp = in + 96;
        int8_t s6[16];                                                                        // the sixteen 6-bit scales: low nibbles in bytes 0..7, top two bits in bytes 8..11
        for (int j = 0; j < 16; j++) s6[j] = (int8_t) ((j < 8 ? sp[j] & 0xF : sp[j - 8] >> 4) | (((sp[8 + j % 4] >> (2 * (j / 4))) & 3) << 4));
        int is = 0;
        uint8_t m = 1;
        for (int n = 0; n < 256; n += 128, q += 32)
            for (int j = 0, shift = 0; j < 4; j++, shift += 2, m = (uint8_t) (m << 1))
                for (int half = 0; half < 2; half++) {
                    const float dl = d_all * (s6[is++] - 32);
                    for (int l = 0; l < 16; l++) *y++ = dl * ((int8_t) ((q[l + 16 * half] >> shift) & 3) - ((hm[l + 16 * half] & m) ? 0 : 4));
                }
    }
}
template <int BITS>
void deq_q45_K(const uint8_t * in, float * y, int64_t k) {
    constexpr int BYTES = BITS == 4 ? 144 : 176;
    for (int64_t b = 0; b < k / 256; b++, in += BYTES) {
        uint16_t dh, mh; memcpy(&dh, in, 2); memcpy(&mh, in + 2, 2);
        const float d = half_to_float(dh), mn = half_to_float(mh);
        const uint8_t * sc = in + 4, * qh = in + 16, * ql = in + (BITS == 4 ? 16 : 48);
        uint8_t u1 = 1, u2 = 2;
        for (int j = 0, is = 0; j < 256; j += 64, ql += 32, is += 2, u1 = (uint8_t) (u1 << 2), u2 = (uint8_t) (u2 << 2)) {
            uint8_t s, m;
            unpack_scale_min(is, sc, s, m);
            const float d1 = d * s, m1 = mn * m;
            unpack_scale_min(is + 1, sc, s, m);
            const float d2 = d * s, m2 = mn * m;
            for (int l = 0; l < 32; l++) *y++ = d1 * ((ql[l] & 0xF) + (BITS == 5 && (qh[l] & u1) ? 16 : 0)) - m1;
            for (int l = 0; l < 32; l++) *y++ = d2 * ((ql[l] >> 4) + (BITS == 5 && (qh[l] & u2) ? 16 : 0)) - m2;
        }
    }
}
void deq_q6_K(const uint8_t * in, float * y, int64_t k) {
    for (int64_t b = 0; b < k / 256; b++, in += 210) {
        uint16_t dh; memcpy(&dh, in + 208, 2);
        const float d = half_to_float(dh);
        const uint8_t * ql = in, * qh = in + 128;
        const int8_t * sc = (const int8_t *) (in + 192);
        for (int n = 0; n < 256; n += 128, y += 128, ql += 64, qh += 32, sc += 8)
            for (int l = 0; l < 32; l++) {
                const int is = l / 16;
                const int8_t q1 = (int8_t) ((int8_t) ((ql[l] & 0xF) | (((qh[l] >> 0) & 3) << 4)) - 32), q2 = (int8_t) ((int8_t) ((ql[l + 32] & 0xF) | (((qh[l] >> 2) & 3) << 4)) - 32);
                const int8_t q3 = (int8_t) ((int8_t) ((ql[l] >> 4) | (((qh[l] >> 4) & 3) << 4)) - 32), q4 = (int8_t) ((int8_t) ((ql[l + 32] >> 4) | (((qh[l] >> 6) & 3) << 4)) - 32);
                y[l] = d * sc[is] * q1; y[l + 32] = d * sc[is + 2] * q2; y[l + 64] = d * sc[is + 4] * q3; y[l + 96] = d * sc[is + 6] * q4;
            }
    }
}

enum { T_F32 = 0, T_F16 = 1, T_Q4_0 = 2, T_Q4_1 = 3, T_Q5_0 = 6, T_Q5_1 = 7, T_Q8_0 = 8, T_Q2_K = 10, T_Q3_K = 11, T_Q4_K = 12, T_Q5_K = 13, T_Q6_K = 14 };

int block_elems(int type) { return type >= T_Q2_K ? 256 : (type >= T_Q4_0 ? 32 : 1); }

bool quantize_elems(int type, const float * x, uint8_t * out, int64_t k) {
    switch (type) {
        case T_F32:  memcpy(out, x, (size_t) k * 4); return true;
        case T_F16:  for (int64_t i = 0; i < k; i++) { const uint16_t h = float_to_half(x[i]); memcpy(out + 2 * i, &h, 2); } return true;
        case T_Q4_0: row_legacy<4, false>(x, out, k); return true;
        case T_Q4_1: row_legacy<4, true>(x, out, k); return true;
        case T_Q5_0: row_legacy<5, false>(x, out, k); return true;
        case T_Q5_1: row_legacy<5, true>(x, out, k); return true;
        case T_Q8_0: row_q8_0(x, out, k); return true;
        case T_Q2_K: row_q2_K(x, out, k); return true;
        case T_Q3_K: row_q3_K(x, out, k); return true;
        case T_Q4_K: row_q45_K<4>(x, out, k); return true;
        case T_Q5_K: row_q45_K<5>(x, out, k); return true;
        case T_Q6_K: row_q6_K(x, out, k); return true;
    }
    return false;
}

bool dequantize_elems(int type, const uint8_t * in, float * y, int64_t k) {
    switch (type) {
        case T_F32:  memcpy(y, in, (size_t) k * 4); return true;
        case T_F16:  for (int64_t i = 0; i < k; i++) { uint16_t h; memcpy(&h, in + 2 * i, 2); y[i] = half_to_float(h); } return true;
        case T_Q4_0: case T_Q4_1: case T_Q5_0: case T_Q5_1: deq_legacy(type, in, y, k); return true;
        case T_Q8_0: deq_q8_0(in, y, k); return true;
        case T_Q2_K: deq_q2_K(in, y, k); return true;
        case T_Q3_K: deq_q3_K(in, y, k); return true;
        case T_Q4_K: deq_q45_K<4>(in, y, k); return true;
        case T_Q5_K: deq_q45_K<5>(in, y, k); return true;
        case T_Q6_K: deq_q6_K(in, y, k); return true;
    }
    return false;
}

// ---- the type mixtures (get_k_quant_type, llama.cpp:7896-8006) ------------------------------------------------------------------------
struct mix_state { bool falcon = false, is_70b = false; int n_attn_v = 0, i_attn_v = 0, n_down = 0, i_down = 0, n_fallback = 0; };

bool more_bits(int i, int n) { return i < n / 8 || i >= 7 * n / 8 || (i - n / 8) % 3 == 2; }
bool has(const std::string & s, const char * part) { return s.find(part) != std::string::npos; }

int mixture_type(mix_state & st, int base, const std::string & name, int64_t nx, int ftype) {
    int t = base;
    if (name == "output.weight") {
        if (st.falcon || nx % 256 != 0) t = T_Q8_0; else if (t != T_Q8_0) t = T_Q6_K;
    } else if (has(name, "attn_v.weight")) {
        if (ftype == LLAMA_FTYPE_MOSTLY_Q2_K) t = T_Q3_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_M) t = st.i_attn_v < 2 ? T_Q5_K : T_Q4_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_L) t = T_Q5_K;
        else if ((ftype == LLAMA_FTYPE_MOSTLY_Q4_K_M || ftype == LLAMA_FTYPE_MOSTLY_Q5_K_M) && more_bits(st.i_attn_v, st.n_attn_v)) t = T_Q6_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q4_K_S && st.i_attn_v < 4) t = T_Q5_K;
        if (st.is_70b && (t == T_Q3_K || t == T_Q4_K)) t = T_Q5_K;                            // eight heads share attn_v: more bits cost almost nothing
        st.i_attn_v++;
    } else if (has(name, "ffn_down.weight")) {
        if (ftype == LLAMA_FTYPE_MOSTLY_Q2_K) t = T_Q3_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_M) t = st.i_down < 2 ? T_Q5_K : (!st.falcon || more_bits(st.i_down, st.n_down) ? T_Q4_K : T_Q3_K);
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_L) t = st.falcon ? T_Q4_K : T_Q5_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q4_K_M) {
            if (st.falcon) t = st.i_down < 2 ? T_Q6_K : (more_bits(st.i_down, st.n_down) ? T_Q5_K : T_Q4_K);
            else if (more_bits(st.i_down, st.n_down)) t = T_Q6_K;
        }
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q5_K_M && more_bits(st.i_down, st.n_down)) t = T_Q6_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q4_K_S && !st.falcon && st.i_down < 4) t = T_Q5_K;
        st.i_down++;
    } else if (has(name, "attn_output.weight")) {
        if (!st.falcon) {
            if (ftype == LLAMA_FTYPE_MOSTLY_Q2_K) t = T_Q3_K;
            else if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_M) t = T_Q4_K;
            else if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_L) t = T_Q5_K;
        } else if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_L) t = T_Q4_K;
    } else if (has(name, "attn_qkv.weight")) {
        if (ftype == LLAMA_FTYPE_MOSTLY_Q3_K_M || ftype == LLAMA_FTYPE_MOSTLY_Q3_K_L) t = T_Q4_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q4_K_M) t = T_Q5_K;
        else if (ftype == LLAMA_FTYPE_MOSTLY_Q5_K_M) t = T_Q6_K;
    } else if (has(name, "ffn_gate.weight") || has(name, "ffn_up.weight")) {
        if (ftype == LLAMA_FTYPE_MOSTLY_Q2_K) t = T_Q3_K;
    }
    if (t >= T_Q2_K && t <= T_Q6_K && nx % 256 != 0) {                                       // rows that are not whole super-blocks: the 32-value format of similar size
        static const int fallback[5] = { T_Q4_0, T_Q4_1, T_Q5_0, T_Q5_1, T_Q8_0 };
        t = fallback[t - T_Q2_K];
        st.n_fallback++;
    }
    return t;
}

struct out_buf {
    std::vector<uint8_t> b;
    template <typename T> void put(T v) { const size_t o = b.size(); b.resize(o + sizeof(T)); memcpy(b.data() + o, &v, sizeof(T)); }
    void str(const std::string & s) { put<uint64_t>(s.size()); b.insert(b.end(), s.begin(), s.end()); }
    void raw(const uint8_t * p, const uint8_t * e) { b.insert(b.end(), p, e); }
};

}  // namespace

// bytes written, or -1 (unknown type / k not a whole number of blocks)
extern "C" int64_t pipeinfer_quantize_row(int32_t type, const float * x, void * y, int64_t k) {
    const int be = block_elems(type);
    if (k < 0 || k % be != 0 || pi::type_row_size(type, be) == 0) return -1;
    if (!quantize_elems(type, x, (uint8_t *) y, k)) return -1;
    return (int64_t) pi::type_row_size(type, k);
}

// fp32 values of k stored values of `type` (dequantize_row_*); returns k, -1 on a bad type / length
extern "C" int64_t pipeinfer_dequantize_row(int32_t type, const void * x, float * y, int64_t k) {
    const int be = block_elems(type);
    if (k < 0 || k % be != 0 || pi::type_row_size(type, be) == 0) return -1;
    return dequantize_elems(type, (const uint8_t *) x, y, k) ? k : -1;
}

extern "C" int llama_model_quantize(const char * fname_inp, const char * fname_out, const llama_model_quantize_params * params) {
    int base;
    switch (params->ftype) {                                                                  // llama.cpp:8013-8036
        case LLAMA_FTYPE_ALL_F32: base = T_F32; break;       case LLAMA_FTYPE_MOSTLY_F16: base = T_F16; break;
        case LLAMA_FTYPE_MOSTLY_Q4_0: base = T_Q4_0; break;  case LLAMA_FTYPE_MOSTLY_Q4_1: base = T_Q4_1; break;
        case LLAMA_FTYPE_MOSTLY_Q5_0: base = T_Q5_0; break;  case LLAMA_FTYPE_MOSTLY_Q5_1: base = T_Q5_1; break;
        case LLAMA_FTYPE_MOSTLY_Q8_0: base = T_Q8_0; break;  case LLAMA_FTYPE_MOSTLY_Q2_K: base = T_Q2_K; break;
        case LLAMA_FTYPE_MOSTLY_Q3_K_S: case LLAMA_FTYPE_MOSTLY_Q3_K_M: case LLAMA_FTYPE_MOSTLY_Q3_K_L: base = T_Q3_K; break;
        case LLAMA_FTYPE_MOSTLY_Q4_K_S: case LLAMA_FTYPE_MOSTLY_Q4_K_M: base = T_Q4_K; break;
        case LLAMA_FTYPE_MOSTLY_Q5_K_S: case LLAMA_FTYPE_MOSTLY_Q5_K_M: base = T_Q5_K; break;
        case LLAMA_FTYPE_MOSTLY_Q6_K: base = T_Q6_K; break;
        default: fprintf(stderr, "llama_model_quantize: invalid output file type %d\n", (int) params->ftype); return 1;
    }
    pi::gguf_file in;
    std::string err;
    if (!in.open(fname_inp, err)) { fprintf(stderr, "llama_model_quantize: %s\n", err.c_str()); return 1; }
    const std::string arch = in.get_str("general.architecture", "");
    mix_state st;
    st.falcon = arch == "falcon";
    const uint32_t n_layer = in.get_u32(arch + ".block_count", 0), n_head = in.get_u32(arch + ".attention.head_count", 0);
    st.is_70b = arch == "llama" && n_layer == 80 && in.get_u32(arch + ".attention.head_count_kv", n_head) != n_head;     // MODEL_70B (llama.cpp:2284-2296)
    for (const auto & t : in.tensors) {
        if (has(t.name, "attn_v.weight") || has(t.name, "attn_qkv.weight")) st.n_attn_v++;
        else if (has(t.name, "ffn_down.weight")) st.n_down++;
    }
    if (st.n_attn_v != st.n_down || (uint32_t) st.n_attn_v != n_layer)
        fprintf(stderr, "llama_model_quantize: strange model: n_attention_wv = %d, n_feed_forward_w2 = %d, n_layer = %u\n", st.n_attn_v, st.n_down, n_layer);
    int ftype = (int) params->ftype;
    if (params->only_copy) ftype = (int) in.get_u32("general.file_type", (uint32_t) ftype);
    int nthread = params->nthread > 0 ? params->nthread : (int) std::thread::hardware_concurrency();
    if (nthread < 1) nthread = 1;

    // pass 1: the new type and size of every tensor (the header holds the offsets, so it is complete before any data is written)
    struct plan { int type; size_t bytes; bool convert; };
    std::vector<plan> plans(in.tensors.size());
    for (size_t i = 0; i < in.tensors.size(); i++) {
        const pi::gguf_tensor & t = in.tensors[i];
        bool q = t.name.size() >= 6 && t.name.compare(t.name.size() - 6, 6, "weight") == 0 && t.n_dims == 2;     // llama.cpp:8137-8141
        q = q && (params->quantize_output_tensor || t.name != "output.weight") && !params->only_copy;
        int nt = t.type;
        if (q) {
            nt = base;
            if (!params->pure) nt = mixture_type(st, nt, t.name, t.ne[0], ftype);
            q = nt != t.type;
        }
        if (q && t.type != T_F32 && t.type != T_F16 && !params->allow_requantize) {                         // llama.cpp:8159-8161
            fprintf(stderr, "llama_model_quantize: requantizing %s from type %d is disabled\n", t.name.c_str(), t.type);
            return 1;
        }
        if (q && t.ne[0] % block_elems(nt) != 0) { fprintf(stderr, "llama_model_quantize: %s: %lld columns are not a multiple of %d\n", t.name.c_str(), (long long) t.ne[0], block_elems(nt)); return 1; }
        plans[i] = { q ? nt : t.type, q ? pi::type_row_size(nt, t.ne[0]) * (size_t) (t.ne[1] * t.ne[2] * t.ne[3]) : t.nbytes, q };
    }
    // header: every key of the input in its order; quantization_version and file_type replaced in place or appended (gguf_set_val_u32)
    out_buf h;
    bool have_qv = false, have_ft = false;
    for (const auto & kvr : in.kv_raw) { have_qv |= kvr.key == "general.quantization_version"; have_ft |= kvr.key == "general.file_type"; }
    h.put<uint32_t>(0x46554747u); h.put<uint32_t>(3); h.put<uint64_t>(in.tensors.size()); h.put<uint64_t>(in.kv_raw.size() + !have_qv + !have_ft);
    auto put_u32_kv = [&](const std::string & key, uint32_t v) { h.str(key); h.put<uint32_t>(4); h.put<uint32_t>(v); };
    for (const auto & kvr : in.kv_raw) {
        if (kvr.key == "general.quantization_version") put_u32_kv(kvr.key, 2);                // GGML_QNT_VERSION
        else if (kvr.key == "general.file_type") put_u32_kv(kvr.key, (uint32_t) ftype);
        else { h.str(kvr.key); h.raw(kvr.begin, kvr.end); }
    }
    if (!have_qv) put_u32_kv("general.quantization_version", 2);
    if (!have_ft) put_u32_kv("general.file_type", (uint32_t) ftype);
    const size_t align = in.get_u32("general.alignment", 32);
    size_t off = 0;
    for (size_t i = 0; i < in.tensors.size(); i++) {
        const pi::gguf_tensor & t = in.tensors[i];
        h.str(t.name); h.put<uint32_t>((uint32_t) t.n_dims);
        for (int d = 0; d < t.n_dims; d++) h.put<uint64_t>((uint64_t) t.ne[d]);
        h.put<uint32_t>((uint32_t) plans[i].type); h.put<uint64_t>(off);
        off += (plans[i].bytes + align - 1) / align * align;
    }
    h.b.resize((h.b.size() + align - 1) / align * align, 0);
    FILE * f = fopen(fname_out, "wb");
    if (!f) { fprintf(stderr, "llama_model_quantize: cannot write %s\n", fname_out); return 1; }
    bool ok = fwrite(h.b.data(), 1, h.b.size(), f) == h.b.size();

    // pass 2: rows are independent; threads take chunks of rows
    std::vector<uint8_t> work;
    std::vector<float> f32;
    size_t total_in = 0, total_out = 0;
    for (size_t i = 0; i < in.tensors.size() && ok; i++) {
        const pi::gguf_tensor & t = in.tensors[i];
        const uint8_t * data = t.data;
        if (plans[i].convert) {
            const int64_t rows = t.ne[1] * t.ne[2] * t.ne[3], k = t.ne[0];
            const size_t row_out = pi::type_row_size(plans[i].type, k);
            work.resize(plans[i].bytes);
            const float * src = (const float *) t.data;
            if (t.type != T_F32) {                                                                            // llama_convert_tensor_internal (llama.cpp:7838-7893): f16 or, with allow_requantize, any block format
                f32.resize((size_t) rows * k);
                if (!dequantize_elems(t.type, t.data, f32.data(), rows * k)) { fprintf(stderr, "llama_model_quantize: cannot read %s (type %d)\n", t.name.c_str(), t.type); fclose(f); return 1; }
                src = f32.data();
            }
            std::atomic<int64_t> next(0);
            const int64_t step = std::max<int64_t>(1, 32 * 512 / std::max<int64_t>(1, k));
            const int type = plans[i].type;
            uint8_t * dst = work.data();
            auto job = [&]() {
                for (;;) {
                    const int64_t r0 = next.fetch_add(step);
                    if (r0 >= rows) break;
                    const int64_t r1 = std::min(rows, r0 + step);
                    quantize_elems(type, src + r0 * k, dst + (size_t) r0 * row_out, (r1 - r0) * k);
                }
            };
            const int nt_use = (int) std::min<int64_t>(nthread, (rows + step - 1) / step);
            std::vector<std::thread> pool;
            for (int w = 1; w < nt_use; w++) pool.emplace_back(job);
            job();
            for (auto & w : pool) w.join();
            data = work.data();
        }
        ok = ok && fwrite(data, 1, plans[i].bytes, f) == plans[i].bytes;
        static const uint8_t zeros[64] = { 0 };
        const size_t pad = (plans[i].bytes + align - 1) / align * align - plans[i].bytes;
        for (size_t p = 0; p < pad && ok; p += sizeof zeros) ok = fwrite(zeros, 1, std::min(sizeof zeros, pad - p), f) == std::min(sizeof zeros, pad - p);
        total_in += t.nbytes; total_out += plans[i].bytes;
    }
    ok = fclose(f) == 0 && ok;
    if (!ok) { fprintf(stderr, "llama_model_quantize: write to %s failed\n", fname_out); return 1; }
    fprintf(stderr, "llama_model_quantize: model size = %8.2f MB, quant size = %8.2f MB\n", total_in / 1024.0 / 1024.0, total_out / 1024.0 / 1024.0);
    if (st.n_fallback > 0) fprintf(stderr, "llama_model_quantize: WARNING: %d tensor(s) incompatible with k-quants required fallback quantization\n", st.n_fallback);
    return 0;
}
