// TEST INFRASTRUCTURE — see oracle.h.  Block formats, activation quantisers, dequantisers and the quantised
// dot products of the reference CPU path, restated in plain scalar C with the AVX2 arm's lane structure.
#include "oracle.h"

#include <math.h>
#include <string.h>

// ---- fp16 <-> fp32: IEEE binary16, round-to-nearest-even, as F16C's _cvtss_sh(x,0)/_cvtsh_ss (ggml-impl.h:88-92)
float o_fp16_to_fp32(uint16_t h) {
    const uint32_t sign = (uint32_t) (h & 0x8000u) << 16;
    const uint32_t exp  = (h >> 10) & 0x1f;
    uint32_t mant = h & 0x3ffu;
    uint32_t bits;
    if (exp == 0) {
        if (mant == 0) {
            bits = sign;
        } else {                       // subnormal: normalise
            int e = -1;
            do { e++; mant <<= 1; } while (!(mant & 0x400u));
            bits = sign | (uint32_t) (112 - e) << 23 | (mant & 0x3ffu) << 13;
        }
    } else if (exp == 31) {
        bits = sign | 0x7f800000u | mant << 13;
    } else {
        bits = sign | (exp + 112) << 23 | mant << 13;
    }
    float f; memcpy(&f, &bits, 4);
    return f;
}

uint16_t o_fp32_to_fp16(float f) {
    uint32_t x; memcpy(&x, &f, 4);
    const uint16_t sign = (uint16_t) ((x >> 16) & 0x8000u);
    x &= 0x7fffffffu;
    if (x > 0x7f800000u) return (uint16_t) (sign | 0x7e00u | ((x >> 13) & 0x3ffu));   // NaN
    if (x >= 0x47800000u) return (uint16_t) (sign | 0x7c00u);                          // >= 65536 or inf
    const uint32_t exp  = x >> 23;
    const uint32_t mant = (x & 0x7fffffu) | 0x800000u;
    if (exp < 113) {                                                                   // result is subnormal/zero
        if (exp < 102) return sign;
        const int shift = 126 - (int) exp;                                             // 14..24
        uint32_t q = mant >> shift;
        const uint32_t rem = mant & ((1u << shift) - 1), half = 1u << (shift - 1);
        if (rem > half || (rem == half && (q & 1))) q++;
        return (uint16_t) (sign | q);
    }
    uint32_t h = (exp - 112) << 10 | (mant & 0x7fffffu) >> 13;
    const uint32_t rem = mant & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1))) h++;                             // carry may roll into inf: correct
    return (uint16_t) (sign | h);
}

int o_blck_size(int type) {
    switch (type) {
        case OT_F32: case OT_F16: return 1;
        case OT_Q4_0: case OT_Q4_1: case OT_Q5_0: case OT_Q5_1: case OT_Q8_0: case OT_Q8_1: return OQK;
        case OT_Q2_K: case OT_Q3_K: case OT_Q4_K: case OT_Q5_K: case OT_Q6_K: case OT_Q8_K: return OQK_K;
    }
    return 0;
}

int o_type_size(int type) {
    switch (type) {
        case OT_F32: return 4; case OT_F16: return 2;
        case OT_Q4_0: return sizeof(oblk_q4_0); case OT_Q8_0: return sizeof(oblk_q8_0);
        case OT_Q4_1: return sizeof(oblk_q4_1); case OT_Q5_0: return sizeof(oblk_q5_0); case OT_Q5_1: return sizeof(oblk_q5_1); case OT_Q8_1: return sizeof(oblk_q8_1);
        case OT_Q4_K: return sizeof(oblk_q4_K); case OT_Q5_K: return sizeof(oblk_q5_K);
        case OT_Q6_K: return sizeof(oblk_q6_K); case OT_Q8_K: return sizeof(oblk_q8_K); case OT_Q3_K: return sizeof(oblk_q3_K); case OT_Q2_K: return sizeof(oblk_q2_K);
    }
    return 0;
}

size_t o_row_size(int type, int64_t n) { return (size_t) (n / o_blck_size(type)) * o_type_size(type); }

// ggml.c:415-600 type_traits[].vec_dot_type
int o_vec_dot_type(int type) {
    switch (type) {
        case OT_F32: return OT_F32; case OT_F16: return OT_F16;
        case OT_Q4_0: case OT_Q5_0: case OT_Q8_0: return OT_Q8_0;
        case OT_Q4_1: case OT_Q5_1: return OT_Q8_1;                       // type_traits[].vec_dot_type, ggml.c:436-470
        case OT_Q2_K: case OT_Q3_K: case OT_Q4_K: case OT_Q5_K: case OT_Q6_K: return OT_Q8_K;
    }
    return -1;
}

// ggml-quants.c:1191-1196 — round-half-even through the 1.5*2^23 magic constant
static inline int nearest_int(float fval) {
    float val = fval + 12582912.f;
    int i; memcpy(&i, &val, sizeof(int));
    return (i & 0x007fffff) - 0x00400000;
}

// ggml-quants.c:2329-2364 (quantize_row_q8_K_reference; there is no SIMD variant)
void o_quantize_row_q8_K(const float * x, void * vy, int k) {
    oblk_q8_K * y = vy;
    const int nb = k / OQK_K;
    for (int i = 0; i < nb; i++, x += OQK_K) {
        float max = 0, amax = 0;
        for (int j = 0; j < OQK_K; ++j) {
            const float ax = fabsf(x[j]);
            if (ax > amax) { amax = ax; max = x[j]; }
        }
        if (!amax) {
            // (the reference leaves bsums untouched here; activations are never all-zero in practice. We zero them.)
            y[i].d = 0; memset(y[i].qs, 0, OQK_K); memset(y[i].bsums, 0, sizeof y[i].bsums);
            continue;
        }
        const float iscale = -128.f / max;
        for (int j = 0; j < OQK_K; ++j) {
            const int v = nearest_int(iscale * x[j]);
            y[i].qs[j] = (int8_t) (v < 127 ? v : 127);
        }
        for (int j = 0; j < OQK_K / 16; ++j) {
            int sum = 0;
            for (int ii = 0; ii < 16; ++ii) sum += y[i].qs[j * 16 + ii];
            y[i].bsums[j] = (int16_t) sum;
        }
        y[i].d = 1 / iscale;
    }
}

// ggml-quants.c:702-760 — the AVX/AVX2 arm of quantize_row_q8_0: d = amax/127 (stored f16), id = 127/amax,
// round-half-even (_mm256_round_ps NEAREST).  (The scalar arm :607-630 uses 1/d and roundf; we follow AVX.)
void o_quantize_row_q8_0(const float * x, void * vy, int k) {
    oblk_q8_0 * y = vy;
    const int nb = k / OQK;
    for (int i = 0; i < nb; i++, x += OQK) {
        float amax = 0.0f;
        for (int j = 0; j < OQK; j++) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float d  = amax / 127.f;
        y[i].d = o_fp32_to_fp16(d);
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        for (int j = 0; j < OQK; ++j) {
            int v = (int) nearbyintf(x[j] * id);
            if (v > 127) v = 127; else if (v < -128) v = -128;       // packs_epi32/epi16 saturate
            y[i].qs[j] = (int8_t) v;
        }
    }
}

// ggml-quants.c:940-1000 — the AVX2 arm of quantize_row_q8_1: as Q8_0 but d stays a float and s = d * (sum of the 32 quants)
void o_quantize_row_q8_1(const float * x, void * vy, int k) {
    oblk_q8_1 * y = vy;
    const int nb = k / OQK;
    for (int i = 0; i < nb; i++, x += OQK) {
        float amax = 0.0f;
        for (int j = 0; j < OQK; j++) { const float a = fabsf(x[j]); if (a > amax) amax = a; }
        const float d  = amax / 127.f;
        y[i].d = d;
        const float id = (amax != 0.0f) ? 127.f / amax : 0.0f;
        int sum = 0;
        for (int j = 0; j < OQK; ++j) {
            int v = (int) nearbyintf(x[j] * id);
            sum += v;                                                 // the sum is taken over the int32 lanes, before packs saturate (|v| <= 127 anyway)
            if (v > 127) v = 127; else if (v < -128) v = -128;
            y[i].qs[j] = (int8_t) v;
        }
        y[i].s = d * (float) sum;
    }
}

void o_quantize_act(int vec_dot_type, const float * x, void * y, int k) {
    if (vec_dot_type == OT_Q8_K) o_quantize_row_q8_K(x, y, k); else if (vec_dot_type == OT_Q8_1) o_quantize_row_q8_1(x, y, k); else o_quantize_row_q8_0(x, y, k);
}

// element b (0..31) of a legacy 32-block as the integer the dot product multiplies: low nibbles are elements 0..15, high nibbles 16..31
// (bytes_from_nibbles_32, ggml-quants.c:92-99); Q5: bit b of qh is the fifth bit (bytes_from_bits_32 :76-90)
static inline int q41_at(const oblk_q4_1 * x, int b) { return b < 16 ? x->qs[b] & 0xF : x->qs[b - 16] >> 4; }
static inline int q5_nib(const uint8_t * qs, const uint8_t * qh, int b) { uint32_t h; memcpy(&h, qh, 4); return (b < 16 ? qs[b] & 0xF : qs[b - 16] >> 4) | (int) (((h >> b) & 1) << 4); }

// ggml-quants.c:1446-1453
static inline void scale_min_k4(int j, const uint8_t * q, uint8_t * d, uint8_t * m) {
    if (j < 4) { *d = q[j] & 63; *m = q[j + 4] & 63; }
    else       { *d = (q[j + 4] & 0xF) | ((q[j - 4] >> 6) << 4); *m = (q[j + 4] >> 4) | ((q[j] >> 6) << 4); }
}

// element e (0..255) of a Q4_K / Q5_K / Q6_K super-block as the integer the dot product multiplies
static inline int q4k_at(const oblk_q4_K * b, int e) { const int j = e >> 6, h = (e >> 5) & 1, l = e & 31; const uint8_t q = b->qs[32 * j + l]; return h ? q >> 4 : q & 0xF; }
static inline int q5k_at(const oblk_q5_K * b, int e) {
    const int j = e >> 6, h = (e >> 5) & 1, l = e & 31; const uint8_t q = b->qs[32 * j + l];
    return (h ? q >> 4 : q & 0xF) + (((b->qh[l] >> (2 * j + h)) & 1) << 4);
}
static inline int q6k_at(const oblk_q6_K * b, int e) {                        // 0..63 (before the -32)
    const int n = e >> 7, g = (e >> 5) & 3, l = e & 31;
    const uint8_t lo = b->ql[64 * n + 32 * (g & 1) + l];
    return ((g & 2) ? lo >> 4 : lo & 0xF) | (((b->qh[32 * n + l] >> (2 * g)) & 3) << 4);
}

// Q2_K (ggml-quants.c:1495-1540): sixteen bytes of (4-bit scale | 4-bit min << 4), 2-bit weights 0..3: x = d*sc*q - dmin*m
static inline int q2k_at(const oblk_q2_K * b, int e) { const int n = e >> 7, j = (e >> 5) & 3, l = e & 31; return (b->qs[32 * n + l] >> (2 * j)) & 3; }

// Q3_K (ggml-quants.c:1711-1759): sixteen 6-bit scales packed in 12 bytes (stored + 32), weight = (2 low bits | high bit << 2) - 4
static inline void q3k_scales(const oblk_q3_K * b, int8_t * sc) {
    const uint32_t kmask1 = 0x03030303, kmask2 = 0x0f0f0f0f;
    uint32_t a[3], aux[4];
    memcpy(a, b->scales, 12);
    aux[0] = (a[0] & kmask2) | (((a[2] >> 0) & kmask1) << 4);
    aux[1] = (a[1] & kmask2) | (((a[2] >> 2) & kmask1) << 4);
    aux[2] = ((a[0] >> 4) & kmask2) | (((a[2] >> 4) & kmask1) << 4);
    aux[3] = ((a[1] >> 4) & kmask2) | (((a[2] >> 6) & kmask1) << 4);
    memcpy(sc, aux, 16);
    for (int i = 0; i < 16; i++) sc[i] = (int8_t) (sc[i] - 32);
}
static inline int q3k_at(const oblk_q3_K * b, int e) {                        // -4..3
    const int n = e >> 7, j = (e >> 5) & 3, l = e & 31;
    return ((b->qs[32 * n + l] >> (2 * j)) & 3) - ((b->hmask[l] & (1 << (4 * n + j))) ? 0 : 4);
}

// ggml-quants.c:1074-1092, 1168-1182, 1711-1759, 1917-1954, 2117-2160, 2263-2308
void o_dequantize_row(int type, const void * vx, float * y, int k) {
    switch (type) {
        case OT_F32: memcpy(y, vx, (size_t) 4 * k); break;
        case OT_F16: { const uint16_t * x = vx; for (int i = 0; i < k; i++) y[i] = o_fp16_to_fp32(x[i]); } break;
        case OT_Q4_0: {
            const oblk_q4_0 * x = vx;
            for (int i = 0; i < k / OQK; i++) {
                const float d = o_fp16_to_fp32(x[i].d);
                for (int j = 0; j < 16; ++j) {
                    y[i * OQK + j]      = ((x[i].qs[j] & 0x0F) - 8) * d;
                    y[i * OQK + j + 16] = ((x[i].qs[j] >> 4) - 8) * d;
                }
            }
        } break;
        case OT_Q4_1: {                                                          // ggml-quants.c:1094-1113
            const oblk_q4_1 * x = vx;
            for (int i = 0; i < k / OQK; i++) {
                const float d = o_fp16_to_fp32(x[i].d), m = o_fp16_to_fp32(x[i].m);
                for (int b = 0; b < OQK; b++) y[i * OQK + b] = q41_at(&x[i], b) * d + m;
            }
        } break;
        case OT_Q5_0: {                                                          // :1115-1139
            const oblk_q5_0 * x = vx;
            for (int i = 0; i < k / OQK; i++) {
                const float d = o_fp16_to_fp32(x[i].d);
                for (int b = 0; b < OQK; b++) y[i * OQK + b] = (q5_nib(x[i].qs, x[i].qh, b) - 16) * d;
            }
        } break;
        case OT_Q5_1: {                                                          // :1141-1166
            const oblk_q5_1 * x = vx;
            for (int i = 0; i < k / OQK; i++) {
                const float d = o_fp16_to_fp32(x[i].d), m = o_fp16_to_fp32(x[i].m);
                for (int b = 0; b < OQK; b++) y[i * OQK + b] = q5_nib(x[i].qs, x[i].qh, b) * d + m;
            }
        } break;
        case OT_Q8_0: {
            const oblk_q8_0 * x = vx;
            for (int i = 0; i < k / OQK; i++) {
                const float d = o_fp16_to_fp32(x[i].d);
                for (int j = 0; j < OQK; ++j) y[i * OQK + j] = x[i].qs[j] * d;
            }
        } break;
        case OT_Q4_K: {
            const oblk_q4_K * x = vx;
            for (int i = 0; i < k / OQK_K; i++) {
                const float d = o_fp16_to_fp32(x[i].d), min = o_fp16_to_fp32(x[i].dmin);
                for (int s = 0; s < 8; s++) {
                    uint8_t sc, m; scale_min_k4(s, x[i].scales, &sc, &m);
                    const float d1 = d * sc, m1 = min * m;
                    for (int l = 0; l < 32; ++l) *y++ = d1 * q4k_at(&x[i], 32 * s + l) - m1;
                }
            }
        } break;
        case OT_Q5_K: {
            const oblk_q5_K * x = vx;
            for (int i = 0; i < k / OQK_K; i++) {
                const float d = o_fp16_to_fp32(x[i].d), min = o_fp16_to_fp32(x[i].dmin);
                for (int s = 0; s < 8; s++) {
                    uint8_t sc, m; scale_min_k4(s, x[i].scales, &sc, &m);
                    const float d1 = d * sc, m1 = min * m;
                    for (int l = 0; l < 32; ++l) *y++ = d1 * q5k_at(&x[i], 32 * s + l) - m1;
                }
            }
        } break;
        case OT_Q2_K: {
            const oblk_q2_K * x = vx;
            for (int i = 0; i < k / OQK_K; i++) {
                const float d = o_fp16_to_fp32(x[i].d), min = o_fp16_to_fp32(x[i].dmin);
                for (int e = 0; e < OQK_K; e++) {
                    const uint8_t sc = x[i].scales[e >> 4];
                    const float dl = d * (sc & 0xF), ml = min * (sc >> 4);
                    *y++ = dl * q2k_at(&x[i], e) - ml;
                }
            }
        } break;
        case OT_Q3_K: {
            const oblk_q3_K * x = vx;
            for (int i = 0; i < k / OQK_K; i++) {
                const float d_all = o_fp16_to_fp32(x[i].d);
                int8_t sc[16]; q3k_scales(&x[i], sc);
                for (int e = 0; e < OQK_K; e++) { const float dl = d_all * sc[e >> 4]; *y++ = dl * q3k_at(&x[i], e); }
            }
        } break;
        case OT_Q6_K: {
            const oblk_q6_K * x = vx;
            for (int i = 0; i < k / OQK_K; i++) {
                const float d = o_fp16_to_fp32(x[i].d);
                for (int e = 0; e < OQK_K; e++) {
                    const int8_t q = (int8_t) q6k_at(&x[i], e) - 32;
                    *y++ = d * x[i].scales[e >> 4] * q;
                }
            }
        } break;
    }
}

// hsum_float_8 (ggml-quants.c:67-73): (a4+a0, a5+a1, a6+a2, a7+a3) -> (r0+r2, r1+r3) -> sum
static inline float hsum8(const float a[8]) {
    const float r0 = a[4] + a[0], r1 = a[5] + a[1], r2 = a[6] + a[2], r3 = a[7] + a[3];
    const float s0 = r0 + r2, s1 = r1 + r3;
    return s0 + s1;
}

// per-lane integer sums of one block: lane k (0..7) covers bytes 4k..4k+3 of every 32-element group,
// the grouping _mm256_maddubs_epi16 + _mm256_madd_epi16 produce.
void o_block_isums(int type, const void * xb, const void * yb, int32_t * isum, int32_t * msum) {
    for (int k = 0; k < 8; k++) isum[k] = 0;
    *msum = 0;
    if (type == OT_Q4_1 || type == OT_Q5_1) {                                    // x Q8_1: unsigned weights
        const oblk_q8_1 * y = yb;
        for (int b = 0; b < 32; b++) {
            const int xv = type == OT_Q4_1 ? q41_at((const oblk_q4_1 *) xb, b) : q5_nib(((const oblk_q5_1 *) xb)->qs, ((const oblk_q5_1 *) xb)->qh, b);
            isum[b >> 2] += xv * y->qs[b];
        }
        return;
    }
    if (type == OT_Q5_0) {
        const oblk_q8_0 * y = yb; const oblk_q5_0 * x = xb;
        for (int b = 0; b < 32; b++) isum[b >> 2] += (q5_nib(x->qs, x->qh, b) - 16) * y->qs[b];
        return;
    }
    if (type == OT_Q4_0 || type == OT_Q8_0) {
        const oblk_q8_0 * y = yb;
        for (int b = 0; b < 32; b++) {
            int xv;
            if (type == OT_Q4_0) { const oblk_q4_0 * x = xb; xv = (b < 16 ? x->qs[b] & 0xF : x->qs[b - 16] >> 4) - 8; }
            else                 { const oblk_q8_0 * x = xb; xv = x->qs[b]; }
            isum[b >> 2] += xv * y->qs[b];
        }
        return;
    }
    const oblk_q8_K * y = yb;
    if (type == OT_Q6_K) {
        const oblk_q6_K * x = xb;
        for (int e = 0; e < OQK_K; e++) isum[(e & 31) >> 2] += x->scales[e >> 4] * (q6k_at(x, e) - 32) * y->qs[e];
        return;
    }
    if (type == OT_Q2_K) {                                                       // ggml-quants.c:3708-3771; *msum is unused: the mins go lane by lane (o_vec_dot)
        const oblk_q2_K * x = xb;
        for (int e = 0; e < OQK_K; e++) isum[(e & 31) >> 2] += (x->scales[e >> 4] & 0xF) * q2k_at(x, e) * y->qs[e];
        return;
    }
    if (type == OT_Q3_K) {                                                       // ggml-quants.c:4443-4533: same lane grouping, scale of element e = sc[e >> 4]
        const oblk_q3_K * x = xb;
        int8_t sc[16]; q3k_scales(x, sc);
        for (int e = 0; e < OQK_K; e++) isum[(e & 31) >> 2] += sc[e >> 4] * q3k_at(x, e) * y->qs[e];
        return;
    }
    const uint8_t * scales = type == OT_Q4_K ? ((const oblk_q4_K *) xb)->scales : ((const oblk_q5_K *) xb)->scales;
    for (int s = 0; s < 8; s++) {
        uint8_t sc, m; scale_min_k4(s, scales, &sc, &m);
        for (int l = 0; l < 32; l++) {
            const int e = 32 * s + l;
            const int q = type == OT_Q4_K ? q4k_at(xb, e) : q5k_at(xb, e);
            isum[l >> 2] += sc * q * y->qs[e];
        }
        *msum += m * (y->bsums[2 * s] + y->bsums[2 * s + 1]);
    }
}

// ggml_vec_dot_q4_K_q8_K AVX2 arm (ggml-quants.c:5318-5381), q5_K (:5967-6050), q6_K (:6741-6810), q3_K (:4443-4533),
// q4_0 (:2500-2523), q8_0 (:3584-3605)
float o_vec_dot(int type, int n, const void * vx, const void * vy) {
    float acc[8] = { 0 };
    if (type == OT_F16) return 0.0f; // see ops.c o_vec_dot_f16
    if (type == OT_F32) { // ggml_vec_dot_f32 AVX arm (ggml.c:1157-1191): 4 accumulators x 8 lanes
        const float * x = vx, * y = vy;
        float sum[4][8] = { { 0 } };
        const int np = n & ~31;
        for (int i = 0; i < np; i += 32)
            for (int j = 0; j < 4; j++)
                for (int l = 0; l < 8; l++) sum[j][l] = fmaf(x[i + 8 * j + l], y[i + 8 * j + l], sum[j][l]);
        for (int l = 0; l < 8; l++) { sum[0][l] += sum[2][l]; sum[1][l] += sum[3][l]; }
        for (int l = 0; l < 8; l++) sum[0][l] += sum[1][l];
        float t[4]; for (int l = 0; l < 4; l++) t[l] = sum[0][l] + sum[0][l + 4];
        float sumf = (t[0] + t[1]) + (t[2] + t[3]);
        for (int i = np; i < n; ++i) sumf += x[i] * y[i];
        return sumf;
    }
    if (type == OT_Q4_1 || type == OT_Q5_1) {        // ggml-quants.c:2808-2842, 3372-3398 (AVX2): acc = fmadd(d0*d1, xy, acc); summs += m * y.s; *s = hsum_float_8(acc) + summs
        const int nb = n / OQK;
        const oblk_q8_1 * y = vy;
        float summs = 0.0f;
        for (int i = 0; i < nb; i++) {
            int32_t is[8], ms;
            const void * xb = type == OT_Q4_1 ? (const void *) ((const oblk_q4_1 *) vx + i) : (const void *) ((const oblk_q5_1 *) vx + i);
            const uint16_t xd = type == OT_Q4_1 ? ((const oblk_q4_1 *) vx)[i].d : ((const oblk_q5_1 *) vx)[i].d;
            const uint16_t xm = type == OT_Q4_1 ? ((const oblk_q4_1 *) vx)[i].m : ((const oblk_q5_1 *) vx)[i].m;
            o_block_isums(type, xb, &y[i], is, &ms);
            const float prod = o_fp16_to_fp32(xm) * y[i].s;
            summs += prod;
            const float d = o_fp16_to_fp32(xd) * y[i].d;
            for (int k = 0; k < 8; k++) acc[k] = fmaf(d, (float) is[k], acc[k]);
        }
        return hsum8(acc) + summs;
    }
    if (type == OT_Q5_0) {                            // ggml-quants.c:3056-3078: like Q4_0 with the fifth bit
        const int nb = n / OQK;
        const oblk_q8_0 * y = vy; const oblk_q5_0 * x = vx;
        for (int i = 0; i < nb; i++) {
            int32_t is[8], ms;
            o_block_isums(type, &x[i], &y[i], is, &ms);
            const float d = o_fp16_to_fp32(x[i].d) * o_fp16_to_fp32(y[i].d);
            for (int k = 0; k < 8; k++) acc[k] = fmaf(d, (float) is[k], acc[k]);
        }
        return hsum8(acc);
    }
    if (type == OT_Q4_0 || type == OT_Q8_0) {
        const int nb = n / OQK;
        const oblk_q8_0 * y = vy;
        for (int i = 0; i < nb; i++) {
            int32_t is[8], ms;
            const void * xb = type == OT_Q4_0 ? (const void *) ((const oblk_q4_0 *) vx + i) : (const void *) ((const oblk_q8_0 *) vx + i);
            const uint16_t xd = type == OT_Q4_0 ? ((const oblk_q4_0 *) vx)[i].d : ((const oblk_q8_0 *) vx)[i].d;
            o_block_isums(type, xb, &y[i], is, &ms);
            const float d = o_fp16_to_fp32(xd) * o_fp16_to_fp32(y[i].d);
            for (int k = 0; k < 8; k++) acc[k] = fmaf(d, (float) is[k], acc[k]);
        }
        return hsum8(acc);
    }
    const int nb = n / OQK_K;
    const oblk_q8_K * y = vy;
    if (type == OT_Q2_K) {   // per block: acc = fmadd(dmin, cvt(madd_epi16(mins, bsums)), acc) — lane k = m[2k]*bs[2k] + m[2k+1]*bs[2k+1] — THEN acc = fmadd(d, cvt(sumi), acc)
        const oblk_q2_K * x = vx;
        for (int i = 0; i < nb; i++) {
            int32_t is[8], ms;
            const float d = y[i].d * o_fp16_to_fp32(x[i].d), dmin = -y[i].d * o_fp16_to_fp32(x[i].dmin);
            for (int k = 0; k < 8; k++) {
                const int32_t prod = (x[i].scales[2 * k] >> 4) * y[i].bsums[2 * k] + (x[i].scales[2 * k + 1] >> 4) * y[i].bsums[2 * k + 1];
                acc[k] = fmaf(dmin, (float) prod, acc[k]);
            }
            o_block_isums(type, &x[i], &y[i], is, &ms);
            for (int k = 0; k < 8; k++) acc[k] = fmaf(d, (float) is[k], acc[k]);
        }
        return hsum8(acc);
    }
    if (type == OT_Q3_K) {                                                       // acc = fmadd(broadcast(d), cvt(sumi), acc) per block, hsum_float_8 (:4450-4533)
        const oblk_q3_K * x = vx;
        for (int i = 0; i < nb; i++) {
            int32_t is[8], ms;
            o_block_isums(type, &x[i], &y[i], is, &ms);
            const float d = y[i].d * o_fp16_to_fp32(x[i].d);
            for (int k = 0; k < 8; k++) acc[k] = fmaf(d, (float) is[k], acc[k]);
        }
        return hsum8(acc);
    }
    if (type == OT_Q6_K) {
        const oblk_q6_K * x = vx;
        for (int i = 0; i < nb; i++) {
            int32_t is[8], ms;
            o_block_isums(type, &x[i], &y[i], is, &ms);
            const float d = y[i].d * o_fp16_to_fp32(x[i].d);
            for (int k = 0; k < 8; k++) acc[k] = fmaf(d, (float) is[k], acc[k]);
        }
        return hsum8(acc);
    }
    float acc_m[4] = { 0 };
    float summs = 0.0f;
    for (int i = 0; i < nb; i++) {
        const uint8_t * scales; uint16_t xd, xdmin; const void * xb;
        if (type == OT_Q4_K) { const oblk_q4_K * x = (const oblk_q4_K *) vx + i; scales = x->scales; xd = x->d; xdmin = x->dmin; xb = x; }
        else                 { const oblk_q5_K * x = (const oblk_q5_K *) vx + i; scales = x->scales; xd = x->d; xdmin = x->dmin; xb = x; }
        const float d    =  y[i].d * o_fp16_to_fp32(xd);
        const float dmin = -y[i].d * o_fp16_to_fp32(xdmin);
        int32_t prod[4];
        for (int k = 0; k < 4; k++) {
            uint8_t sc, m0, m1;
            scale_min_k4(2 * k, scales, &sc, &m0); scale_min_k4(2 * k + 1, scales, &sc, &m1);
            const int16_t s0 = (int16_t) (y[i].bsums[4 * k] + y[i].bsums[4 * k + 1]), s1 = (int16_t) (y[i].bsums[4 * k + 2] + y[i].bsums[4 * k + 3]);
            prod[k] = m0 * s0 + m1 * s1;
        }
        if (type == OT_Q4_K) for (int k = 0; k < 4; k++) acc_m[k] = fmaf(dmin, (float) prod[k], acc_m[k]);
        else                 summs += dmin * (float) (prod[0] + prod[1] + prod[2] + prod[3]);
        int32_t is[8], ms;
        o_block_isums(type, xb, &y[i], is, &ms);
        for (int k = 0; k < 8; k++) acc[k] = fmaf(d, (float) is[k], acc[k]);
    }
    if (type == OT_Q4_K) {
        const float a0 = acc_m[0] + acc_m[2], a1 = acc_m[1] + acc_m[3];
        return hsum8(acc) + (a0 + a1);
    }
    return hsum8(acc) + summs;
}
