// TEST INFRASTRUCTURE — see oracle.h.  The non-matmul operators of the reference CPU path and the
// mul_mat / attention wiring, restated in scalar C (AVX arm summation order where the reference is SIMD).
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

// ---- fp16 lookup tables (ggml.c:2190-2201): value = f16(fn(f32(f16 bits))) for all 65536 bit patterns ----
static uint16_t t_exp[1 << 16], t_silu[1 << 16], t_gelu[1 << 16];
static int tables_ready = 0;

// ggml.c:1418-1421
static inline float gelu_f32(float x) {
    const float GELU_COEF_A = 0.044715f, SQRT_2_OVER_PI = 0.79788456080286535587989211986876f;
    return 0.5f * x * (1.0f + tanhf(SQRT_2_OVER_PI * x * (1.0f + GELU_COEF_A * x * x)));
}
// ggml.c:1479-1481
static inline float silu_f32(float x) { return x / (1.0f + expf(-x)); }

void o_tables_init(void) {
    if (tables_ready) return;
    for (int i = 0; i < (1 << 16); ++i) {
        const float f = o_fp16_to_fp32((uint16_t) i);
        t_gelu[i] = o_fp32_to_fp16(gelu_f32(f));
        t_silu[i] = o_fp32_to_fp16(silu_f32(f));
        t_exp[i]  = o_fp32_to_fp16(expf(f));
    }
    tables_ready = 1;
}
const uint16_t * o_table_exp_f16(void)  { o_tables_init(); return t_exp; }
const uint16_t * o_table_silu_f16(void) { o_tables_init(); return t_silu; }
const uint16_t * o_table_gelu_f16(void) { o_tables_init(); return t_gelu; }

// ggml.c:9028-9074 — sum of squares in double, scale = 1/sqrtf(mean + eps), y = x*scale
void o_rms_norm(const float * x, float * y, int n, int rows, float eps) {
    for (int r = 0; r < rows; r++, x += n, y += n) {
        double sum = 0.0;
        for (int i = 0; i < n; i++) sum += (double) (x[i] * x[i]);
        const float mean  = sum / n;
        const float scale = 1.0f / sqrtf(mean + eps);
        for (int i = 0; i < n; i++) y[i] = x[i] * scale;
    }
}

// ggml.c:8959-9008 — mean and centred variance in double
void o_norm(const float * x, float * y, int n, int rows, float eps) {
    for (int r = 0; r < rows; r++, x += n, y += n) {
        double sum = 0.0;
        for (int i = 0; i < n; i++) sum += (double) x[i];
        const float mean = sum / n;
        double sum2 = 0.0;
        for (int i = 0; i < n; i++) { const float v = x[i] - mean; y[i] = v; sum2 += (double) (v * v); }
        const float variance = sum2 / n;
        const float scale = 1.0f / sqrtf(variance + eps);
        for (int i = 0; i < n; i++) y[i] *= scale;
    }
}

// ggml.c:11006-11168 with ext_factor = 0, attn_factor = 1 (all configs): theta advances by repeated multiplication
void o_rope(const float * x, float * y, int head_dim, int n_head, int n_tok, const int32_t * pos, int mode, float freq_base, float freq_scale) {
    o_rope_ex(x, y, head_dim, head_dim, n_head, n_tok, pos, mode, freq_base, freq_scale);
}
// n_dims < head_dim (StableLM's partial rotary, llama.cpp:4860-4871) exists in NeoX mode only: the reference walks head_dim / n_dims blocks and
// theta keeps advancing from block to block (ggml.c:11133-11160)
void o_rope_ex(const float * x, float * y, int head_dim, int n_dims, int n_head, int n_tok, const int32_t * pos, int mode, float freq_base, float freq_scale) {
    const float theta_scale = powf(freq_base, -2.0f / n_dims);
    const int is_neox = mode & 2;
    for (int t = 0; t < n_tok; t++) {
        for (int h = 0; h < n_head; h++) {
            const float * src = x + ((size_t) t * n_head + h) * head_dim;
            float       * dst = y + ((size_t) t * n_head + h) * head_dim;
            float theta_base = (float) pos[t];
            if (!is_neox) {
                for (int i0 = 0; i0 < head_dim; i0 += 2) {
                    const float theta = freq_scale * theta_base;            // rope_yarn, ggml.c:10979-10989
                    const float cos_theta = cosf(theta) * 1.0f, sin_theta = sinf(theta) * 1.0f;
                    theta_base *= theta_scale;
                    const float x0 = src[i0], x1 = src[i0 + 1];
                    dst[i0]     = x0 * cos_theta - x1 * sin_theta;
                    dst[i0 + 1] = x0 * sin_theta + x1 * cos_theta;
                }
            } else {
                theta_base *= freq_scale;                                     // ggml.c:11136
                for (int ib = 0; ib < head_dim / n_dims; ib++)
                for (int ic = 0; ic < n_dims; ic += 2) {
                    const float theta = freq_scale * theta_base;            // (the reference applies freq_scale twice here)
                    const float cos_theta = cosf(theta) * 1.0f, sin_theta = sinf(theta) * 1.0f;
                    theta_base *= theta_scale;
                    const int i0 = ib * n_dims + ic / 2;
                    const float x0 = src[i0], x1 = src[i0 + n_dims / 2];
                    dst[i0]              = x0 * cos_theta - x1 * sin_theta;
                    dst[i0 + n_dims / 2] = x0 * sin_theta + x1 * cos_theta;
                }
            }
        }
    }
}

// ggml.c:10551-10620 — exp through the fp16 table, sum in double, scale by (float)(1/sum); masked entries exactly 0
void o_soft_max(const float * x, float * y, int nc, int nr) {
    o_tables_init();
    for (int r = 0; r < nr; r++, x += nc, y += nc) {
        float max = -INFINITY;
        for (int i = 0; i < nc; i++) if (x[i] > max) max = x[i];
        double sum = 0.0;
        for (int i = 0; i < nc; i++) {
            if (x[i] == -INFINITY) { y[i] = 0.0f; continue; }
            const float val = o_fp16_to_fp32(t_exp[o_fp32_to_fp16(x[i] - max)]);
            sum += (double) val;
            y[i] = val;
        }
        const float inv = (float) (1.0 / sum);
        for (int i = 0; i < nc; i++) y[i] *= inv;
    }
}

// ggml.c:1490-1498, 1433-1441 (GGML_SILU_FP16 / GGML_GELU_FP16 are defined, ggml.c:148-150)
void o_silu(const float * x, float * y, int n) { o_tables_init(); for (int i = 0; i < n; i++) y[i] = o_fp16_to_fp32(t_silu[o_fp32_to_fp16(x[i])]); }
void o_gelu(const float * x, float * y, int n) { o_tables_init(); for (int i = 0; i < n; i++) y[i] = o_fp16_to_fp32(t_gelu[o_fp32_to_fp16(x[i])]); }

// ggml_vec_dot_f16, AVX+F16C arm (ggml.c:1193-1228, macros :751-850): 32 halfs per step into 4 accumulators of
// 8 fp32 lanes (fma), tree-reduced in fp32, leftovers added in double.
float o_vec_dot_f16(int n, const uint16_t * x, const uint16_t * y) {
    float sum[4][8] = { { 0 } };
    const int np = n & ~31;
    for (int i = 0; i < np; i += 32)
        for (int j = 0; j < 4; j++)
            for (int l = 0; l < 8; l++)
                sum[j][l] = fmaf(o_fp16_to_fp32(x[i + 8 * j + l]), o_fp16_to_fp32(y[i + 8 * j + l]), sum[j][l]);
    for (int l = 0; l < 8; l++) { sum[0][l] += sum[2][l]; sum[1][l] += sum[3][l]; }
    for (int l = 0; l < 8; l++) sum[0][l] += sum[1][l];
    float t[4]; for (int l = 0; l < 4; l++) t[l] = sum[0][l] + sum[0][l + 4];
    double sumf = (t[0] + t[1]) + (t[2] + t[3]);
    for (int i = np; i < n; ++i) sumf += (double) (o_fp16_to_fp32(x[i]) * o_fp16_to_fp32(y[i]));
    return (float) sumf;
}

// ggml_compute_forward_mul_mat (ggml.c:9388-9597): INIT quantises every activation row to the weight type's
// vec_dot_type, then each output element is one vec_dot.
void o_mul_mat(int type, const void * w, int ne00, int ne01, const float * act, int n_tok, float * out) {
    const int vdt = o_vec_dot_type(type);
    const size_t arow = vdt == OT_F32 ? (size_t) 4 * ne00 : vdt == OT_F16 ? (size_t) 2 * ne00 : o_row_size(vdt, ne00);
    const size_t wrow = o_row_size(type, ne00);
    char * wdata = malloc(arow * n_tok);
    for (int t = 0; t < n_tok; t++) {
        const float * a = act + (size_t) t * ne00;
        if (vdt == OT_F32) memcpy(wdata + t * arow, a, arow);
        else if (vdt == OT_F16) { uint16_t * d = (uint16_t *) (wdata + t * arow); for (int i = 0; i < ne00; i++) d[i] = o_fp32_to_fp16(a[i]); }
        else o_quantize_act(vdt, a, wdata + t * arow, ne00);
    }
    #pragma omp parallel for schedule(static)
    for (int r = 0; r < ne01; r++) {
        const char * wr = (const char *) w + (size_t) r * wrow;
        for (int t = 0; t < n_tok; t++) {
            out[(size_t) t * ne01 + r] = type == OT_F16 ? o_vec_dot_f16(ne00, (const uint16_t *) wr, (const uint16_t *) (wdata + t * arow))
                                                        : o_vec_dot(type, ne00, wr, wdata + t * arow);
        }
    }
    free(wdata);
}

// ggml_compute_forward_get_rows_q (ggml.c:10161-10188)
void o_get_rows(int type, const void * w, int ne00, const int32_t * ids, int n, float * out) {
    const size_t wrow = o_row_size(type, ne00);
    for (int i = 0; i < n; i++) o_dequantize_row(type, (const char *) w + (size_t) ids[i] * wrow, out + (size_t) i * ne00, ne00);
}

// llm_build_kv_store + llm_build_kqv up to kqv_merged_cont (llama.cpp:3546-3577, 3698-3764), one layer.
// Reference layouts: K rows [cell][n_embd_gqa] f16; V transposed [n_embd_gqa][n_ctx] f16.
void o_attention(uint16_t * k_cache, uint16_t * v_cache, int n_ctx, const float * q_cur, const float * k_cur, const float * v_cur,
                 const float * mask, int hd, int n_head, int n_head_kv, int n_tok, int kv_head, int n_kv, float * out) {
    o_attention_alibi(k_cache, v_cache, n_ctx, q_cur, k_cur, v_cur, mask, hd, n_head, n_head_kv, n_tok, kv_head, n_kv, 0.0f, out);
}
// max_alibi_bias > 0: ggml_alibi between the scale and the mask (llama.cpp:3735-3740; ggml_compute_forward_alibi_f32, ggml.c:10736-10793)
void o_attention_alibi(uint16_t * k_cache, uint16_t * v_cache, int n_ctx, const float * q_cur, const float * k_cur, const float * v_cur,
                       const float * mask, int hd, int n_head, int n_head_kv, int n_tok, int kv_head, int n_kv, float max_alibi_bias, float * out) {
    const int n_heads_log2_floor = 1 << (int) floor(log2(n_head));
    const float m0 = powf(2.0f, -(max_alibi_bias) / n_heads_log2_floor), m1 = powf(2.0f, -(max_alibi_bias / 2.0f) / n_heads_log2_floor);
    const int n_embd = hd * n_head, n_embd_gqa = hd * n_head_kv, r = n_head / n_head_kv;
    for (int t = 0; t < n_tok; t++) {
        for (int e = 0; e < n_embd_gqa; e++) {
            k_cache[(size_t) (kv_head + t) * n_embd_gqa + e] = o_fp32_to_fp16(k_cur[(size_t) t * n_embd_gqa + e]);
            v_cache[(size_t) e * n_ctx + kv_head + t]        = o_fp32_to_fp16(v_cur[(size_t) t * n_embd_gqa + e]);
        }
    }
    const float scale = 1.0f / sqrtf((float) hd);                               // llama.cpp:5189-5193
    #pragma omp parallel for schedule(static)
    for (int h = 0; h < n_head; h++) {
        uint16_t * q16 = malloc((size_t) 2 * hd);
        uint16_t * p16 = malloc((size_t) 2 * n_kv);
        float    * kq  = malloc((size_t) 4 * n_kv);
        for (int t = 0; t < n_tok; t++) {
            for (int e = 0; e < hd; e++) q16[e] = o_fp32_to_fp16(q_cur[((size_t) t * n_head + h) * hd + e]);
            for (int i = 0; i < n_kv; i++) {
                float s = o_vec_dot_f16(hd, k_cache + (size_t) i * n_embd_gqa + (h / r) * hd, q16);
                s = s * scale;
                if (max_alibi_bias > 0.0f) {
                    const float m_k = h < n_heads_log2_floor ? powf(m0, h + 1) : powf(m1, 2 * (h - n_heads_log2_floor) + 1);
                    s = i * m_k + s;                                         // ggml.c:10789
                }
                kq[i] = s + mask[(size_t) t * n_kv + i];
            }
            o_soft_max(kq, kq, n_kv, 1);
            for (int i = 0; i < n_kv; i++) p16[i] = o_fp32_to_fp16(kq[i]);
            for (int e = 0; e < hd; e++)
                out[(size_t) t * n_embd + h * hd + e] = o_vec_dot_f16(n_kv, v_cache + (size_t) ((h / r) * hd + e) * n_ctx, p16);
        }
        free(q16); free(p16); free(kq);
    }
}
